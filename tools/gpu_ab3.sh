#!/bin/bash
# tools/gpu_ab3.sh -- diagnostics: sensitivity of the measured step time to the length of the run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab3.txt
run() { echo -n "$* : " >> $O/ab3.txt; env $1 timeout 300 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'], 'ms_per_step', d['ms_per_step'])" >> $O/ab3.txt; }
for rep in 1 2; do
run BENCH_STREAMS=1 200 20
run BENCH_STREAMS=1 2000 20
run BENCH_STREAMS=1 2000 2000
run BENCH_STREAMS=1 20000 2000
run BENCH_STREAMS=32 30 5
run BENCH_STREAMS=32 100 10
run BENCH_STREAMS=32 300 100
run "BENCH_STREAMS=32 SPEEXHIP_PERSISTENT=0" 30 5
run "BENCH_STREAMS=32 SPEEXHIP_PERSISTENT=0" 100 10
run "BENCH_STREAMS=32 SPEEXHIP_PERSISTENT=0" 300 100
done
cat $O/ab3.txt
