#!/bin/bash
# tools/gpu_ab.sh -- diagnostics at sustained clocks: staging helper waves for split tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or baseline or edge or many_rates" 2>&1 | tail -2 >> $O/ab.txt
run() { echo -n "$1 $2 steps=$3 : " >> $O/ab.txt; env $1 timeout 300 python bench.py $2 --steps $3 --warmup 20 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'])" >> $O/ab.txt; }
for rep in 1 2; do for H in 0 1; do
run "BENCH_STREAMS=1 SPEEXHIP_HELPERS=$H" "--config cfg2" 2000
run "BENCH_STREAMS=1 SPEEXHIP_HELPERS=$H SPEEXHIP_SPLITS=4" "--config cfg2" 2000
run "BENCH_STREAMS=1 SPEEXHIP_HELPERS=$H" "--config cfg4" 1000
run "BENCH_STREAMS=1 SPEEXHIP_HELPERS=$H" "--custom 2,48000,44100,5" 1000
done; done
cat $O/ab.txt
