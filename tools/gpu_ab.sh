#!/bin/bash
# tools/gpu_ab.sh -- diagnostics at sustained clocks (bench.py preheats)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $O/ab.txt
run() { echo -n "$1 $2 steps=$3 : " >> $O/ab.txt; env $1 timeout 300 python bench.py $2 --steps $3 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'], 'value', d['value'], 'valu', d['valu']['frac'])" >> $O/ab.txt; }
run "BENCH_STREAMS=32 SPEEXHIP_PAD=4" "--config cfg2" 100
run "BENCH_STREAMS=32 SPEEXHIP_PAD=4 SPEEXHIP_SKIP=12" "--config cfg2" 100
run "BENCH_STREAMS=1" "--config cfg4" 500
run "BENCH_STREAMS=32" "--config cfg4" 100
for C in "2,48000,44100,5" "2,48000,44100,10" "2,96000,44100,7" "2,32000,44100,7" "4,48000,44100,5" "6,44100,48000,7"; do
run "BENCH_STREAMS=32" "--custom $C" 50
done
cat $O/ab.txt
