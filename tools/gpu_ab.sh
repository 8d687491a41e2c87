#!/bin/bash
# tools/gpu_ab.sh -- diagnostics at sustained clocks: slide kernel, waves per workgroup
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
run() { echo -n "$1 $2 steps=$3 : " >> $O/ab.txt; env $1 timeout 300 python bench.py $2 --steps $3 --warmup 20 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'], 'valu', d['valu']['frac'])" >> $O/ab.txt; }
for C in "2,48000,24000,7" "2,44100,44100,7" "2,24000,48000,7" "1,24000,48000,10" "1,24000,48000,5" "1,16000,48000,7"; do
for W in 16 8 4; do
run "BENCH_STREAMS=32 SPEEXHIP_SLIDE_WAVES=$W" "--custom $C" 50
done; done
cat $O/ab.txt
