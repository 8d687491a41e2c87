#!/bin/bash
# tools/gpu_ab.sh -- diagnostics at sustained clocks (bench.py preheats the GPU): a table of
# configurations, optionally with parts of the kernels skipped.  Run through gpurun; edit the
# lists below for the experiment at hand.  SPEEXHIP_SKIP bits: 2 = window staging, 4 = FIR loop,
# 8 = stores, 64 = return at once (bare dispatch), 128 = return after staging.
# Other switches: SPEEXHIP_SPLITS, SPEEXHIP_WAVES, SPEEXHIP_ROWS, SPEEXHIP_HELPERS, SPEEXHIP_PAD,
# SPEEXHIP_SLIDE_WAVES.  Boxes differ by up to 10 %: compare variants
# inside ONE run (e.g. two builds of libspeexhip.so swapped by the script), never across runs.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
run() { echo -n "$1 $2 steps=$3 : " >> $O/ab.txt; env $1 timeout 300 python bench.py $2 --steps $3 --warmup 20 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'], 'value', d['value'], 'hbm', d['roofline']['frac'], 'valu', d['valu']['frac'], 'path', d['config']['fast_path'])" >> $O/ab.txt; }
for CFG in cfg2 cfg3 cfg4 f3; do
  run "BENCH_STREAMS=1" "--config $CFG" 1000
  run "BENCH_STREAMS=32" "--config $CFG" 100
done
for SKIP in 10 12 6; do run "BENCH_STREAMS=32 SPEEXHIP_SKIP=$SKIP" "--config cfg2" 100; done
cat $O/ab.txt
