#!/bin/bash
# tools/gpu_ab.sh -- diagnostics: single-stream geometry variants of the one-shot period kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
for SPL in 1 2 4 8; do for SKIP in 0 2 4 8 32; do
  echo -n "splits=$SPL skip=$SKIP " >> $O/ab.txt
  SPEEXHIP_SPLITS=$SPL SPEEXHIP_SKIP=$SKIP timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'])" >> $O/ab.txt
done; done
cat $O/ab.txt
