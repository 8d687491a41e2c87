#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ablate.txt
for S in 1 32; do
  for SKIP in 0 4 8 12 15; do
    echo -n "streams=$S skip=$SKIP " >> $O/ablate.txt
    SPEEXHIP_SKIP=$SKIP timeout 200 python bench.py --steps 50 --warmup 5 --streams $S --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'])" >> $O/ablate.txt
  done
done
cat $O/ablate.txt
