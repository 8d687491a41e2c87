#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3; mkdir -p $O
{
SPEEXHIP_WALK=1 SPEEXHIP_WALK_VERBOSE=1 timeout 120 python bench.py --no-cpu-baseline --reps 1 --streams 32 --config cfg4 2>&1 | tail -12 | cut -c1-400
for sk in 8 2 10; do
for w in 0 1; do
 echo "SKIP=$sk WALK=$w"; SPEEXHIP_SKIP=$sk SPEEXHIP_WALK=$w timeout 120 python bench.py --no-cpu-baseline --no-parity --reps 3 --streams 32 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['roofline']['launch_us'])"
done; done
} > $O/walk_ab3.txt 2>&1
cat $O/walk_ab3.txt
