#!/bin/bash
# tools/r04_phases_decim.sh -- where a small decimator launch (32 streams x 131072 frames) spends its time: the launch with
# parts of the period kernel skipped (SPEEXHIP_SKIP bits: 2 staging, 4 FIR loop, 8 stores, 64 return at once, 128 return
# after staging).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for C in ${CASES:-1,48000,11025,7 3,48000,11025,7 2,48000,11025,7 1,48000,22050,7}; do
  SPEEXHIP_PLAN_VERBOSE=1 python bench.py --custom $C --streams 32 --frames 131072 --steps 2 --warmup 1 --reps 1 --preheat-ms 1 --no-cpu-baseline --no-parity 2>&1 | grep -m1 "launch:"
  for SKIP in 0 64 128 130 14 12 6 10; do
    SPEEXHIP_SKIP=$SKIP python bench.py --custom $C --streams 32 --frames 131072 --steps 20 --warmup 3 --reps 3 --preheat-ms 50 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s skip %3d  %7.1f us' % ('$C', $SKIP, d['roofline']['launch_us']))"
  done
done
