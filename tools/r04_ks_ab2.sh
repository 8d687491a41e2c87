#!/bin/bash
# tools/r04_ks_ab2.sh -- phase pairs x tap-range shares on the stereo / 4-channel decimators the unsplit-shares rule left
# behind (SPEEXHIP_PP, SPEEXHIP_KSPLIT forced), 32 streams x 131072 frames.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for C in ${CASES:-2,48000,11025,7 2,44100,8000,7 4,48000,11025,7 1,48000,11025,7}; do
  for PP in 0 1; do for KS in default 2 4; do
    if [ $KS = default ]; then unset SPEEXHIP_KSPLIT; else export SPEEXHIP_KSPLIT=$KS; fi
    SPEEXHIP_PP=$PP SPEEXHIP_PLAN_VERBOSE=1 python bench.py --custom $C --streams 32 --frames 131072 --steps 20 --warmup 3 --reps 3 --preheat-ms 50 --no-cpu-baseline --no-parity 2>&1 | python3 -c "
import sys, json
shape = ''
for l in sys.stdin:
    if 'launch:' in l and not shape: shape = l.strip().split('launch:')[1]
    if l.startswith('{'):
        d = json.loads(l)
        print('%-18s pp $PP ks %-7s %7.1f us  valu %.3f |%s' % ('$C', '$KS', d['roofline']['launch_us'], d['valu']['frac'], shape))"
  done; done
done
