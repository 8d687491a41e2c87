#!/bin/bash
# tools/r04_pad_ab.sh -- bank padding of the period kernel's window at a finer grain than 16 bytes (SPEEXHIP_PAD, elements):
# launch time at 32 streams x 131072 frames, bench.py's parity block on.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for C in ${CASES:-1,48000,11025,7 2,48000,11025,7 3,48000,11025,7 1,48000,22050,7 1,96000,44100,7 1,44100,32000,7 3,44100,8000,7 2,48000,44100,7}; do
  for PAD in ${PADS:-default 1 2 4 6 10}; do
    if [ $PAD = default ]; then unset SPEEXHIP_PAD; else export SPEEXHIP_PAD=$PAD; fi
    SPEEXHIP_PLAN_VERBOSE=1 python bench.py --custom $C --streams 32 --frames 131072 --steps 20 --warmup 3 --reps 3 --preheat-ms 50 --no-cpu-baseline 2>&1 | python3 -c "
import sys, json
shape = ''
for l in sys.stdin:
    if 'launch:' in l and not shape: shape = l.strip().split('launch:')[1]
    if l.startswith('{'):
        d = json.loads(l)
        print('%-18s pad %-7s %7.1f us  valu %.3f  parity %s |%s' % ('$C', '$PAD', d['roofline']['launch_us'], d['valu']['frac'], json.dumps(d.get('parity'))[:90], shape))"
  done
done
