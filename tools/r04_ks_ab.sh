#!/bin/bash
# tools/r04_ks_ab.sh -- tap-range shares (SPEEXHIP_KSPLIT) forced on launches that are not split over phase groups:
# launch time at 32 streams x 131072 frames, bench.py's parity block on.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for C in ${CASES:-3,48000,11025,7 2,48000,11025,7 1,48000,11025,7 1,48000,22050,7 1,96000,44100,7 3,44100,8000,7 1,44100,32000,7 1,44100,8000,7 3,44100,16000,7 1,32000,44100,7 2,48000,44100,7}; do
  for KS in ${KSS:-default 2 4}; do
    if [ $KS = default ]; then unset SPEEXHIP_KSPLIT; else export SPEEXHIP_KSPLIT=$KS; fi
    SPEEXHIP_PLAN_VERBOSE=1 python bench.py --custom $C --streams ${STREAMS:-32} --frames ${FRAMES:-131072} --steps 20 --warmup 3 --reps 3 --preheat-ms 50 --no-cpu-baseline 2>&1 | python3 -c "
import sys, json
shape = ''
for l in sys.stdin:
    if 'launch:' in l and not shape: shape = l.strip().split('launch:')[1]
    if l.startswith('{'):
        d = json.loads(l)
        par = d.get('parity') or {}
        print('%-18s ks %-7s %7.1f us  valu %.3f  parity %s |%s' % ('$C', '$KS', d['roofline']['launch_us'], d['valu']['frac'], {k: par.get(k) for k in ('max_abs_lsb', 'mismatch_rate', 'ok')}, shape))"
  done
done
