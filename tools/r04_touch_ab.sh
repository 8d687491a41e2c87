#!/bin/bash
# tools/r04_touch_ab.sh -- the tap rows fetched into L2 at the top of every workgroup (touch_rows), against the same
# library with the fetch off (SPEEXHIP_TOUCH=0), forced on (=1) and by the host's rule (unset), same box: launch time,
# bench.py's parity block on.  (profiles/r04_touch_ab.txt is the first form: on everywhere against off.)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # custom-or-config streams frames
  for SK in off on rule; do
    if [ $SK = rule ]; then unset SPEEXHIP_TOUCH; elif [ $SK = on ]; then export SPEEXHIP_TOUCH=1; else export SPEEXHIP_TOUCH=0; fi
    python bench.py $1 --streams $2 --frames $3 --steps ${STEPS:-30} --warmup 5 --reps 3 --preheat-ms 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
par = d.get('parity') or {}
print('%-28s S=%-3s F=%-8s %-9s %8.1f us  valu %.3f  mismatch %s' % ('$1', '$2', '$3', 'touch $SK', d['roofline']['launch_us'], d['valu']['frac'], par.get('mismatch_rate')))"
  done
}
for C in 2,48000,11025,7 3,48000,11025,7 1,48000,11025,7 1,48000,22050,7 3,44100,16000,7 2,44100,8000,7 4,48000,11025,7 2,48000,44100,7 6,44100,8000,7; do run "--custom $C" 32 131072; done
run "--custom 2,48000,11025,7" 32 1048576
run "--custom 3,48000,11025,7" 32 1048576
for CFG in cfg2 cfg4 f3; do run "--config $CFG" 1 1048576; run "--config $CFG" 32 1048576; done
run "--custom 2,44100,48000,10" 1 1048576
run "--custom 2,44100,48000,10" 32 1048576
