# tools/r05_w16_3ch.sh -- round 5: an int16 window for the two-period plan of three channels (SPEEXHIP_W16_3CH=1) against
# the library's default (three channels take an int16 window only through their phase-pair plans); gpurun.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 48000,11025 44100,8000 48000,22050 44100,16000 44100,32000 32000,11025 96000,11025 96000,44100; do
for SHAPE in 1,441000 1,1048576 8,131072 32,131072 32,1048576; do for V in 0 1; do
SPEEXHIP_W16_3CH=$V python bench.py --custom 3,$P,7 --streams ${SHAPE%,*} --frames ${SHAPE#*,} --steps 8 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch 3 $P streams ${SHAPE%,*} frames ${SHAPE#*,} w16_3ch=$V: %.1f us parity %s' % (d['roofline']['launch_us'], d.get('parity', {}).get('max_abs_diff_lsb')))"
done; done; done
