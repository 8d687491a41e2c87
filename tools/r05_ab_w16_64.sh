# tools/r05_ab_w16_64.sh -- round 5: fp64 period kernel, float window (SPEEXHIP_W16_ALWAYS=0) against the int16 window
# (=1) and the launch rule (unset), q10 decimators over launch shapes (gpurun).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 48000,11025 44100,8000 44100,16000 48000,22050 44100,32000; do for CH in 1 2 4; do for SHAPE in 1,1048576 8,131072 32,131072 32,1048576; do
S=${SHAPE%,*}; F=${SHAPE#*,}
for W in 0 1 rule; do
if [ $W = rule ]; then unset SPEEXHIP_W16_ALWAYS; else export SPEEXHIP_W16_ALWAYS=$W; fi
python bench.py --custom $CH,$P,10 --streams $S --frames $F --steps 8 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P streams $S frames $F w16=$W: %.1f us' % d['roofline']['launch_us'])"
done; done; done; done
