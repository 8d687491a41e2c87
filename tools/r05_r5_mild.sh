# tools/r05_r5_mild.sh -- round 5: groups of 5 phases on an unpadded window where the ratio's R = 10 plan runs unpadded too or
# the conflicts are two lanes per bank (SPEEXHIP_R5_MILD=1, the default) against the library before (=0); gpurun.
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { for V in 0 1; do
SPEEXHIP_R5_MILD=$V python bench.py --custom $1,$2,7 --streams $3 --frames $4 --steps 8 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $1 $2 streams $3 frames $4 mild=$V: %.1f us parity %s' % (d['roofline']['launch_us'], d.get('parity', {}).get('max_abs_diff_lsb')))"
done; }
for P in 88200,48000 44100,8000; do for CH in 3 5 6 7; do for SHAPE in 1,1048576 8,131072 32,131072 32,1048576; do one $CH $P ${SHAPE%,*} ${SHAPE#*,}; done; done; done
for P in 44100,48000 22050,48000; do for CH in 3 6 7; do for SHAPE in 1,1048576 1,441000; do one $CH $P ${SHAPE%,*} ${SHAPE#*,}; done; done; done
