# tools/r05_wide_grid.sh -- round 5: phase-group splits x tap-range shares over the widest windows (num = 640, 1280),
# three launch shapes, q7: what the launch rule takes against the best of the grid (gpurun).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 32000,11025 96000,11025 48000,11025; do for CH in 1 2 4; do for SHAPE in 1,1048576 8,131072 32,131072; do
S=${SHAPE%,*}; F=${SHAPE#*,}
for SP in rule 1 2 3 4 6 8; do for KS in rule 2; do for TO in rule 1; do
[ $TO = 1 ] && [ "$SP$KS" != rulerule ] && continue
unset SPEEXHIP_SPLITS SPEEXHIP_KSPLIT SPEEXHIP_TOUCH
[ $SP != rule ] && export SPEEXHIP_SPLITS=$SP
[ $KS != rule ] && export SPEEXHIP_KSPLIT=$KS
[ $TO != rule ] && export SPEEXHIP_TOUCH=$TO
python bench.py --custom $CH,$P,7 --streams $S --frames $F --steps 6 --warmup 2 --reps 1 --preheat-ms 30 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P streams $S frames $F splits=$SP ksplit=$KS touch=$TO: %.1f us' % d['roofline']['launch_us'])"
done; done; done; done; done; done
