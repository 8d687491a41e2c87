cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 96000,11025 32000,11025; do for SHAPE in 32,131072 1,1048576 8,131072; do for MF in 8 9; do
SPEEXHIP_MIN_FILL=$MF python bench.py --custom 7,$P,7 --streams ${SHAPE%,*} --frames ${SHAPE#*,} --steps 6 --warmup 2 --reps 2 --preheat-ms 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch 7 $P $SHAPE min_fill=$MF: %.1f us path %d parity %s' % (d['roofline']['launch_us'], d['config']['fast_path'], d.get('parity', {}).get('max_abs_diff_lsb')))"
done; done; done
