cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 96000,11025 32000,11025; do for CH in 1 2; do for MF in 4 8; do for S in 32 1; do
SPEEXHIP_MIN_FILL=$MF python bench.py --custom $CH,$P,10 --streams $S --frames 131072 --steps 10 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('min_fill $MF ch $CH $P q10 streams $S: launch_us %.1f valu %.3f path %d parity %s %s' % (d['roofline']['launch_us'], d['valu']['frac'], d['config']['fast_path'], d.get('parity',{}).get('max_abs_diff_lsb'), d.get('parity',{}).get('mismatch_rate')))"
done; done; done; done
