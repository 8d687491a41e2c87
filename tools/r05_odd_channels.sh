cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 44100,48000 48000,44100 48000,11025; do for CH in 3 4 5 6 7 8; do
python bench.py --custom $CH,$P,7 --streams 32 --frames 131072 --steps 10 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P: launch_us %.1f valu %.3f per-channel-us %.2f path %d parity %s' % (d['roofline']['launch_us'], d['valu']['frac'], d['roofline']['launch_us']/$CH, d['config']['fast_path'], d.get('parity',{}).get('max_abs_diff_lsb')))"
done; done
