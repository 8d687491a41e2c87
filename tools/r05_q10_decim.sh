cd ${GRAFT_REPO_ROOT:-/root/repo}
for P in 48000,11025 44100,8000 48000,22050 44100,16000 96000,11025 56000,48000 88000,8000; do for CH in 1 2; do for Q in 7 10; do
python bench.py --custom $CH,$P,$Q --streams 32 --frames 131072 --steps 10 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P q$Q: launch_us %.1f valu %.3f (%s) path %d taps %d parity %s' % (d['roofline']['launch_us'], d['valu']['frac'], d['valu']['arithmetic'], d['config']['fast_path'], d['config']['filt_len'], d.get('parity',{}).get('max_abs_diff_lsb')))"
done; done; done
