cd ${GRAFT_REPO_ROOT:-/root/repo}
PAIRS="44100,48000 48000,44100 22050,48000 48000,22050 16000,44100 44100,16000 8000,44100 44100,8000 48000,11025 96000,44100 32000,44100 44100,32000 24000,48000 48000,8000 32000,11025 96000,11025"
for CH in 1 2 3 4 6 8; do for P in $PAIRS; do
python bench.py --custom $CH,$P,10 --streams 32 --frames 131072 --steps 6 --warmup 2 --reps 2 --preheat-ms 40 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%.4f valu (%s) | %8.1f us | ch $CH %s -> %s | taps %d | path %d | acc %s | parity %s' % (d['valu']['frac'], d['valu']['arithmetic'], d['roofline']['launch_us'], '$P'.split(',')[0], '$P'.split(',')[1], d['config']['filt_len'], d['config']['fast_path'], d['config'].get('accumulate','')[:12], d.get('parity',{}).get('max_abs_diff_lsb')))"
done; done | sort -n
