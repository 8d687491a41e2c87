#!/bin/bash
# tools/perf_sweep.sh -- fast-mode launch time and vector-ALU fraction over channel counts x common rate pairs
# (32 streams x 131072 frames, q7): a table to spot configurations that sit far below their neighbours
# (a planner rule gone wrong shows up as a 2x outlier).  Run through gpurun; prints rows sorted by valu.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
PAIRS=${PAIRS:-"44100,48000 48000,44100 22050,48000 48000,22050 16000,44100 44100,16000 8000,44100 44100,8000 11025,48000 48000,11025 88200,48000 96000,44100 32000,44100 44100,32000 8000,48000 48000,8000 16000,48000 48000,16000 24000,48000 96000,48000 32000,48000 48000,32000 44100,22050 8000,16000 32000,11025 96000,11025"}
for CH in ${CHANNELS:-1 2 3 4 5 6 7 8}; do
  for P in $PAIRS; do
    python bench.py --custom $CH,$P,${Q:-7} --streams 32 --frames ${FRAMES:-131072} --steps 8 --warmup 3 --reps ${REPS:-2} --preheat-ms 50 --no-cpu-baseline --no-parity ${EXTRA} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%.4f valu | %8.1f us | hbm %.3f | ch %s %s -> %s | taps %d | path %d | %.0f Msamples/s' % (d['valu']['frac'], d['roofline']['launch_us'], d['roofline']['frac'], '$CH', '$P'.split(',')[0], '$P'.split(',')[1], d['config']['filt_len'], d['config']['fast_path'], d['value']))"
  done
done | sort -n
