#!/bin/bash
# one-stream launches of quality-10 filters: fp64 accumulate (no tap-range shares in the period kernel's fp64 instances) against the fp32 chain
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
for C in 2,48000,11025,10 2,44100,8000,10 1,48000,22050,10 2,44100,48000,10 2,48000,44100,10 2,48000,8000,10 1,96000,48000,10; do
  for F in 48000 441000 1048576; do
    for M in fast fast_f32; do
      python bench.py --custom $C --mode $M --streams 1 --frames $F --steps 30 --warmup 5 --reps 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-20s F=%-8s %-9s %8.1f us  path %d  parity %s' % ('$C', '$F', '$M', d['roofline']['launch_us'], d['config']['fast_path'], d.get('parity', {}).get('max_abs_diff_lsb')))"
    done
  done
done 2>&1 | tee $O/q10_small.txt
