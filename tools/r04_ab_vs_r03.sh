#!/bin/bash
# tools/r04_ab_vs_r03.sh -- same-box A/B of the round-4 library against the round-3 one (ab/libspeexhip_r03.so, loaded
# through SPEEXHIP_LIB_PATH) on rows of the sweep that read slower than in profiles/r03_perf_sweep.txt, and on the
# BASELINE configs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=node-speex-resampler_amd
row() {  # custom streams frames
  for lib in $L/ab/libspeexhip_r03.so $L/libspeexhip.so; do
    SPEEXHIP_LIB_PATH=$lib python bench.py --custom $1 --streams $2 --frames $3 --steps 12 --warmup 3 --reps 3 --preheat-ms 80 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-22s %-18s S=%-3s F=%-8s %8.1f us (min %.1f)' % ('$lib'.split('/')[-1], '$1', '$2', '$3', d['roofline']['launch_us'], d['roofline']['launch_us_min']))"
  done
}
for rep in 1 2; do
for C in 1,8000,48000,7 2,32000,48000,7 2,48000,32000,7 2,48000,44100,7 2,88200,48000,7 6,44100,16000,7 2,48000,11025,7; do row $C 32 131072; done
row 2,44100,48000,7 32 1048576; row 2,44100,48000,7 1 1048576; row 8,48000,44100,5 32 1048576; row 1,24000,48000,5 32 1048576
done 2>&1 | tee $O/ab_vs_r03.txt
