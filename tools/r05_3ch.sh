#!/bin/bash
# three channels: round-4 library against this build (ISA loop + tap-range shares for the two-period plan), same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L4=node-speex-resampler_amd/ab/libspeexhip_r04.so; L5=node-speex-resampler_amd/libspeexhip.so
for P in 44100,48000 48000,44100 48000,11025 44100,32000 44100,16000 44100,8000 48000,22050 8000,44100; do
  for ARGS in "--streams 32 --frames 131072" "--streams 1 --frames 441000" "--streams 32 --frames 1048576"; do
    bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$L4 $L5" --custom 3,$P,7 $ARGS --steps 20 --warmup 5 2>&1 | sed "s/SPEEXHIP_LIB_PATH=node-speex-resampler_amd\///"
  done
done
