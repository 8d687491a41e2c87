#!/bin/bash
# tools/r05_ab_cfg4.sh -- same-box A/B of BASELINE configs[3] (8 channels 48k->44.1k q5) and its neighbours: the round-4
# library against this build (row mapping of the padded 8-channel instance, whole-frame stores through permlane swaps).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L4=node-speex-resampler_amd/ab/libspeexhip_r04.so; L5=node-speex-resampler_amd/libspeexhip.so
for ARGS in "--config cfg4 --streams 32" "--config cfg4 --streams 1" "--custom 8,32000,44100,7 --streams 32 --frames 262144" "--custom 8,48000,11025,7 --streams 32 --frames 262144" "--config cfg2 --streams 32" "--config cfg2 --streams 1"; do
  for rep in 1 2; do
    bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$L4 $L5" $ARGS --steps 60 --warmup 10
  done
done
