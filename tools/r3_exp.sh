#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_prev.so
{
timeout 2400 python -m pytest tests -m gpu -x -q -k float 2>&1 | tail -4
for rep in 1 2; do
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --steps 300
done
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --custom 1,44100,48000,7 --streams 32 --steps 60
} > $E1 2>&1
cat $E1
