#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_b64.so
{
for rep in 1 2 3; do
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 32 --steps 100
done
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg4 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 4,44100,48000,7 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 2,48000,44100,7 --streams 32 --steps 100
} > $E1 2>&1
cat $E1
