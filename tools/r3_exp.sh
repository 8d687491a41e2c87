#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_cxxloop.so
{
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 1,44100,48000,7 --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 1,48000,44100,5 --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 3,44100,48000,7 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 6,44100,48000,7 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --steps 300
} > $E1 2>&1
cat $E1
