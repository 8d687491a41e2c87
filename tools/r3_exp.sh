#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_r02.so
{
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or baseline or window_layout or eight_channel or many_rates or edge" 2>&1 | tail -3
for rep in 1 2 3; do
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --steps 300
done
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 1,44100,48000,7 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --frames 441000 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg4 --steps 200
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 2 --steps 300
} > $E1 2>&1
cat $E1
