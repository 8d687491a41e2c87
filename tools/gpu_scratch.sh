#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_prev.so
{
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or baseline or small_ratio or n_to_one or slide or many_rates or edge" 2>&1 | tail -3
for rep in 1 2; do
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg3 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config f3 --steps 300
done
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg3 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config f3 --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 2,48000,8000,7 --streams 32 --steps 40
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 2,96000,48000,7 --streams 32 --steps 40
} > $E1 2>&1
cat $E1
