#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_cxxloop.so
{
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or baseline or window_layout or eight_channel or many_rates or ragged or batched or edge or configs4" 2>&1 | tail -3
python tools/stamps.py --streams 1 --launches 1 2>&1 | grep "phase\|last stamp"
for rep in 1 2; do
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 32 --steps 100
done
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg4 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg4 --steps 200
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 8 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 1,44100,48000,7 --steps 300
} > $E1 2>&1
cat $E1
