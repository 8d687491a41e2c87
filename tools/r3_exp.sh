#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
A=$R/node-speex-resampler_amd/libspeexhip.so; B=$R/node-speex-resampler_amd/ab/libspeexhip_r02.so
{
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or baseline or window_layout or eight_channel or many_rates or edge or single_stream or float or mono or node" 2>&1 | tail -3
echo "== K-split (helper waves compute) A/B"
bash tools/gpu_ab.sh SPEEXHIP_KSPLIT "0 1" --custom 1,44100,48000,7 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_KSPLIT "0 1" --frames 441000 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_KSPLIT "0 1" --frames 131072 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_KSPLIT "0 1" --frames 16384 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_KSPLIT "0 1" --custom 1,48000,44100,5 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_KSPLIT "0 1" --custom 2,44100,48000,7 --frames 441000 --io float --steps 300
echo "== round-2 library (git d74056f) against this one, same box"
for rep in 1 2; do
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --streams 32 --steps 100
done
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg4 --steps 200
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg4 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg3 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config cfg3 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config f3 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --config f3 --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --io float --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 1,44100,48000,7 --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 1,44100,48000,7 --steps 300
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 4,44100,48000,7 --streams 32 --steps 60
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 2,48000,44100,7 --streams 32 --steps 100
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 2,48000,11025,7 --streams 32 --steps 20
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --custom 2,44100,16000,7 --streams 32 --steps 20
bash tools/gpu_ab.sh SPEEXHIP_LIB_PATH "$B $A" --frames 441000 --steps 300
} > $E1 2>&1
cat $E1
