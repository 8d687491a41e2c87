#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3; mkdir -p $O
{ python tools/gpu_scratch.py; SPEEXHIP_ZERO_COPY_BELOW=1000000000 python tools/gpu_scratch.py; SPEEXHIP_ZERO_COPY_BELOW=0 python tools/gpu_scratch.py; } 2>&1 | grep -v amdgpu.ids > $O/zc_sweep.txt
cat $O/zc_sweep.txt
