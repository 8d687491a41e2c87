#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/fuzz2_$(date +%H%M%S).txt
{
python tools/fuzz_gpu.py --seconds 300 --seed 3101 --batch 2>&1 | tail -6
SPEEXHIP_FORCE_W16=1 python tools/fuzz_gpu.py --seconds 150 --seed 3102 2>&1 | tail -6
python tools/fuzz_gpu.py --seconds 150 --seed 3103 --max-frames 1200000 --batch 2>&1 | tail -6
} > $E1 2>&1
cat $E1
