#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
{
bash tools/gpu_ab.sh SPEEXHIP_SKIP "0 2 8 10 6 12" --io float --streams 32 --steps 40
bash tools/gpu_ab.sh SPEEXHIP_SKIP "0 2 8 10 6 12" --streams 32 --steps 40
} > $E1 2>&1
cat $E1
