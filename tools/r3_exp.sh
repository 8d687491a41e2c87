#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
{
SPEEXHIP_KSPLIT=0 python tools/stamps.py --streams 1 --launches 1 2>&1 | grep "phase\|last stamp\|share\|clock"
SPEEXHIP_KSPLIT=1 python tools/stamps.py --streams 1 --launches 1 2>&1 | grep "phase\|last stamp\|share\|clock\|start  \|stores issued"
} > $E1 2>&1
cat $E1
