#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
{
SPEEXHIP_PRIO=3 python tools/stamps.py --streams 2 --launches 1
SPEEXHIP_PRIO=3 python tools/stamps.py --streams 4 --launches 1
SPEEXHIP_PRIO=0 python tools/stamps.py --streams 32 --launches 1
SPEEXHIP_PRIO=3 SPEEXHIP_SKIP=10 python tools/stamps.py --streams 32 --launches 1
} > $E1 2>&1
cat $E1
