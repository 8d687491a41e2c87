#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
{
SPEEXHIP_NO_W16=1 python tools/stamps.py --streams 32 --launches 1 --custom 1,44100,16000,7
python tools/stamps.py --streams 32 --launches 1 --custom 1,44100,16000,7
SPEEXHIP_NO_W16=1 python tools/stamps.py --streams 32 --launches 1 --custom 2,48000,11025,7
python tools/stamps.py --streams 32 --launches 1 --custom 2,48000,11025,7
} > $E1 2>&1
cat $E1
