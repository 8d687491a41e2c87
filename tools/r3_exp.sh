#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
{
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} > $E1 2>&1
cat $E1
