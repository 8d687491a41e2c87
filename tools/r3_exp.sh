#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/exp_$(date +%H%M%S).txt
{
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "overflow or control_calls" 2>&1 | tail -15
timeout 300 tools/ubench_ctl > $O/ubench_ctl.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
} > $E1 2>&1
cat $E1
