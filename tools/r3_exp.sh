#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
E1=$O/final_$(date +%H%M%S).txt
{
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-1200
} > $E1 2>&1
cat $E1
