#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_after_image_removal.txt 2>&1; tail -5 $O/tests_after_image_removal.txt
timeout 600 python tools/fuzz_gpu.py --seconds 150 --seed 4101 --batch 2>&1 | tail -2
timeout 600 python tools/fuzz_gpu.py --seconds 100 --seed 4102 --many-channels 2>&1 | tail -2
