#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
python tools/stamps.py --streams 1 --launches 2 > $O/stamps_cfg2_s1.txt 2>&1
python tools/stamps.py --streams 32 --launches 2 > $O/stamps_cfg2_s32.txt 2>&1
python tools/stamps.py --streams 32 --launches 1 --config cfg4 > $O/stamps_cfg4_s32.txt 2>&1
python tools/stamps.py --streams 2 --launches 1 > $O/stamps_cfg2_s2.txt 2>&1
bash tools/perf_sweep.sh > $O/perf_sweep.txt 2>&1
tail -5 $O/perf_sweep.txt
