#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "layouts_without_an_isa or window_layout or int16_window or golden or many_rates" > $O/r06_pytest_w16g.txt 2>&1; echo "rc=$?" >> $O/r06_pytest_w16g.txt; tail -12 $O/r06_pytest_w16g.txt
PAIRS="48000,11025 44100,16000 96000,11025 44100,48000" CHANNELS="9 11 13 14 15 17 20 24" Q=7 REPS=2 bash tools/perf_sweep.sh 2>/dev/null > $O/r06_sweep_frames_w16g.txt
sort -t'|' -k4 $O/r06_sweep_frames_w16g.txt
timeout 300 python tools/fuzz_gpu.py --seconds 120 --seed 611 --many-channels > $O/r06_fuzz_c.txt 2>&1; tail -3 $O/r06_fuzz_c.txt
