#!/bin/bash
# round 6: bounds and sweeps -- cfg4 with its staging / stores skipped (the most a staging rewrite could buy), the headline
# launch under R = 5 / 10, frames of 8-24 channels on a few ratios (C++ loop layouts against the ISA-loop ones)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r06_cfg4_bound.txt $O/r06_headline_r.txt
for S in 32 1; do
tools/ab.sh -o $O/r06_cfg4_bound.txt -f "'launch_us %s (min %s, max %s)  valu %s' % (d['roofline']['launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], d['valu']['frac'])" -- "" "SPEEXHIP_SKIP=2" "SPEEXHIP_SKIP=8" "SPEEXHIP_SKIP=10" "" "SPEEXHIP_SKIP=2" -- python bench.py --config cfg4 --streams $S --steps 40 --warmup 5 --reps 3 --mode fast_fixed --no-cpu-baseline --no-parity
done
tools/ab.sh -o $O/r06_headline_r.txt -f "'launch_us %s (min %s, max %s)  valu %s' % (d['roofline']['launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], d['valu']['frac'])" -- "" "SPEEXHIP_R=10" "SPEEXHIP_R=5" "SPEEXHIP_SPLITS=1" "SPEEXHIP_SPLITS=2" "SPEEXHIP_SPLITS=4" "SPEEXHIP_TILE_PERIODS=56" "SPEEXHIP_TILE_PERIODS=48" "SPEEXHIP_TILE_PERIODS=32" "" -- python bench.py --steps 200 --warmup 20 --reps 5 --no-cpu-baseline --no-parity
PAIRS="44100,48000 48000,44100 48000,11025 44100,16000 24000,48000 48000,8000" CHANNELS="8 9 10 11 12 13 14 15 16 17 20 24" Q=7 REPS=2 bash tools/perf_sweep.sh 2>/dev/null > $O/r06_sweep_frames.txt
sort -t'|' -k4 $O/r06_sweep_frames.txt | head -80
