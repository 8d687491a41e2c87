#!/bin/bash
# the slide kernel's phases by skip mask (diagnostics build): what its staging and stores cost
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r06_slide_bound.txt
for C in "--config f3" "--custom 2,24000,48000,5" "--custom 2,48000,24000,5" "--custom 1,16000,48000,7" "--custom 2,48000,16000,7" "--config cfg3 --mode fast_f32"; do
tools/ab.sh -o $O/r06_slide_bound.txt -f "'launch_us %s (min %s)  valu %s' % (d['roofline']['launch_us'], d['roofline']['launch_us_min'], d['valu']['frac'])" -- "" "SPEEXHIP_SKIP=2" "SPEEXHIP_SKIP=8" "SPEEXHIP_SKIP=10" -- python bench.py $C --streams 32 --steps 40 --warmup 5 --reps 3 --no-cpu-baseline --no-parity
done
