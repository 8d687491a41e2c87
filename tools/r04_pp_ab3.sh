#!/bin/bash
# tools/r04_pp_ab3.sh -- stereo and three channels: the planner's rule (phase pairs for wide windows) against the other lanes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
row() {  # label env custom streams frames
  env $2 SPEEXHIP_PLAN_VERBOSE=1 python bench.py --custom $3 --streams $4 --frames $5 --steps 10 --warmup 3 --reps 2 --preheat-ms 60 --no-cpu-baseline 2> $O/plan.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-10s %-18s S=%-3s F=%-8s %8.1f us  valu %.3f  parity %s %s' % ('$1', '$3', '$4', '$5', d['roofline']['launch_us'], d['valu']['frac'], d.get('parity', {}).get('max_abs_diff_lsb'), d.get('parity', {}).get('mismatch_rate')))"
  grep -h "period launch" $O/plan.err | tail -1 | sed 's/^/      /'
}
for C in 2,48000,11025,7 2,48000,22050,7 2,44100,32000,7 2,44100,8000,7 2,44100,16000,7 2,32000,44100,7 3,48000,11025,7 3,44100,32000,7 3,44100,16000,7 3,48000,22050,7; do
  for SF in "32 131072" "32 1048576" "8 131072" "1 1048576" "1 48000"; do
    set -- $SF
    row other SPEEXHIP_PP=0 $C $1 $2
    row rule SPEEXHIP_PPX=1 $C $1 $2
  done
done 2>&1 | tee $O/pp_ab3.txt
