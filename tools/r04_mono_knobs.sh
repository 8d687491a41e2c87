#!/bin/bash
# tools/r04_mono_knobs.sh -- the sweep's lowest rows (mono / 3-channel decimators, 32 streams x 131072 frames) under
# the planner's diagnostic knobs: which launch shape would be faster than the one the rules pick?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
run() {  # label, env..., -- custom
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" SPEEXHIP_PLAN_VERBOSE=1 python bench.py --custom $1 --streams 32 --frames 131072 --steps 8 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline --no-parity 2> $O/plan.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-28s %-18s %8.1f us  valu %.3f' % ('$label', '$1', d['roofline']['launch_us'], d['valu']['frac']))"
  grep -h "period launch" $O/plan.err | tail -1 | sed 's/^/      /'
}
for C in 1,48000,11025,7 1,48000,22050,7 1,44100,32000,7 1,44100,8000,7 1,44100,16000,7 3,48000,11025,7 2,48000,11025,7 1,32000,44100,7; do
  run default -- $C
  run splits1 SPEEXHIP_SPLITS=1 -- $C
  run splits2 SPEEXHIP_SPLITS=2 -- $C
  run splits4 SPEEXHIP_SPLITS=4 -- $C
  run ksplit0 SPEEXHIP_KSPLIT=0 -- $C
  run ksplit2 SPEEXHIP_KSPLIT=2 -- $C
  run ksplit4 SPEEXHIP_KSPLIT=4 -- $C
  run no_w16 SPEEXHIP_NO_W16=1 -- $C
  run force_w16 SPEEXHIP_W16_ALWAYS=1 -- $C
  run r5 SPEEXHIP_R=5 -- $C
  run r5_no_w16 SPEEXHIP_R=5 SPEEXHIP_NO_W16=1 -- $C
  run full_tile SPEEXHIP_FULL_TILE=1 -- $C
done 2>&1 | tee $O/mono_knobs.txt
