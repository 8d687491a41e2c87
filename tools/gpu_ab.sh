#!/bin/bash
# tools/gpu_ab.sh VAR "v1 v2 ..." [bench args...] -- same-box A/B of one environment knob of the
# library (SPEEXHIP_*): runs bench.py once per value inside ONE gpurun call and prints launch_us.
VAR=$1; VALS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in $VALS; do
  env $VAR=$V python bench.py --no-cpu-baseline --reps 3 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('$VAR=$V', sys.argv[1:], 'launch_us', d['roofline']['launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], 'valu', d['valu']['frac'], 'parity', d.get('parity', {}).get('max_abs_diff_lsb'))
" "$@"
done
