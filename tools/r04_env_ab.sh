#!/bin/bash
# tools/r04_env_ab.sh -- same-box A/B of runtime knobs on the BENCH launch (kernel arguments in device or host memory)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
for rep in 1 2; do
for K in unset 0 1; do
  if [ $K = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$K; fi
  for S in 1 32; do
  python bench.py --no-cpu-baseline --no-parity --steps 300 --warmup 20 --streams $S 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HIP_FORCE_DEV_KERNARG=$K streams $S launch_us', d['roofline']['launch_us'], d['roofline']['launch_us_min'], 'ms_per_step', d['ms_per_step'], 'host_issue', d['timing']['host_issue_us_per_step'], 'box', d['box'])"
  done
done
done 2>&1 | tee $O/env_ab.txt
