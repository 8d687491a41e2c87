#!/bin/bash
# tools/r05_probe4.sh -- init trace; wide windows just under the planner's quarter-of-the-lanes rule (32k / 96k -> 11.025k)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05_probe4; mkdir -p $O
SPEEXHIP_INIT_TRACE=1 SPEEXHIP_POOL_TRACE=1 timeout 300 python tools/first_call.py > $O/first_call_trace.txt 2>&1
head -70 $O/first_call_trace.txt
for CH in 1 2; do for P in 96000,11025 32000,11025; do for MF in 4 8; do
  for S in 32 1; do
  SPEEXHIP_MIN_FILL=$MF python bench.py --custom $CH,$P,7 --streams $S --frames 131072 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('min_fill $MF ch $CH $P streams $S: launch_us', d['roofline']['launch_us'], 'valu', d['valu']['frac'], 'path', d['config']['fast_path'], 'parity', d.get('parity', {}).get('max_abs_diff_lsb'), d.get('parity', {}).get('mismatch_rate'))"
  done
done; done; done 2>&1 | tee $O/wide_windows.txt
