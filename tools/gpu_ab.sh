#!/bin/bash
# tools/gpu_ab.sh -- diagnostics: A/B the persistent period kernel's variants (32 streams).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
for CFG in "0 2" "1 1"; do set -- $CFG; for SKIP in 0 8 10; do
  echo -n "wide=$1 wg_per_cu=$2 skip=$SKIP " >> $O/ab.txt
  SPEEXHIP_WIDE=$1 SPEEXHIP_WG_PER_CU=$2 SPEEXHIP_SKIP=$SKIP timeout 200 python bench.py --steps 50 --warmup 5 --streams 32 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'])" >> $O/ab.txt
done; done
cat $O/ab.txt
