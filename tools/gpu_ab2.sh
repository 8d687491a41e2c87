#!/bin/bash
# tools/gpu_ab2.sh -- diagnostics: rocprofv3 kernel durations of the single-stream launch with parts
# of the kernel skipped (SPEEXHIP_SKIP bits: 2 = window staging, 4 = FIR loop, 8 = stores).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for SKIP in 64 130 128 14; do
  export SPEEXHIP_SKIP=$SKIP
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$SKIP -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/t$SKIP.log 2>&1
  f=$(ls $O/t$SKIP/*/*kernel_stats.csv | head -1)
  echo "skip=$SKIP $(grep resample_ $f | cut -d, -f1-7 | cut -c1-40,120-)" >> $O/summary.txt
  grep launch_us $O/t$SKIP.log | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   bench launch_us', d['roofline']['launch_us'])" >> $O/summary.txt
done
cat $O/summary.txt
cd $O; for s in 64 130 128 14; do f=$(ls t$s/*/*kernel_stats.csv|head -1); python3 - "$f" $s <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'resample_' in r['Name']: print('skip',sys.argv[2],'calls',r['Calls'],'avg_ns',r['AverageNs'],'min',r['MinNs'],'max',r['MaxNs'])
PY
done
