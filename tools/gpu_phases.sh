#!/bin/bash
# tools/gpu_phases.sh -- rocprofv3 kernel durations of the single-stream launch with parts of the
# period kernel skipped (SPEEXHIP_SKIP bits: 2 = window staging, 4 = FIR loop, 8 = stores,
# 64 = return at once, 128 = return after staging).  Differences between rows are the serial phase
# costs of a launch that is one generation of workgroups.  bench.py preheats, so these are
# sustained-clock durations.  The mask exists only in the DIAGNOSTICS build of the library (csrc/diag.h).
R=${GRAFT_REPO_ROOT:-/root/repo}; export SPEEXHIP_LIB_PATH=$R/node-speex-resampler_amd/ab/libspeexhip_diag.so; O=$R/gpurun_out/phases; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for SKIP in 0 64 130 128 14 12 6 10; do
  export SPEEXHIP_SKIP=$SKIP
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$SKIP -- python3 $R/bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-parity > $O/t$SKIP.log 2>&1
done
cd $O; for s in 0 64 130 128 14 12 6 10; do f=$(ls t$s/*/*kernel_stats.csv | head -1); python3 - "$f" $s <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'resample_' in r['Name']:
        print('skip', sys.argv[2], 'calls', r['Calls'], 'avg_ns', round(float(r['AverageNs']), 1), 'min_ns', r['MinNs'])
PY
done | tee $O/summary.txt
