#!/bin/bash
# tools/gpu_ab.sh -- diagnostics: A/B the period kernel's variants.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
for S in 32; do for WAVES in 16 8 4; do for SKIP in 16 24 48 56; do
  echo -n "streams=$S waves=$WAVES skip=$SKIP " >> $O/ab.txt
  SPEEXHIP_WAVES=$WAVES SPEEXHIP_SKIP=$SKIP timeout 200 python bench.py --steps 50 --warmup 5 --streams $S --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'])" >> $O/ab.txt
done; done; done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for W in 16 8; do
rm -rf $O/pmc_w$W
SPEEXHIP_WAVES=$W SPEEXHIP_SKIP=16 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $O/pmc_w$W -- python3 $R/bench.py --steps 10 --warmup 2 --streams 32 --no-cpu-baseline --no-parity > $O/pmc_w$W.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_w*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'resample_' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[4], {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
