#!/bin/bash
# tools/gpu_ablate.sh -- diagnostics: time the fast kernel with phases left out (SPEEXHIP_SKIP
# bits: 1 rows staging (tiled only), 2 window staging, 4 FIR loop, 8 stores) and collect PMC counters.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ablate.txt
for S in 1 32; do
  for SKIP in 0 2 4 8 12 14; do
    echo -n "streams=$S skip=$SKIP " >> $O/ablate.txt
    SPEEXHIP_SKIP=$SKIP timeout 200 python bench.py --steps 50 --warmup 5 --streams $S --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'])" >> $O/ablate.txt
  done
done
cat $O/ablate.txt
cd /tmp && export TMPDIR=/tmp
for PASS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAVES_EQ_64 SQ_LEVEL_WAVES"; do
  TAG=$(echo $PASS | cut -d' ' -f1)
  rm -rf $O/pmc_$TAG
  timeout 300 rocprofv3 --pmc $PASS --output-format csv -d $O/pmc_$TAG -- python3 $R/bench.py --steps 10 --warmup 2 --streams 32 --no-cpu-baseline --no-parity > $O/pmc_$TAG.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'resample_' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(k, 'avg per dispatch %.4g' % (sum(v) / len(v)), 'n', len(v))
PY
