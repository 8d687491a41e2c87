#!/bin/bash
# tools/gpu_profile.sh -- collect the evidence kept under profiles/: rocprofv3 kernel-trace
# stats of the bench command, HBM traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes, as MI355X_MICROARCH.md prescribes; never combined with trace domains), bench lines of
# every BASELINE config.  Run through gpurun; results land in gpurun_out/prof_r01/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_r01; rm -rf $O; mkdir -p $O; cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 32 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_s32.json 2>/dev/null
python bench.py --mode exact --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_s1_exact.json
python bench.py --io float --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_s1_float.json 2>/dev/null
python bench.py --io float --streams 32 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_s32_float.json 2>/dev/null 2>/dev/null
for CFG in cfg3 cfg4 f3; do
  python bench.py --config $CFG --steps 500 --warmup 20 > $O/bench_${CFG}_s1.json 2>/dev/null
  python bench.py --config $CFG --streams 32 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_${CFG}_s32.json 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
for S in 1 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s$S -- python3 $R/bench.py --streams $S --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/trace_s$S.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_s$S -- python3 $R/bench.py --streams $S --steps 20 --warmup 3 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/pmc_${C}_s$S.log 2>&1
  done
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/pmc_sq_s32 -- python3 $R/bench.py --streams 32 --steps 20 --warmup 3 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/pmc_sq_s32.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq2_s32 -- python3 $R/bench.py --streams 32 --steps 20 --warmup 3 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/pmc_sq2_s32.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/prof_r01'
summary = {}
for d in sorted(glob.glob(O + '/pmc_*')):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'resample_' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    summary[os.path.basename(d)] = {k: {'avg_per_dispatch': sum(v) / len(v), 'dispatches': len(v)} for k, v in acc.items()}
for d in sorted(glob.glob(O + '/trace_s*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/*/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            if 'resample_' in r['Name']:
                summary[os.path.basename(d)] = {'kernel': r['Name'][:90], 'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']), 'min_ns': int(r['MinNs']), 'max_ns': int(r['MaxNs'])}
json.dump(summary, open(O + '/summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1)[:3000])
PY
for f in $O/bench_*.json; do echo "== $f"; python3 -c "
import sys, json
d = json.loads(open('$f').readline())
print(d['config']['workload'][:70], '| value', d['value'], d['unit'], '| launch_us', d['roofline']['launch_us'], '| hbm', d['roofline']['frac'], '| valu', d['valu']['frac'], '| fast_path', d['config']['fast_path'], '| parity', d.get('parity', {}).get('max_abs_diff_lsb'), d.get('parity', {}).get('mismatch_rate'), '| cpu', d.get('cpu_baseline', {}).get('value'))
"; done
