#!/bin/bash
# tools/gpu_profile.sh [ROUND] -- collect the evidence kept under profiles/ (run through gpurun; ~12 min):
# for every BASELINE config (cfg2, cfg3, cfg4 + SURVEY F3) at 1 and 32 streams
#   * rocprofv3 --kernel-trace --stats of the bench command          -> r0N_kernel_stats_<cfg>_s<S>.csv
#   * HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md; never
#     combined with trace domains)                                      -> pmc_traffic.json, r0N_pmc_summary.json
#   * SQ counters for every config at both sizes; kernel stats for the EXACT-mode and float-I/O lines too
# and, AFTER the PMC passes (bench.py reads profiles/pmc_traffic.json), the bench lines of every
# workload (parity block in each; cpu_baseline in the 1-stream ones)   -> r0N_bench_lines.jsonl
# Results land in gpurun_out/prof_r0N/; copy the summaries into profiles/ by hand.
N=${1:-06}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_r$N; [ -z "$LINES_ONLY" ] && rm -rf $O; mkdir -p $O; cd $R
# LINES_ONLY=1: skip the rocprofv3 passes (a second lease that only re-collects the bench lines and the host-side tools on
# the traffic file of the first)
if [ -z "$LINES_ONLY" ]; then
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg3 cfg4 f3; do
  for S in 1 32; do
    ST=200; [ $S = 32 ] && ST=60
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${CFG}_s$S -- python3 $R/bench.py --config $CFG --streams $S --steps $ST --warmup 20 --reps 3 --no-cpu-baseline --no-parity > $O/trace_${CFG}_s$S.log 2>&1
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_${CFG}_s$S -- python3 $R/bench.py --config $CFG --streams $S --steps 12 --warmup 3 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/pmc_${C}_${CFG}_s$S.log 2>&1
    done
  done
done
# SQ counters for every config at both sizes (round 3: cfg3 / cfg4 / F3 too), two passes of <= 8 counters
for CFG in cfg2 cfg3 cfg4 f3; do
  for S in 1 32; do
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/pmc_sq_${CFG}_s$S -- python3 $R/bench.py --config $CFG --streams $S --steps 12 --warmup 3 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/pmc_sq_${CFG}_s$S.log 2>&1
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq2_${CFG}_s$S -- python3 $R/bench.py --config $CFG --streams $S --steps 12 --warmup 3 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/pmc_sq2_${CFG}_s$S.log 2>&1
  done
done
# kernel stats of the lines that had none in round 2: EXACT mode (cfg2, cfg3) and float I/O
for S in 1 32; do
  ST=100; [ $S = 32 ] && ST=20
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3exact_s$S -- python3 $R/bench.py --config cfg3 --mode exact --streams $S --steps $ST --warmup 5 --reps 3 --no-cpu-baseline --no-parity > $O/trace_cfg3exact_s$S.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2float_s$S -- python3 $R/bench.py --io float --streams $S --steps $ST --warmup 5 --reps 3 --no-cpu-baseline --no-parity > $O/trace_cfg2float_s$S.log 2>&1
done
# round 4: configs[2] on the fp32 chain (MODE_FAST_F32) beside its fp64-accumulate default, and the period kernel's fp64 instances
for S in 1 32; do
  ST=100; [ $S = 32 ] && ST=20
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3f32chain_s$S -- python3 $R/bench.py --config cfg3 --mode fast_f32 --streams $S --steps $ST --warmup 5 --reps 3 --no-cpu-baseline --no-parity > $O/trace_cfg3f32chain_s$S.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_q10period_s$S -- python3 $R/bench.py --custom 2,44100,48000,10 --streams $S --steps $ST --warmup 5 --reps 3 --no-cpu-baseline --no-parity > $O/trace_q10period_s$S.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2exact_s1 -- python3 $R/bench.py --mode exact --steps 100 --warmup 5 --reps 3 --no-cpu-baseline --no-parity > $O/trace_cfg2exact_s1.log 2>&1
cd $R
python3 - "$N" <<'PY'
import csv, glob, json, collections, os, sys, shutil
N = sys.argv[1]
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
O = R + '/gpurun_out/prof_r' + N
summary, traffic = {}, {"_note": "HBM bytes per launch of the dominant kernel from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB per dispatch averaged over the run's dispatches of the resample_* kernel; see r%s_pmc_summary.json). FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of 16 B/lane coalesced reads); WRITE_SIZE as is. Collected by tools/gpu_profile.sh with the kernels of the commit the profiles were committed with." % N}
for d in sorted(glob.glob(O + '/pmc_*')):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'resample_' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    summary[os.path.basename(d)] = {k: {'avg_per_dispatch': sum(v) / len(v), 'dispatches': len(v)} for k, v in acc.items()}
for d in sorted(glob.glob(O + '/trace_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/*/*kernel_stats.csv'):
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if 'resample_' in r['Name']]
        for r in keep:
            summary[os.path.basename(d)] = {'kernel': r['Name'][:110], 'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']), 'min_ns': int(r['MinNs']), 'max_ns': int(r['MaxNs'])}
        # the stats file as rocprofv3 wrote it, kernels of this library first
        with open(O + '/r%s_kernel_stats_%s.csv' % (N, os.path.basename(d)[6:]), 'w') as out:
            w = csv.DictWriter(out, fieldnames=rows[0].keys())
            w.writeheader()
            for r in keep + [r for r in rows if r not in keep][:6]:
                w.writerow(r)
for cfg in ('cfg2', 'cfg3', 'cfg4', 'f3'):
    for s in (1, 32):
        f = summary.get('pmc_FETCH_SIZE_%s_s%d' % (cfg, s), {}).get('FETCH_SIZE')
        w = summary.get('pmc_WRITE_SIZE_%s_s%d' % (cfg, s), {}).get('WRITE_SIZE')
        if f and w:
            traffic['%s_s%d_fast_fixed' % (cfg, s)] = int(round(2 * f['avg_per_dispatch'] * 1024 + w['avg_per_dispatch'] * 1024))
json.dump(summary, open(O + '/r%s_pmc_summary.json' % N, 'w'), indent=1)
json.dump(traffic, open(O + '/pmc_traffic.json', 'w'), indent=1)
shutil.copy(O + '/pmc_traffic.json', R + '/profiles/pmc_traffic.json')   # bench.py reads it from here
print(json.dumps(traffic, indent=1))
PY
fi  # LINES_ONLY
cd $R
# bench lines AFTER the PMC passes: every line carries the traffic of this very collection
: > $O/r${N}_bench_lines.jsonl
python bench.py >> $O/r${N}_bench_lines.jsonl 2> $O/bench_default.err
python bench.py --streams 32 --steps 100 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
python bench.py --total-streams 32 --steps 100 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
for CFG in cfg3 cfg4 f3; do
  python bench.py --config $CFG --steps 300 >> $O/r${N}_bench_lines.jsonl 2>/dev/null
  python bench.py --config $CFG --streams 32 --steps 60 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
done
python bench.py --mode exact --steps 200 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
python bench.py --config cfg3 --mode exact --steps 100 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
python bench.py --config cfg3 --mode exact --streams 32 --steps 20 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
for S in 1 32; do
  python bench.py --config cfg3 --mode fast_f32 --streams $S --steps 60 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
  python bench.py --custom 2,44100,48000,10 --streams $S --steps 60 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
done
python bench.py --io float --steps 300 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
python bench.py --io float --streams 32 --steps 60 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
# mono (not a BASELINE config; the store path of round 3): 44.1k->48k q7, int16 and float
for IO in int16 float; do
  python bench.py --custom 1,44100,48000,7 --io $IO --steps 300 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
  python bench.py --custom 1,44100,48000,7 --io $IO --streams 32 --steps 60 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
done
# round 6: the default mode is the pinned summation order (fast_fixed); the opt-in mode with tap-range shares (fast) beside
# it at the BASELINE configs; configs[4] whole on one GPU with its CPU column on min(cores, 256) worker processes
for CFG in cfg2 cfg3 cfg4; do
  for S in 1 32; do
    python bench.py --config $CFG --mode fast --streams $S --steps 60 --no-cpu-baseline >> $O/r${N}_bench_lines.jsonl 2>/dev/null
  done
done
SPEEXHIP_PY_NO_TORCH=1 python tools/pinned_path_bench.py > $O/r${N}_pinned_path.json 2>/dev/null   # (the runtime a Node / C caller loads)
python bench.py --total-streams 256 --steps 10 --warmup 2 > $O/r${N}_bench_total256.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_form.json 2>/dev/null
python tools/call_stamps.py > $O/r${N}_call_stamps.txt 2>/dev/null
python tools/first_call.py > $O/r${N}_first_call.txt 2>/dev/null
FIRST_CALL_WARMUP=1 python tools/first_call.py > $O/r${N}_first_call_warm.txt 2>/dev/null
node --expose-gc tools/steady.js > $O/r${N}_steady.txt 2>/dev/null
python tools/host_path_bench.py > $O/r${N}_host_path.json 2>/dev/null
python tools/small_call_latency.py > $O/r${N}_small_call_latency.txt 2>/dev/null
python tools/init_cost.py > $O/r${N}_init_cost.txt 2>/dev/null
(cd node-speex-resampler_amd && node test/bench.js $O/r${N}_node_bench.json > /dev/null 2>&1)
python3 - "$N" <<'PY'
import json, os, sys
N = sys.argv[1]
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/prof_r' + N
for l in open(O + '/r%s_bench_lines.jsonl' % N):
    d = json.loads(l)
    r = d['roofline']
    print(d['config']['workload'][:58], '| S', d['config']['streams_per_gpu'], d['config']['mode'], d['config']['io'], '| value', d['value'], '| launch_us', r['launch_us'], '| hbm', r['frac'], 'read-only', r['read_only_frac'], '| valu', d['valu']['frac'], '| traffic/alg', round(r['traffic'] / r['algorithmic_bytes_per_launch'], 3) if r['traffic'] else None, '| parity', d.get('parity', {}).get('max_abs_diff_lsb', d.get('parity', {}).get('max_abs_diff')), '| cpu', d.get('cpu_baseline', {}).get('value'))
PY
