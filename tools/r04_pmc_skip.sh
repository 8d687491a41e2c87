#!/bin/bash
# tools/r04_pmc_skip.sh -- cache counters of one decimator launch with and without its staging phase (SPEEXHIP_SKIP=2):
# why the same FIR loop takes 2.7x longer behind a staged window when every CU is busy.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_skip; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export SPEEXHIP_PP=1 SPEEXHIP_KS_UNSPLIT=0 SPEEXHIP_SPLITS=1
G1="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM"
G2="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_READ_sum"
G3="SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
G4="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
G5="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA_RDREQ_32B_sum TCC_TAG_STALL_sum"
for SK in 0 2; do
  for G in 1 2 3 4 5; do
    eval "CS=\$G$G"
    SPEEXHIP_SKIP=$SK timeout 300 rocprofv3 --pmc $CS --output-format csv -d $O/skip${SK}_g$G -- python3 $R/bench.py --custom ${CASE:-2,48000,11025,7} --streams 32 --frames 131072 --steps 6 --warmup 2 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/skip${SK}_g$G.log 2>&1 || echo "pass $G skip $SK failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmc_skip'
table = collections.OrderedDict()
for d in sorted(glob.glob(O + '/*_g?')):
    if not os.path.isdir(d): continue
    case = os.path.basename(d)[:-3]
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'resample_' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            table.setdefault(k, {})[case] = sum(v) / len(v)
cases = sorted({c for v in table.values() for c in v})
with open(O + '/summary.txt', 'w') as out:
    out.write('%-28s' % 'counter (avg per launch)' + ''.join('%18s' % c for c in cases) + '\n')
    for k, v in table.items():
        out.write('%-28s' % k + ''.join('%18.0f' % v.get(c, float('nan')) for c in cases) + '\n')
print(open(O + '/summary.txt').read())
PY
