#!/bin/bash
# tools/r04_pmc_decim.sh -- SQ counters of small decimator launches (32 streams x 131072 frames) beside a ratio that runs
# three times closer to the vector peak at the same size: what the FIR loop waits for.  Counter passes only (no trace
# domains), one group of counters per pass; an unknown counter fails its pass, the others still run.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_decim; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
G3="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ"
G4="SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU"
G5="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_IFETCH SQ_WAVE32_INSTS"
G6="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_INPUT_VALID_READYB SQC_DCACHE_ATOMIC"
for C in ${CASES:-1,48000,11025,7 2,48000,11025,7 3,48000,11025,7 2,48000,44100,7}; do
  T=$(echo $C | tr , _)
  for G in ${GROUPS_TO_RUN:-1 2 3 4 5 6}; do
    eval "CS=\$G$G"
    timeout 300 rocprofv3 --pmc $CS --output-format csv -d $O/${T}_g$G -- python3 $R/bench.py --custom $C --streams 32 --frames 131072 --steps 6 --warmup 2 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/${T}_g$G.log 2>&1 || echo "pass $G of $C failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmc_decim'
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
