#!/bin/bash
# tools/r05_pmc_phases.sh -- VERDICT r4 #2 / #3: which phase of the period kernel owns the LDS bank-conflict cycles
# (10 % of the LDS-active cycles on BASELINE configs[4]'s share, 14 % on configs[3]) and the vector instructions that
# are not FMAs.  SQ counters of the 32-stream launches of cfg2 and cfg4 with parts of the kernel skipped
# (SPEEXHIP_SKIP bits: 2 = window staging, 4 = FIR loop, 8 = stores): everything / staging only / FIR only / stores only.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_pmc_phases; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
G1="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
for CFG in cfg2 cfg4; do
  for SK in 0 12 10 6; do
    for G in 1 2 3; do
      eval "CS=\$G$G"
      SPEEXHIP_SKIP=$SK timeout 300 rocprofv3 --pmc $CS --output-format csv -d $O/${CFG}_skip${SK}_g$G -- python3 $R/bench.py --config $CFG --streams 32 --steps 6 --warmup 2 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/${CFG}_skip${SK}_g$G.log 2>&1 || echo "$CFG pass $G skip $SK failed"
    done
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/r05_pmc_phases'
names = {'0': 'everything', '12': 'staging only', '10': 'FIR only', '6': 'stores only'}
with open(O + '/summary.txt', 'w') as out:
    for cfg in ('cfg2', 'cfg4'):
        table = collections.OrderedDict()
        for d in sorted(glob.glob(O + '/%s_skip*_g?' % cfg)):
            if not os.path.isdir(d): continue
            sk = os.path.basename(d).split('_')[1][4:]
            for f in glob.glob(d + '/*/*counter_collection.csv'):
                acc = collections.defaultdict(list)
                for r in csv.DictReader(open(f)):
                    if 'resample_' in r['Kernel_Name']:
                        acc[r['Counter_Name']].append(float(r['Counter_Value']))
                for k, v in acc.items():
                    table.setdefault(k, {})[sk] = sum(v) / len(v)
        out.write('%s, 32 streams x 2^20 frames: SQ counters per launch\n' % cfg)
        out.write('%-24s' % 'counter' + ''.join('%16s' % names[c] for c in ('0', '12', '10', '6')) + '\n')
        for k, v in table.items():
            out.write('%-24s' % k + ''.join('%16.0f' % v.get(c, float('nan')) for c in ('0', '12', '10', '6')) + '\n')
        out.write('\n')
print(open(O + '/summary.txt').read())
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
