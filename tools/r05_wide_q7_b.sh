# tools/r05_wide_q7_b.sh -- round 5: bank padding of the wide int16 windows (SPEEXHIP_PAD sweep, FIR phase only and
# whole kernel) and SQ counters of the 32k -> 11.025k launches (gpurun).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { local label="$1"; shift
  env "$@" python bench.py --custom $CH,$P,7 --streams 32 --frames 131072 --steps 8 --warmup 3 --reps 2 --preheat-ms 50 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('ch $CH $P %-28s %.1f us' % ('$label', d['roofline']['launch_us']))"
}
for P in 32000,11025 48000,11025; do for CH in 1 2 4; do
for PAD in 0 2 4 6 8 10 16 24 32 34; do
run "PAD=$PAD" SPEEXHIP_PAD=$PAD
run "PAD=$PAD FIR only" SPEEXHIP_PAD=$PAD SPEEXHIP_SKIP=10
done; done; done
O=$R/gpurun_out/r05_wide_pmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
G1="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
for CH in 1 4; do for G in 1 2 3; do eval "CS=\$G$G"
  SPEEXHIP_SKIP=10 timeout 300 rocprofv3 --pmc $CS --output-format csv -d $O/ch${CH}_g$G -- python3 $R/bench.py --custom $CH,32000,11025,7 --streams 32 --frames 131072 --steps 6 --warmup 2 --reps 1 --preheat-ms 0 --no-cpu-baseline --no-parity > $O/ch${CH}_g$G.log 2>&1 || echo "ch $CH pass $G failed"
done; done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/r05_wide_pmc'
for ch in (1, 4):
    print('32k -> 11.025k q7, %d channel(s), 32 streams x 131072 frames, FIR phase only: SQ counters per launch' % ch)
    for d in sorted(glob.glob(O + '/ch%d_g?' % ch)):
        for f in glob.glob(d + '/*/*counter_collection.csv'):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'resample_' in r['Kernel_Name']:
                    acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
            for k, v in acc.items():
                print('  %-60s %-24s %16.0f' % (k[0], k[1], sum(v) / len(v)))
PY
