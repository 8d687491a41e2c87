#!/bin/bash
# tools/gpu_check.sh [quick|full] -- the standard on-GPU sequence (run through gpurun):
# smoke, GPU parity tests, the Node harness, bench lines, and a rocprofv3 kernel trace of the bench.
# Outputs land in gpurun_out/ (scratch); summaries worth keeping are copied to profiles/ by hand.
MODE=${1:-quick}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$MODE" = "full" ]; then
  timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
  timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
else
  timeout 900 python -m pytest tests -m gpu -q -x -k "golden or baseline or batched or configs4 or node" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
fi
timeout 300 python bench.py > $O/bench1.txt 2>&1; echo "rc=$?" >> $O/bench1.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench1_driver_form.txt 2>&1
timeout 300 python bench.py --steps 100 --warmup 20 --streams 32 --no-cpu-baseline > $O/bench_s32.txt 2>&1
timeout 300 python bench.py --steps 100 --warmup 20 --total-streams 32 --no-cpu-baseline > $O/bench_total32.txt 2>&1
timeout 300 python bench.py --gpus 2 > $O/bench_gpus2_on_1gpu_box.txt 2>&1; echo "rc=$? (must be non-zero on a 1-GPU box, with no JSON line)" >> $O/bench_gpus2_on_1gpu_box.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/prof1.log 2>&1
cd $R
tail -3 $O/pytest_gpu.txt
cat $O/bench_gpus2_on_1gpu_box.txt | tail -4
grep -h metric $O/bench1.txt $O/bench1_driver_form.txt $O/bench_s32.txt $O/bench_total32.txt | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('streams', d['config']['streams_per_gpu'], d['scaling'], 'steps', d['steps'], 'value', d['value'], 'ms/step', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], 'launch_us', d['roofline']['launch_us'], 'hbm_frac', d['roofline']['frac'], 'valu_frac', d['valu']['frac'], 'parity', d.get('parity'))
"
grep -h resample $O/prof1/*/*kernel_stats.csv | cut -d, -f1-8 | cut -c1-60,150-400
