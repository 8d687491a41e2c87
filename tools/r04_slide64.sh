#!/bin/bash
# tools/r04_slide64.sh -- round 4, on the GPU box: the fp64-accumulate slide kernel: parity tests, then BASELINE
# configs[2] in the three modes at 1 and 32 streams (bench lines with parity blocks).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
./tools/probe_stream_gone > $O/probe_stream_gone.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "fp64 or small_ratio or n_to_one or workgroups_shrink or destroy or every_golden or float_entry or baseline_configs" > $O/pytest_slide64.txt 2>&1
tail -8 $O/pytest_slide64.txt
: > $O/bench_cfg3.jsonl
for mode in fast fast_f32 exact; do
  for S in 1 32; do
    timeout 300 python bench.py --config cfg3 --mode $mode --streams $S --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl
  done
done
python3 - <<PY
import json
for l in open("$O/bench_cfg3.jsonl"):
    d = json.loads(l)
    print(d["config"]["mode"], d["config"]["streams_per_gpu"], "launch_us", d["roofline"]["launch_us"], "valu", d["valu"]["frac"], d["valu"]["arithmetic"], "acc", d["config"]["accumulate"][:20], "parity", d.get("parity", {}).get("max_abs_diff_lsb"), d.get("parity", {}).get("mismatch_rate"))
PY
cat $O/probe_stream_gone.txt
