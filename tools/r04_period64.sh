#!/bin/bash
# tools/r04_period64.sh -- round 4, on the GPU box: the fp64-accumulate period kernel: parity tests, then
# 44.1k -> 48k stereo q10 in the three modes at 1 and 32 streams.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout 2000 python -m pytest tests -m gpu -x -q -s -k "fp64 or window_layout or every_golden or float_entry or baseline_configs or many_rates or control_scripts_fast or soak or release_stream" > $O/pytest_period64.txt 2>&1
tail -8 $O/pytest_period64.txt
: > $O/bench_q10.jsonl
for mode in fast fast_f32 exact; do
  for S in 1 32; do
    timeout 300 python bench.py --custom 2,44100,48000,10 --mode $mode --streams $S --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_q10.jsonl
  done
done
python3 - <<PY
import json
for l in open("$O/bench_q10.jsonl"):
    d = json.loads(l)
    print(d["config"]["mode"], d["config"]["streams_per_gpu"], "launch_us", d["roofline"]["launch_us"], "valu", d["valu"]["frac"], d["valu"]["arithmetic"], "fast_path", d["config"]["fast_path"], "parity", d.get("parity", {}).get("max_abs_diff_lsb"), d.get("parity", {}).get("mismatch_rate"))
PY
