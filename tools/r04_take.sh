#!/bin/bash
# tools/r04_take.sh -- round 4, on the GPU box: the owned-block host calls: tests, then timings
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "owned_block or node_drop_in or host_buffer_call or python_mirror or chunk_coalescing or states_recycle" > $O/pytest_take.txt 2>&1
tail -8 $O/pytest_take.txt
for cfg in cfg2 cfg3 cfg4 f3; do
  python bench.py --config $cfg --steps 50 --warmup 10 --reps 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', json.dumps({k: v for k, v in d['end_to_end'].items() if k != 'what'}))"
done | tee $O/end_to_end.txt
python tools/small_call_latency.py > $O/small_call_latency.txt 2>&1; tail -12 $O/small_call_latency.txt
cd node-speex-resampler_amd/test
node bench.js $O/node_bench.json > $O/node_bench.log 2>&1; SPEEXHIP_NAPI_COPY=1 node bench.js $O/node_bench_copy.json > $O/node_bench_copy.log 2>&1
python3 - <<PY
import json
a = json.load(open("$O/node_bench.json"))["rows"]; b = json.load(open("$O/node_bench_copy.json"))["rows"]
for x, y in zip(a, b):
    print(x["inRate"], x["outRate"], x["channels"], "q", x["quality"], "| external vs copy: whole", x["whole_ms"], y["whole_ms"], "steady", x["steady_ms"], y["steady_ms"], "pipe", x["pipe_ms"], y["pipe_ms"], "first", x["whole_first_ms"], y["whole_first_ms"])
PY
