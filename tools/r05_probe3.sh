#!/bin/bash
# tools/r05_probe3.sh -- pipelined many-call (test + bench leg), where a first state's 28 ms go
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05_probe3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "large_many_states or many_states_in_one" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
SPEEXHIP_MANY_PIPELINE=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-parity > $O/bench_default_nopipe.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default.json", "bench_default_nopipe.json"):
    d = json.loads(open("gpurun_out/r05_probe3/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["launch_us"], json.dumps({k: v for k, v in d["end_to_end_streams"].items() if k != "what"}))
PY
SPEEXHIP_INIT_TRACE=1 SPEEXHIP_POOL_TRACE=1 timeout 300 python tools/first_call.py > $O/first_call_trace.txt 2>&1
head -60 $O/first_call_trace.txt
