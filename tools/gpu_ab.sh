#!/bin/bash
# tools/gpu_ab.sh -- diagnostics at sustained clocks (bench.py preheats): table of configurations
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $O/ab.txt
run() { echo -n "$1 $2 steps=$3 : " >> $O/ab.txt; env $1 timeout 300 python bench.py $2 --steps $3 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('launch_us', d['roofline']['launch_us'], 'value', d['value'], 'hbm', d['roofline']['frac'], 'valu', d['valu']['frac'])" >> $O/ab.txt; }
for CFG in cfg2 cfg3 cfg4 f3; do
run "BENCH_STREAMS=1" "--config $CFG" 1000
run "BENCH_STREAMS=32" "--config $CFG" 100
done
cat $O/ab.txt
