#!/bin/bash
# tools/r05_probe1.sh -- first GPU call of round 5: the round's new tests, the call-2 / steady-state anomaly, the dispatch
# floor of other workgroup shapes, a bench line with the host-fed 32-stream leg.  Writes gpurun_out/r05_probe1/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05_probe1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "many_states or placement_rule or fast_float_pieces or node_drop_in or large_owned or loaded_in_tree" > $O/pytest_new.txt 2>&1; echo "pytest rc $?" >> $O/pytest_new.txt
timeout 300 python tools/r05_call_stamps.py > $O/call_stamps.txt 2>&1
timeout 300 node --expose-gc tools/r05_steady.js > $O/steady.txt 2>&1
timeout 120 tools/probe_dispatch > $O/probe_dispatch.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_dispatch -- $R/tools/probe_dispatch > /dev/null 2>&1)
find $O/prof_dispatch -name "*kernel_stats.csv" -exec cp {} $O/probe_dispatch_kernel_stats.csv \; 2>/dev/null
rm -rf $O/prof_dispatch
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --streams 32 --no-cpu-baseline > $O/bench_s32.json 2> $O/bench_s32.err
tail -5 $O/pytest_new.txt
