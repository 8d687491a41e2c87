#!/bin/bash
# tools/r05_probe2.sh -- second GPU call of round 5: the whole GPU suite, the dispatch probe under rocprofv3, SQ counters by phase
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05_probe2; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dispatch -- $R/tools/probe_dispatch > $R/$O/probe_dispatch_rocprof.log 2>&1)
find $O/prof_dispatch -name "*kernel_stats.csv" | head -3
cp $(find $O/prof_dispatch -name "*kernel_stats.csv" | head -1) $O/probe_dispatch_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_dispatch
timeout 300 node --expose-gc tools/r05_steady.js > $O/steady.txt 2>&1
bash tools/r05_pmc_phases.sh > $O/pmc_phases.log 2>&1
cp gpurun_out/r05_pmc_phases/summary.txt $O/pmc_phases_summary.txt
