#!/bin/bash
# Which HIP runtime a process loads matters to the host-fed path: python + torch brings torch's bundled libamdhip64, a
# process without torch (the Node addon, a C caller) /opt/rocm's.  The link probe, the strategies for a pinned input and the
# many-states legs under both.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
OUT=$O/r06_runtime_ab.txt; rm -f $OUT
for NT in 0 1; do
  export SPEEXHIP_PY_NO_TORCH=$NT
  echo "## SPEEXHIP_PY_NO_TORCH=$NT ($( [ $NT = 1 ] && echo "/opt/rocm runtime" || echo "torch's bundled runtime"))" | tee -a $OUT
  python3 -c "
import sys; sys.path.insert(0,'node-speex-resampler_amd/python')
import speexhip
for mb in (4, 64):
    print('pcie probe %d MiB: h2d %.1f d2h %.1f both-each %.1f GB/s' % ((mb,) + speexhip.pcie_peak(mb << 20)))
import os
print('libamdhip64 loaded:', sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))
" 2>&1 | tee -a $OUT
  tools/ab.sh -o $OUT -- "" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=2" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=4" "SPEEXHIP_PIECES=1" "SPEEXHIP_PIECES=2" "SPEEXHIP_PIECES=4" -- python tools/pinned_one.py cfg2 1048576
  python tools/pinned_path_bench.py cfg2 > $O/r06_pinned_path_notorch$NT.json 2>&1
  python3 -c "
import json;d=json.load(open('$O/r06_pinned_path_notorch$NT.json'))
for k,v in d.items():
    if k=='pcie': print(k,{a:b for a,b in v.items() if a!='what'}); continue
    print(k,{a:(b['ms'] if isinstance(b,dict) and 'ms' in b else b) for a,b in v.items()})" | tee -a $OUT
done
(cd node-speex-resampler_amd && node test/bench.js $O/r06_node_bench_quick.json 2>&1 | tail -15) | tee -a $OUT
