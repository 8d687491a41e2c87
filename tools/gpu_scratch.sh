#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3; mkdir -p $O
for sk in 0 256; do
  echo "== float cfg2 s32 SPEEXHIP_SKIP=$sk"; SPEEXHIP_SKIP=$sk python tools/stamps.py --streams 32 --io float --launches 1 2>&1 | head -22
done > $O/stamps_float.txt 2>&1
echo "== int16 cfg2 s32" >> $O/stamps_float.txt; python tools/stamps.py --streams 32 --launches 1 2>&1 | head -22 >> $O/stamps_float.txt
cat $O/stamps_float.txt
