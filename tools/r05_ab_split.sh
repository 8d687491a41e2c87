#!/bin/bash
# tools/r05_ab_split.sh -- the padded 8-channel loop split at its one period boundary (SPEEXHIP_SKIP=256 keeps the
# counting loop): same library, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for ARGS in "--config cfg4 --streams 32" "--config cfg4 --streams 1" "--custom 8,32000,44100,7 --streams 32 --frames 262144" "--custom 8,48000,11025,7 --streams 32 --frames 262144"; do
  for rep in 1 2 3; do
    bash tools/gpu_ab.sh SPEEXHIP_SKIP "256 0" $ARGS --steps 60 --warmup 10
  done
done
