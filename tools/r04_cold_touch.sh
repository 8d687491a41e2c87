#!/bin/bash
# tools/r04_cold_touch.sh -- launches on warm caches (back to back) and on cold ones (a 64 MB read in between), with the tap
# rows fetched into L2 behind the window (SPEEXHIP_TOUCH=1) and without (=0): what a first call pays, what the fetch costs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for W in "2,44100,48000,7 1 1048576" "2,44100,48000,7 32 1048576" "2,44100,48000,10 1 1048576" "1,24000,48000,10 1 1048576" "8,48000,44100,5 1 1048576" \
         "1,48000,11025,7 32 131072" "2,48000,11025,7 32 131072" "4,48000,11025,7 32 131072" "3,48000,11025,7 32 131072" "1,48000,22050,7 32 131072" "2,48000,44100,7 32 131072" "2,48000,11025,7 1 441000" \
         "3,48000,11025,7 32 1048576" "2,48000,11025,7 32 1048576"; do
  set -- $W
  for T in 0 1; do
    SIZES_MB=64 SPEEXHIP_TOUCH=$T python tools/r04_cold_state.py $1 $2 $3 2>/dev/null | grep median | tail -2 | sed "s/^/touch $T  /"
  done
done
