#!/bin/bash
# tools/lease.sh -- the named recipes one gpurun lease runs (replaces the per-step r06_* shells; what they wrote stays under
# profiles/).  Every recipe writes under gpurun_out/, the files to keep are copied into profiles/ by hand.
#
#   gpurun --timeout 3000 -- 'bash tools/lease.sh collect'        the round's collection: smoke, GPU suite, fuzz, profiles, floor
#   gpurun --timeout 2400 -- 'bash tools/lease.sh soak'           differential fuzz, ~25 minutes over the five call shapes
#   gpurun --timeout 900  -- 'bash tools/lease.sh suite [-k expr]'
#   gpurun ...            -- 'bash tools/lease.sh bounds'         cfg4 and the slide kernel by skip mask, the headline under R / splits / tiles
#   gpurun ...            -- 'bash tools/lease.sh pinned'         the link under both HIP runtimes, pinned input strategies, lanes
#   gpurun ...            -- 'BASE=node-speex-resampler_amd/ab/libX.so bash tools/lease.sh lib-ab NAME "--config cfg4" "--custom 2,56000,48000,4" ...'
#                                                                 library against library, same box (a kept build of another commit)
#   gpurun ...            -- 'bash tools/lease.sh sweep-frames'   frames of 9-24 channels over a few ratios
# N = round number in the output names (default 06).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; N=${N:-06}; mkdir -p $O; cd $R
PKG=$R/node-speex-resampler_amd
LAUNCH="'launch_us %s (min %s, max %s)  valu %s' % (d['roofline']['launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], d['valu']['frac'])"
what=$1; shift

suite() {  # [-k expr]
  timeout 2400 python -m pytest tests -m gpu -q -x "$@" > $O/r${N}_pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/r${N}_pytest_gpu.txt
  tail -6 $O/r${N}_pytest_gpu.txt
}
fuzz() {  # seconds-scale, seed base
  local k=$1 s=$2
  ( timeout $((7*k+80)) python tools/fuzz_gpu.py --seconds $((7*k)) --seed $s
    timeout $((5*k+80)) python tools/fuzz_gpu.py --seconds $((5*k)) --seed $((s+1)) --batch
    timeout $((4*k+60)) python tools/fuzz_gpu.py --seconds $((4*k)) --seed $((s+2)) --many
    timeout $((4*k+60)) python tools/fuzz_gpu.py --seconds $((4*k)) --seed $((s+3)) --many-channels
    SPEEXHIP_MODE=fast timeout $((4*k+60)) python tools/fuzz_gpu.py --seconds $((4*k)) --seed $((s+4)) ) 2>&1 | grep "fuzz:\|FAIL"
}

case "$what" in
suite) suite "$@";;
soak) fuzz 60 ${SEED:-631} | tee $O/r${N}_soak.txt;;
collect)
  timeout 300 python __graft_entry__.py smoke > $O/r${N}_smoke.txt 2>&1; tail -5 $O/r${N}_smoke.txt
  suite
  fuzz 20 ${SEED:-620} | tee $O/r${N}_fuzz_final.txt
  bash tools/gpu_profile.sh $N > $O/r${N}_profile_log.txt 2>&1; tail -45 $O/r${N}_profile_log.txt
  timeout 1200 python tools/perf_floor.py --measure --merge > $O/r${N}_perf_floor.txt 2>&1; tail -5 $O/r${N}_perf_floor.txt
  cp profiles/perf_floor.json profiles/pmc_traffic.json $O/
  timeout 300 python bench.py --steps 20 --warmup 5 > $O/r${N}_bench_driver_form.json 2> $O/r${N}_bench.err; echo "bench rc=$?";;
bounds)
  rm -f $O/r${N}_cfg4_bound.txt $O/r${N}_headline_r.txt $O/r${N}_slide_bound.txt
  for S in 32 1; do
    tools/ab.sh -o $O/r${N}_cfg4_bound.txt -f "$LAUNCH" -- "" "SPEEXHIP_SKIP=2" "SPEEXHIP_SKIP=8" "SPEEXHIP_SKIP=10" "" "SPEEXHIP_SKIP=2" -- \
      python bench.py --config cfg4 --streams $S --steps 40 --warmup 5 --reps 3 --mode fast_fixed --no-cpu-baseline --no-parity
  done
  tools/ab.sh -o $O/r${N}_headline_r.txt -f "$LAUNCH" -- "" "SPEEXHIP_R=10" "SPEEXHIP_R=5" "SPEEXHIP_SPLITS=1" "SPEEXHIP_SPLITS=2" "SPEEXHIP_SPLITS=4" \
    "SPEEXHIP_TILE_PERIODS=56" "SPEEXHIP_TILE_PERIODS=48" "SPEEXHIP_TILE_PERIODS=32" "" -- python bench.py --steps 200 --warmup 20 --reps 5 --no-cpu-baseline --no-parity
  for C in "--config f3" "--custom 2,24000,48000,5" "--custom 2,48000,24000,5" "--custom 1,16000,48000,7" "--custom 2,48000,16000,7" "--config cfg3 --mode fast_f32"; do
    tools/ab.sh -o $O/r${N}_slide_bound.txt -f "$LAUNCH" -- "" "SPEEXHIP_SKIP=2" "SPEEXHIP_SKIP=8" "SPEEXHIP_SKIP=10" -- \
      python bench.py $C --streams 32 --steps 40 --warmup 5 --reps 3 --no-cpu-baseline --no-parity
  done;;
lib-ab)  # NAME config...   (BASE = the other library)
  NAME=$1; shift; OUT=$O/r${N}_${NAME}_ab.txt; rm -f $OUT
  [ -f "$BASE" ] || { echo "lease.sh lib-ab: BASE=$BASE is not a library" >&2; exit 2; }
  for REP in 1 2; do for C in "$@"; do
    tools/ab.sh -o $OUT -f "'launch_us %s (min %s) valu %s fast_path %s parity %s' % (d['roofline']['launch_us'], d['roofline']['launch_us_min'], d['valu']['frac'], d['config']['fast_path'], d['parity'].get('max_abs_diff_lsb', d['parity'].get('max_abs_diff')))" \
      -- "SPEEXHIP_LIB_PATH=$R/$BASE" "SPEEXHIP_LIB_PATH=$PKG/libspeexhip.so" -- python bench.py $C --streams ${STREAMS:-32} --steps 40 --warmup 5 --reps 3 --mode fast_fixed --no-cpu-baseline
  done; done;;
pinned)
  # Which HIP runtime a process loads matters to the host-fed path: python + torch brings torch's bundled libamdhip64, a
  # process without torch (the Node addon, a C caller) /opt/rocm's.
  OUT=$O/r${N}_runtime_ab.txt; rm -f $OUT $O/r${N}_pinned_ab.txt
  for NT in 0 1; do
    export SPEEXHIP_PY_NO_TORCH=$NT
    echo "## SPEEXHIP_PY_NO_TORCH=$NT" | tee -a $OUT
    python3 -c "
import sys; sys.path.insert(0,'node-speex-resampler_amd/python')
import speexhip
for mb in (4, 64):
    print('pcie probe %d MiB: h2d %.1f d2h %.1f both-each %.1f GB/s' % ((mb,) + speexhip.pcie_peak(mb << 20)))
print('libamdhip64 loaded:', sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))" 2>&1 | tee -a $OUT
    tools/ab.sh -o $OUT -- "" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=2" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=4" "SPEEXHIP_PIECES=1" "SPEEXHIP_PIECES=2" \
      "SPEEXHIP_PIECES=4" -- python tools/pinned_one.py cfg2 1048576
    python tools/pinned_path_bench.py cfg2 > $O/r${N}_pinned_path_notorch$NT.json 2>&1
  done
  export SPEEXHIP_PY_NO_TORCH=1
  for CFG in cfg2 cfg3; do
    tools/ab.sh -o $O/r${N}_pinned_ab.txt -- "" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=2" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=4" "SPEEXHIP_PINNED_IN_PIECES=1 SPEEXHIP_PIECES=8" \
      "SPEEXHIP_TILE_PERIODS=32" "SPEEXHIP_TILE_PERIODS=16" "SPEEXHIP_TILE_PERIODS=8" "SPEEXHIP_MODE=fast_fixed SPEEXHIP_TILE_PERIODS=16" -- python tools/pinned_one.py $CFG 1048576
  done
  # the many-states call: lanes and what a pinned input beside pageable results does
  tools/ab.sh -o $O/r${N}_lanes_ab.txt -f "{k: {a: b['ms'] for a, b in v.items() if a in ('pageable', 'pinned_in', 'pinned_in_pinned_out')} for k, v in d.items() if k.startswith('many32_1048576')}" \
    -- "SPEEXHIP_MANY_LANES=1" "SPEEXHIP_MANY_LANES=2" "SPEEXHIP_MANY_LANES=1 SPEEXHIP_PIN_IN_COPY=0" "SPEEXHIP_MANY_LANES=2 SPEEXHIP_PIN_IN_COPY=0" -- python tools/pinned_path_bench.py cfg2 --one-line
  for C in 0 1; do SPEEXHIP_NAPI_COPY=$C node --expose-gc tools/node_pinned_ab.js 2>&1 | tail -2 | tee -a $O/r${N}_node_pinned_ab.txt; done;;
sweep-frames)
  PAIRS="48000,11025 44100,16000 96000,11025 44100,48000 48000,44100 24000,48000" CHANNELS="8 9 10 11 12 13 14 15 16 17 20 24" Q=7 REPS=2 \
    bash tools/perf_sweep.sh 2>/dev/null > $O/r${N}_sweep_frames.txt
  sort -t'|' -k4 $O/r${N}_sweep_frames.txt | head -100;;
*) sed -n '2,15p' "$0"; exit 2;;
esac
