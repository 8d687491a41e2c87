#!/bin/bash
# tools/scale.sh [TOTAL_STREAMS] [GPUS...] -- the BASELINE configs[4] scaling curve with one command, on a node
# with several MI355X: `bench.py --gpus N --total-streams 256` (256 independent stereo 44.1k->48k q7 streams,
# stream s on rank s % N, one launch per rank per step, no data-path collective) for N = 1 2 4 8, then the
# whole-job speed-up of every N against N = 1.  bench.py starts its own ranks (one process per GPU).
#   usage: bash tools/scale.sh            # 256 streams, N = 1 2 4 8
#          bash tools/scale.sh 64 1 2     # 64 streams, N = 1 2
# No multi-GPU node has been available to this repo so far (SCALE_r01/r02.json are "skipped" records): the
# curve below has never been measured; the per-GPU share (32 streams, one launch) is in profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
TOTAL=${1:-256}; shift
GPUS=${@:-1 2 4 8}
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
: > /tmp/scale_lines.jsonl
for N in $GPUS; do
  if [ "$N" -gt "$HAVE" ] && [ "${BENCH_SHARE_GPU:-0}" != "1" ]; then echo "N=$N: only $HAVE GPU(s) visible, skipped"; continue; fi
  STEPS=$(( 10 * N )); [ $STEPS -gt 40 ] && STEPS=40
  python bench.py --gpus $N --total-streams $TOTAL --steps $STEPS --warmup 3 --no-cpu-baseline ${EXTRA} | tail -1 >> /tmp/scale_lines.jsonl || echo "N=$N failed"
done
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open('/tmp/scale_lines.jsonl') if l.startswith('{')]
if not rows:
    raise SystemExit('no bench line')
base = next((r for r in rows if r['n_gpus'] == 1), rows[0])
print('%4s %14s %12s %10s %10s  %s' % ('GPUs', 'Msamples/s', 'ms/step', 'speed-up', 'per-GPU', 'streams per GPU'))
for r in rows:
    sp = r['value'] / base['value'] * base['n_gpus']
    print('%4d %14.1f %12.4f %9.2fx %9.0f%%  %d' % (r['n_gpus'], r['value'], r['ms_per_step'], sp, 100 * sp / r['n_gpus'], r['config']['streams_per_gpu']))
PY
# Round 5: the same streams as ONE host process reaches them -- the library's own placement (SPEEXHIP_DEVICES=all: state
# k on GPU k mod the GPU count) and one many-states call per step, host buffers in and out: every GPU is a PCIe link.
echo "one process, host-fed, library placement (tools/host_many_bench.py):"
for N in $GPUS; do
  if [ "$N" -gt "$HAVE" ]; then continue; fi
  LIST=$(python3 -c "print(','.join(str(i) for i in range($N)))")
  SPEEXHIP_DEVICES=$LIST python tools/host_many_bench.py --streams $TOTAL --steps 4 | tail -1
done
