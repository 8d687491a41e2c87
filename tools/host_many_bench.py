#!/usr/bin/env python3
"""tools/host_many_bench.py -- BASELINE configs[4] as ONE host process reaches it (round 5): T independent single-stream
states, placed by the library's rule (SPEEXHIP_DEVICES=all: state k on GPU k mod the GPU count), fed pageable host
buffers through one speexhip_resampler_process_many_int call per step -- per GPU one transfer in, one launch per <= 32
states, one transfer out, the GPUs side by side from one thread each.  PCIe-inclusive: every GPU is a PCIe link.

  SPEEXHIP_DEVICES=all python tools/host_many_bench.py --streams 256            # an 8-GPU node: 32 states per GPU
  SPEEXHIP_ALIAS_DEVICES=2 SPEEXHIP_DEVICES=all python tools/host_many_bench.py --streams 64   # one GPU as two (a test)
Prints one JSON line."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, ROOT)
import speexhip
from bench import lcg_pcm, wrapper_capacity

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=32)
ap.add_argument("--frames", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--custom", default="2,44100,48000,7")
a = ap.parse_args()
ch, fi, fo, q = (int(v) for v in a.custom.split(","))
lib = speexhip.lib()
states = [speexhip.Resampler(ch, fi, fo, q) for _ in range(a.streams)]
devices = [s.info()["device"] for s in states]
cap = wrapper_capacity(a.frames * ch * 2, fi, fo, ch)
base = [np.ascontiguousarray(lcg_pcm(a.frames * ch, 12345 + s).reshape(a.frames, ch)) for s in range(min(a.streams, 8))]
xs = [base[s % len(base)] for s in range(a.streams)]
ys = [np.ones((cap, ch), np.int16) for _ in range(a.streams)]
n = a.streams
hs = (C.c_void_p * n)(*[st._h for st in states])
ins = (C.c_void_p * n)(*[x.ctypes.data for x in xs])
outs = (C.c_void_p * n)(*[y.ctypes.data for y in ys])
il, ol, codes = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_int * n)()


def step():
    for i in range(n):
        il[i], ol[i] = a.frames, cap
    rc = lib.speexhip_resampler_process_many_int(n, hs, ins, il, outs, ol, codes)
    assert rc == 0, rc


for _ in range(2):
    step()
ts = []
for _ in range(a.steps):
    t0 = time.perf_counter()
    step()
    ts.append(time.perf_counter() - t0)
ts.sort()
med = ts[len(ts) // 2]
print(json.dumps({"what": "one process, %d states by the library's placement rule, host buffers through one many-states call per step "
                          "(PCIe-inclusive)" % n, "config": [ch, fi, fo, q], "frames_per_stream": a.frames,
                  "devices": sorted(set(devices)), "states_per_device": {str(d): devices.count(d) for d in sorted(set(devices))},
                  "ms_per_step": round(med * 1e3, 4), "ms_min": round(ts[0] * 1e3, 4),
                  "input_msamples_per_s": round(n * a.frames * ch / med / 1e6, 1)}))
for s in states:
    s.close()
