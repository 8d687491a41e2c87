#!/usr/bin/env python3
"""tools/host_path_bench.py -- end-to-end (PCIe-inclusive) rate of the drop-in call with HOST buffers:
speexhip_resampler_process_interleaved_int = pinned staging + H2D + kernel + D2H, synchronous.
Reported in DESIGN.md next to (never instead of) bench.py's HBM-resident `value`."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import speexhip
import oracle as orc

out = {}
for name, (ch, fi, fo, q) in {"cfg2": (2, 44100, 48000, 7), "cfg3": (1, 24000, 48000, 10), "cfg4": (8, 48000, 44100, 5)}.items():
    for frames in (16384, 1 << 20):
        x = orc.lcg_pcm(frames * ch, 12345).reshape(frames, ch)
        cap, _ = orc.wrapper_capacity(x.size * 2, fi, fo, ch)
        r = speexhip.Resampler(ch, fi, fo, q)
        # the C call itself, on preallocated buffers (no numpy allocation / copy in the timed loop)
        import ctypes as C
        L = speexhip.lib()
        y = np.zeros((cap, ch), np.int16)
        px, py = x.ctypes.data_as(C.POINTER(C.c_int16)), y.ctypes.data_as(C.POINTER(C.c_int16))

        def call():
            il, ol = C.c_uint32(frames), C.c_uint32(cap)
            rc = L.speexhip_resampler_process_interleaved_int(r._h, px, C.byref(il), py, C.byref(ol))
            assert rc == 0

        for _ in range(50 if frames < 100000 else 10):
            call()
        n = 2000 if frames < 100000 else 100
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        dt = (time.perf_counter() - t0) / n
        out["%s_%d" % (name, frames)] = {"ms_per_call": round(dt * 1e3, 4), "input_msamples_per_s": round(frames * ch / dt / 1e6, 1)}
        r.close()
print(json.dumps(out, indent=1))
