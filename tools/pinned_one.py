#!/usr/bin/env python3
"""tools/pinned_one.py CFG FRAMES -- one line: ms per call of the owned-block call (..._take) on a pageable and on a pinned
input, median and min (for tools/ab.sh: the strategies for a pinned input, profiles/r06_pinned_ab.txt)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np
import speexhip
from bench import lcg_pcm, wrapper_capacity, CONFIGS
L = speexhip.lib()
ch, fi, fo, q = CONFIGS[sys.argv[1]]
frames = int(sys.argv[2])
x = np.ascontiguousarray(lcg_pcm(frames * ch, 12345).reshape(frames, ch))
cap = wrapper_capacity(x.size * 2, fi, fo, ch)
bi = speexhip.PinnedBlock(x.nbytes)
xi = bi.array(np.int16, x.shape)
xi[...] = x
r = speexhip.Resampler(ch, fi, fo, q)
p16 = C.POINTER(C.c_int16)
res = {}
for label, ptr in (("pageable", x.ctypes.data), ("pinned", xi.ctypes.data)):
    def f():
        il, ol, blk = C.c_uint32(frames), C.c_uint32(cap), p16()
        rc = L.speexhip_resampler_process_interleaved_int_take(r._h, C.c_void_p(ptr), C.byref(il), C.byref(ol), C.byref(blk))
        assert rc == 0 and blk, rc
        L.speexhip_block_release(C.cast(blk, C.c_void_p))
    for _ in range(5):
        f()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    res[label] = [round(ts[len(ts) // 2] * 1e3, 4), round(ts[0] * 1e3, 4)]
print(json.dumps(res))
