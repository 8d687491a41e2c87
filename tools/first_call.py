#!/usr/bin/env python3
"""tools/first_call.py -- where the FIRST call of a process goes (7.97 ms for the first processChunk in a
Node process).  Each configuration runs in a fresh process: HIP initialisation (the first runtime call), the first
state (filter design, first device / pinned allocations, stream), its first call (the code object of the kernel's
translation unit is loaded at the first launch from it: the library keeps its kernels in seven translation units so
that a process loads only what it runs), its second call, and the first call of a SECOND configuration whose kernel
lives in the same unit (no load).  python tools/first_call.py   (one MI355X)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(%r, "node-speex-resampler_amd", "python"))
t0 = time.perf_counter()
hip = ctypes.CDLL("libamdhip64.so")
n = ctypes.c_int()
hip.hipGetDeviceCount(ctypes.byref(n)); hip.hipFree(None)
t_init = (time.perf_counter() - t0) * 1e3
t1 = time.perf_counter()
try:
    import torch  # (speexhip.lib() imports it so that one HIP runtime is shared; not part of what a Node process pays)
except ImportError:
    pass
t_torch = (time.perf_counter() - t1) * 1e3
import speexhip
def ms(f):
    t = time.perf_counter(); r = f(); return (time.perf_counter() - t) * 1e3, r
ch, i, o, q, frames = %s
x = (np.random.RandomState(1).randn(frames, ch) * 3000).astype(np.int16)
t_lib, _ = ms(speexhip.lib)
t_warm = 0.0
if os.environ.get("FIRST_CALL_WARMUP") == "1":   # (round 5: what the addon does at import, behind initPromise)
    t_warm, _ = ms(lambda: speexhip.lib().speexhip_warmup(-1))
t_new, r = ms(lambda: speexhip.Resampler(ch, i, o, q))
t_first, _ = ms(lambda: r.process(x, frames * 7))
t_second, _ = ms(lambda: r.process(x, frames * 7))
ch2, i2, o2, q2 = %s
r2 = speexhip.Resampler(ch2, i2, o2, q2)
x2 = (np.random.RandomState(2).randn(frames, ch2) * 3000).astype(np.int16)
t_other, _ = ms(lambda: r2.process(x2, frames * 7))
print("%%-34s HIP init %%7.1f ms | (import torch %%.0f) | dlopen libspeexhip %%5.1f | warm-up %%6.1f | first state %%5.2f | first call %%6.2f | second %%5.2f | first call of %%s (same unit) %%5.2f   [fast_path %%d]" %%
      ("%%dch %%d->%%d q%%d (%%s)" %% (ch, i, o, q, %r), t_init, t_torch, t_lib, t_warm, t_new, t_first, t_second, "%%dch %%d->%%d q%%d" %% (ch2, i2, o2, q2), t_other, r.info()["fast_path"]))
'''
CASES = [((2, 44100, 48000, 7, 441000), (2, 48000, 44100, 5), "kernels_period.hip"),
         ((1, 24000, 48000, 5, 441000), (1, 16000, 48000, 7), "kernels_slide_i16.hip"),
         ((1, 24000, 48000, 10, 441000), (2, 16000, 48000, 9), "kernels_slide64_i16.hip"),
         ((2, 44100, 48000, 10, 441000), (1, 48000, 44100, 9), "kernels_period64.hip"),
         ((1, 44100, 8000, 7, 441000), (1, 48000, 22050, 5), "kernels_period.hip / _pp.hip"),
         ((2, 44100, 48300, 3, 441000), (2, 88000, 8000, 5), "kernels_exact.hip")]
print("code objects in libspeexhip.so: %.1f MB" % (os.path.getsize(os.path.join(ROOT, "node-speex-resampler_amd", "libspeexhip.so")) / 1e6))
for a, b, unit in CASES:
    res = subprocess.run([sys.executable, "-c", CHILD % (ROOT, repr(a), repr(b), unit)], capture_output=True, text=True)
    print((res.stdout.strip() or res.stderr.strip()[-400:]), flush=True)
    if os.environ.get("SPEEXHIP_INIT_TRACE") or os.environ.get("SPEEXHIP_POOL_TRACE"):   # the library's own stamps (stderr)
        for line in res.stderr.splitlines():
            if line.startswith("speexhip "):
                print("    " + line, flush=True)
