"""Diagnostics: the many-states call over pinned chunks and pageable results (32 x 2^20 stereo frames), leg by leg in the
order given on the command line: many / apart / pinned_in / pinned_both (alloc_outs: hold result blocks without a leg
of their own).  Prints the median ms of each leg.  The order matters to the runtime's choice of copy engines: this is
the harness behind profiles/r06_pinned_in_leg.txt and tests/test_gpu_perf_gate.py's second test.
  python tools/many_pinned_probe.py [--calls N] leg leg ..."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPEEXHIP_PY_NO_TORCH", "1")
import speexhip  # noqa: E402
from bench import lcg_pcm, wrapper_capacity  # noqa: E402

lib = speexhip.lib()
args = sys.argv[1:]
calls = 8
if args and args[0] == "--calls":
    calls = int(args[1])
    args = args[2:]
ch, fi, fo, q = 2, 44100, 48000, 7
n, F = 32, 1 << 20
cap = wrapper_capacity(F * ch * 2, fi, fo, ch)
states = [speexhip.Resampler(ch, fi, fo, q) for _ in range(n)]
xs = [np.ascontiguousarray(lcg_pcm(F * ch, 12345 + s).reshape(F, ch)) for s in range(n)]
ys = [np.zeros((cap, ch), np.int16) for _ in range(n)]
hs = (C.c_void_p * n)(*[st._h for st in states])
ins = (C.c_void_p * n)(*[x.ctypes.data for x in xs])
outs = (C.c_void_p * n)(*[y.ctypes.data for y in ys])
il, ol, codes = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_int * n)()
bis = [speexhip.PinnedBlock(xs[0].nbytes) for _ in range(n)]
bos = [speexhip.PinnedBlock(cap * ch * 2) for _ in range(n)] if "pinned_both" in args or "alloc_outs" in args else []
for s in range(n):
    bis[s].array(np.int16, xs[s].shape)[...] = xs[s]
pin_ins = (C.c_void_p * n)(*[b.ptr for b in bis])
pin_outs = (C.c_void_p * n)(*[b.ptr for b in bos]) if bos else None


def many_with(a, b):
    def f():
        for i in range(n):
            il[i], ol[i] = F, cap
        rc = lib.speexhip_resampler_process_many_int(n, hs, a, il, b, ol, codes)
        assert rc == 0, rc
    return f


def apart():
    fn = lib.speexhip_resampler_process_interleaved_int
    for i in range(n):
        a, b = C.c_uint32(F), C.c_uint32(cap)
        rc = fn(states[i]._h, xs[i].ctypes.data_as(C.POINTER(C.c_int16)), C.byref(a), ys[i].ctypes.data_as(C.POINTER(C.c_int16)), C.byref(b))
        assert rc == 0, rc


legs = {"many": many_with(ins, outs), "apart": apart, "pinned_in": many_with(pin_ins, outs),
        "pinned_both": many_with(pin_ins, pin_outs) if bos else None}
res = []
for name in args:
    if name == "alloc_outs":
        continue
    fn = legs[name]
    sys.stderr.write("## leg %s\n" % name)
    sys.stderr.flush()
    for _ in range(2):
        fn()
    ts = []
    sys.stderr.write("## timed calls of %s\n" % name)
    sys.stderr.flush()
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    res.append("%s %.3f (min %.3f max %.3f)" % (name, sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3, max(ts) * 1e3))
print("  ".join(res))
