#!/usr/bin/env python3
"""tools/pinned_path_bench.py -- what the pinned-input path (round 6: speexhip_block_acquire, buffers used in place) buys
a host caller, PCIe-inclusive, beside the link's own rate measured in the same run (plain pinned hipMemcpy, one way and
both ways at once).  One JSON document; profiles/r06_pinned_path.json is a run of it."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np
import speexhip
from bench import lcg_pcm, wrapper_capacity, pcie_peak

L = speexhip.lib()
out = {"pcie": pcie_peak(speexhip, 4 << 20)}


def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


cfgs = {"cfg2": (2, 44100, 48000, 7), "cfg3": (1, 24000, 48000, 10), "cfg4": (8, 48000, 44100, 5)}
one_line = '--one-line' in sys.argv  # for tools/ab.sh, which reads a command's last line
only = [a for a in sys.argv[1:] if not a.startswith('--')] or list(cfgs)
for name in only:
    ch, fi, fo, q = cfgs[name]
    for frames in (16384, 1 << 20):
        x = np.ascontiguousarray(lcg_pcm(frames * ch, 12345).reshape(frames, ch))
        cap = wrapper_capacity(x.size * 2, fi, fo, ch)
        y = np.zeros((cap, ch), np.int16)
        bi, bo = speexhip.PinnedBlock(x.nbytes), speexhip.PinnedBlock(cap * ch * 2)
        xi, yo = bi.array(np.int16, x.shape), bo.array(np.int16, (cap, ch))
        xi[...] = x
        r = speexhip.Resampler(ch, fi, fo, q)
        p16 = C.POINTER(C.c_int16)

        def plain(px, py):
            def f():
                il, ol = C.c_uint32(frames), C.c_uint32(cap)
                rc = L.speexhip_resampler_process_interleaved_int(r._h, C.cast(px, p16), C.byref(il), C.cast(py, p16), C.byref(ol))
                assert rc == 0, rc
                f.made = ol.value
            return f

        def take(px):
            def f():
                il, ol, blk = C.c_uint32(frames), C.c_uint32(cap), p16()
                rc = L.speexhip_resampler_process_interleaved_int_take(r._h, C.c_void_p(px), C.byref(il), C.byref(ol), C.byref(blk))
                assert rc == 0 and blk, rc
                f.made = ol.value
                L.speexhip_block_release(C.cast(blk, C.c_void_p))
            return f

        n = 40 if frames > 100000 else 400
        row = {}
        legs = (("pageable_copy_out", plain(x.ctypes.data, y.ctypes.data)), ("pageable_owned_block", take(x.ctypes.data)),
                ("pinned_in_copy_out", plain(xi.ctypes.data, y.ctypes.data)), ("pinned_in_owned_block", take(xi.ctypes.data)),
                ("pinned_in_pinned_out", plain(xi.ctypes.data, yo.ctypes.data)))
        for label, fn in legs:
            med, lo = timed(fn, n)
            row[label] = {"ms": round(med * 1e3, 4), "ms_min": round(lo * 1e3, 4)}
        made = legs[-1][1].made
        best = row["pinned_in_owned_block"]["ms"] * 1e-3
        row["bytes_in"], row["bytes_out"] = int(x.nbytes), int(made * ch * 2)
        row["pinned_in_owned_block_GBs"] = {"in": round(x.nbytes / best / 1e9, 2), "out": round(made * ch * 2 / best / 1e9, 2)}
        out["%s_%d" % (name, frames)] = row
        r.close()
        bi.close()
        bo.close()

# 32 states, one many-states call per step
ch, fi, fo, q = cfgs["cfg2"]
for frames in (16384, 1 << 20):
    S = 32
    cap = wrapper_capacity(frames * ch * 2, fi, fo, ch)
    states = [speexhip.Resampler(ch, fi, fo, q) for _ in range(S)]
    xs = [np.ascontiguousarray(lcg_pcm(frames * ch, 12345 + s).reshape(frames, ch)) for s in range(S)]
    ys = [np.zeros((cap, ch), np.int16) for _ in range(S)]
    try:
        bis = [speexhip.PinnedBlock(xs[0].nbytes) for _ in range(S)]
        bos = [speexhip.PinnedBlock(cap * ch * 2) for _ in range(S)]
    except MemoryError as e:
        out["many_%d" % frames] = {"error": str(e)}
        continue
    for s in range(S):
        bis[s].array(np.int16, xs[s].shape)[...] = xs[s]
    hs = (C.c_void_p * S)(*[st._h for st in states])
    il, ol, codes = (C.c_uint32 * S)(), (C.c_uint32 * S)(), (C.c_int * S)()

    def many(ins, outs):
        a = (C.c_void_p * S)(*ins)
        b = (C.c_void_p * S)(*outs)

        def f():
            for k in range(S):
                il[k], ol[k] = frames, cap
            rc = L.speexhip_resampler_process_many_int(S, hs, a, il, b, ol, codes)
            assert rc == 0, rc
        return f

    n = 8 if frames > 100000 else 100
    row = {}
    for label, fn in (("pageable", many([x.ctypes.data for x in xs], [y.ctypes.data for y in ys])),
                      ("pinned_in", many([b.ptr for b in bis], [y.ctypes.data for y in ys])),
                      ("pinned_in_pinned_out", many([b.ptr for b in bis], [b.ptr for b in bos]))):
        med, lo = timed(fn, n, warm=2)
        row[label] = {"ms": round(med * 1e3, 4), "ms_min": round(lo * 1e3, 4)}
    tot_in, tot_out = S * xs[0].nbytes, sum(int(ol[k]) for k in range(S)) * ch * 2
    best = row["pinned_in_pinned_out"]["ms"] * 1e-3
    row["bytes_in"], row["bytes_out"] = tot_in, tot_out
    row["pinned_GBs"] = {"in": round(tot_in / best / 1e9, 2), "out": round(tot_out / best / 1e9, 2)}
    out["many32_%d" % frames] = row
    for st in states:
        st.close()
    for b in bis + bos:
        b.close()
print(json.dumps(out) if one_line else json.dumps(out, indent=1))
