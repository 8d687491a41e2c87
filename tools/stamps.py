#!/usr/bin/env python3
"""tools/stamps.py -- where a launch of the period kernel spends its time (diagnostics).

Runs the bench workload against node-speex-resampler_amd/ab/libspeexhip_stamps.so (the library
built with -DSPEEXHIP_STAMPS: every workgroup records, on the 100 MHz clock all CUs share, when it
0 started, 1 had its descriptor and window geometry, 2 had issued its staging loads, 3 had written
its LDS image, 4 left the staging barrier, 5 finished its (last) FIR loop, 6 had issued its last
stores), warms the clocks, then stamps single launches and prints the distribution of every stamp
relative to the first workgroup's start, and of the phases between stamps.
usage: python tools/stamps.py [--streams S] [--config cfg2] (run through gpurun)"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SPEEXHIP_LIB_PATH"] = os.path.join(ROOT, "node-speex-resampler_amd", "ab", "libspeexhip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import speexhip
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--config", default="cfg2")
ap.add_argument("--frames", type=int, default=1 << 20)
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--custom", default=None, help="channels,in_rate,out_rate,quality (overrides --config)")
ap.add_argument("--io", default="int16", choices=["int16", "float"])
a = ap.parse_args()
ch, fi, fo, q = bench.CONFIGS[a.config] if not a.custom else tuple(int(v) for v in a.custom.split(","))
S, F = a.streams, a.frames
cap = bench.wrapper_capacity(F * ch * 2, fi, fo, ch)
b = speexhip.Batch(S, ch, fi, fo, q)
x = torch.from_numpy(np.stack([bench.lcg_pcm(F * ch, 12345 + s).reshape(F, ch) for s in range(S)])).cuda()
xs = [x, torch.roll(x, 17, 1).contiguous(), torch.roll(x, 34, 1).contiguous()]
y = torch.zeros((S, cap, ch), dtype=torch.int16, device="cuda")
if a.io == "float":
    xs = [v.to(torch.float32) for v in xs]
    y = y.to(torch.float32)
sp = torch.cuda.current_stream().cuda_stream
lib = speexhip.lib()
lib.speexhip_debug_stamps.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
lib.speexhip_debug_stamps_pp.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
N = 8192
W = 16
names = ["start", "desc+geometry", "loads issued", "LDS image written", "barrier passed", "FIR done", "stores issued"]
for launch in range(a.launches):
    # warm clocks with a train of launches, stamp the launch that follows it directly
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.3:
        for _ in range(32):
            b.process_device(xs[i % 3].data_ptr(), F * ch, F, y.data_ptr(), cap * ch, cap, sp, a.io == 'float'); i += 1
        torch.cuda.synchronize()
    assert lib.speexhip_debug_stamps(None, 0, 1) == 0 and lib.speexhip_debug_stamps_pp(None, 0, 1) == 0
    b.process_device(xs[launch % 3].data_ptr(), F * ch, F, y.data_ptr(), cap * ch, cap, sp, a.io == 'float')
    buf = np.zeros(N * W, np.uint64)
    assert lib.speexhip_debug_stamps(buf.ctypes.data, buf.size, 0) == 0
    if not buf.any():  # a phase-pair launch: the stamps of its translation unit
        assert lib.speexhip_debug_stamps_pp(buf.ctypes.data, buf.size, 0) == 0
    st = buf.reshape(N, W).astype(np.int64)
    live = st[:, 4] > 0          # workgroups that staged a window (not the history / padding blocks)
    w = st[live]
    origin = st[st[:, 0] > 0][:, 0].min()
    rel = (w[:, :7] - origin) * 0.01  # us
    print("launch %d: %d workgroups with a tile, %d blocks in all; last stamp at %.2f us" % (
        launch, live.sum(), (st[:, 0] > 0).sum(), rel.max()))
    for k, nm in enumerate(names):
        col = rel[:, k]
        print("  %-18s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f" % (
            nm, col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
    if (st[live][:, 11] > 0).any():  # tap-range shares: first / last wave out of the FIR loop, barrier, partial sums added
        raw = buf.reshape(N, W)[live]
        first = ((~raw[:, 13]).astype(np.int64) - origin) * 0.01
        b1 = (w[:, 11] - origin) * 0.01
        s2 = (w[:, 12] - origin) * 0.01
        for nm, col in (("first wave out of FIR", first - rel[:, 4]), ("last wave out of FIR", rel[:, 5] - rel[:, 4]),
                        ("barrier behind the FIR", b1 - rel[:, 5]), ("partial sums written + added", s2 - b1),
                        ("stores", rel[:, 6] - s2)):
            print("  shares: %-30s median %7.2f  p90 %7.2f  max %7.2f us" % (nm, np.median(col), np.percentile(col, 90), col.max()))
    d7 = w[:, 7].astype(np.float64)
    print("  longest FIR loop among a workgroup's waves, shader cycles (s_memtime): median %.0f p90 %.0f max %.0f" % (
        np.median(d7), np.percentile(d7, 90), d7.max()))
    ok = w[:, 10] > 0
    clk = w[ok, 9] / (w[ok, 10] * 10.0)  # cycles per ns = GHz (s_memtime over s_memrealtime, wave 0's FIR loop)
    if not ok.any():  # (launches with tap-range shares record no loop clocks)
        clk = np.array([float("nan")])
    print("  in-kernel clock over wave 0's FIR loop (s_memtime / s_memrealtime): median %.2f GHz  p10 %.2f  p90 %.2f" % (
        np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
    for k in range(1, 7):
        d = rel[:, k] - rel[:, k - 1]
        print("  phase %-38s median %6.2f  p90 %6.2f  max %6.2f us" % (names[k - 1] + " -> " + names[k], np.median(d), np.percentile(d, 90), d.max()))
    # per-CU view: workgroups grouped by (XCC, SE, SH, CU) of their first wave
    hw = w[:, 8]
    cu_key = ((hw >> 32) & 0xf) * 65536 + ((hw >> 8) & 0xff)
    keys = np.unique(cu_key)
    gaps, conc = [], []
    for k in keys:
        sel = rel[cu_key == k]
        order = np.argsort(sel[:, 0])
        sel = sel[order]
        # resident workgroups of this kernel on the CU over time; idle time between an end and the next start
        ev = sorted([(t, +1) for t in sel[:, 0]] + [(t, -1) for t in sel[:, 6]])
        cur, last_t, area = 0, ev[0][0], 0.0
        for t, dlt in ev:
            area += cur * (t - last_t)
            cur += dlt
            last_t = t
        conc.append(area / (ev[-1][0] - ev[0][0]))
        ends = np.sort(sel[:, 6])
        starts = np.sort(sel[:, 0])
        # slot turnover: the n-th end is followed by the (n+2)-th start when two slots alternate
        if len(starts) > 2:
            gaps.extend(list(starts[2:] - ends[: len(starts) - 2]))
    # how the workgroups that share a CU stand to each other: time with 0 / 1 / 2 of them inside their FIR loops
    # (between stamps 4 and 5), and the start offset of every workgroup to the one resident beside it
    in_fir = np.zeros(3)
    offs = []
    for k in keys:
        sel = rel[cu_key == k]
        ev = sorted([(t, +1) for t in sel[:, 4]] + [(t, -1) for t in sel[:, 5]])
        cur, last_t = 0, sel[:, 0].min()
        for t, dlt in ev:
            in_fir[min(cur, 2)] += t - last_t
            cur += dlt
            last_t = t
        in_fir[0] += sel[:, 6].max() - last_t
        for row in sel:
            mates = sel[(sel[:, 0] < row[6]) & (sel[:, 6] > row[0]) & (sel[:, 0] != row[0])]
            if len(mates):
                offs.append(np.abs(mates[:, 0] - row[0]).min())
    in_fir /= in_fir.sum()
    o = np.array(offs if offs else [0.0])
    print("  per CU, share of the launch with 0 / 1 / 2 workgroups inside their FIR loops: %.3f / %.3f / %.3f" % tuple(in_fir))
    print("  start offset between a workgroup and the nearest one resident beside it: median %.2f p10 %.2f p90 %.2f us" % (
        np.median(o), np.percentile(o, 10), np.percentile(o, 90)))
    print("  %d CUs seen; workgroups of this launch resident per CU (time average): median %.2f min %.2f max %.2f" % (
        len(keys), np.median(conc), np.min(conc), np.max(conc)))
    if gaps:
        g = np.array(gaps)
        print("  slot turnover (a workgroup's last store issued -> start of the workgroup after next on that CU): "
              "median %.2f p10 %.2f p90 %.2f us" % (np.median(g), np.percentile(g, 10), np.percentile(g, 90)))
b.close()
