#!/usr/bin/env python3
"""tools/seg64_sim.py -- would a FAST kernel that sums short fp32 FMA chains in fp64 match the reference's
`direct_double` / `interpolate_double` arithmetic (deps/speex/resample.c:389-435: fp64 sums of fp32-ROUNDED
products, four partial sums) well enough to be worth building?  CPU simulation on BASELINE configs[2]'s filter
(24 kHz -> 48 kHz mono q10), full-scale white noise, no GPU needed.  Prints, per segment length, the share of
output samples that round to a different int16 than the reference's.

Finding (profiles/r03_cfg3_seg64_sim.txt): one fp32 FMA chain over all 256 taps -- what the FAST kernels do --
differs on 3.3-3.5e-3 of the samples.  Summing segments in fp64 helps the half-sample phase only slowly: 1.7e-3
at 64 taps per segment, 1.3e-3 at 32, 9e-4 at 16, 6.5e-4 at 8, 4.6e-4 at 4 (the on-grid phase, nearly a delta,
sits at 4-5e-4 from 64 taps down) -- and there it stops: what remains is the reference's OWN rounding of every
product to fp32 before it adds, which an FMA (unrounded product) does not reproduce at any segment length.
Both phases below 5e-4 needs segments of 4 taps: a convert and an fp64 add per accumulator every 4 FMAs, i.e.
the cost of SPEEXHIP_MODE_EXACT (separate multiply and add in the reference's order, bit-identical) for a result
that is still not identical.  So no segmented variant was built; configs[2] at the reference's precision is the
EXACT-mode line in profiles/r03_bench_lines.jsonl."""
import os
import sys

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
import speexhip


def main():
    info, table = speexhip.design_filter(24000, 48000, 10)  # host-only entry point: the reference-layout sinc table
    taps, den = info["filt_len"], info["den_rate"]
    print("filter: %d taps, kernel %s, %d phases" % (taps, speexhip.KERNEL_NAMES[info["kernel"]], den))
    rows = [table[p * taps:(p + 1) * taps].astype(np.float32) for p in range(den)]  # resample.c:398: sinc_table[samp_frac_num*N + j]
    rng = np.random.RandomState(1)
    n_out = 200000
    x = rng.randint(-32768, 32768, size=n_out + 400).astype(np.float32)

    def reference(h, w):  # fp64 sum of fp32-rounded products, four interleaved partial sums (resample.c:409-417)
        prod = (h[None, :] * w).astype(np.float32).astype(np.float64)
        return sum(prod[:, j::4].sum(axis=1) for j in range(4))

    def segmented(h, w, seg):  # fp32 FMA chains of `seg` taps (product unrounded, one rounding per FMA), summed in fp64
        out = np.zeros(w.shape[0], np.float64)
        for s0 in range(0, len(h), seg):
            a = np.zeros(w.shape[0], np.float32)
            for j in range(s0, min(s0 + seg, len(h))):
                a = (a.astype(np.float64) + h[j].astype(np.float64) * w[:, j].astype(np.float64)).astype(np.float32)
            out += a.astype(np.float64)
        return out

    for p, h in enumerate(rows):
        w = sliding_window_view(x, len(h))[:n_out]
        ref = reference(h, w)
        want = np.floor(ref + 0.5)
        for seg in (len(h), 64, 32, 16, 8, 4):  # (the kernels' own order differs in detail; the scaling is what matters)
            got = segmented(h, w, seg)
            label = "one fp32 chain (FAST kernels)" if seg == len(h) else "segments of %3d taps summed in fp64" % seg
            print("phase %d (%d taps)  %-36s differs on %.2e of the samples (mean |error| %.2e LSB)" % (
                p, len(h), label, (np.floor(got + 0.5) != want).mean(), np.abs(got - ref).mean()))


if __name__ == "__main__":
    main()
