#!/usr/bin/env python3
"""Per-call latency of the host-buffer path on small chunks (what a Transform stream or a realtime caller sees).
python tools/small_call_latency.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import speexhip
import oracle as orc

for ch, i, o, q in [(2, 44100, 48000, 7), (1, 24000, 48000, 10), (2, 48000, 16000, 5)]:
    for frames in (480, 960, 4096, 16384, 65536):
        x = (np.random.RandomState(frames).randn(frames, ch) * 3000).astype(np.int16)
        r = speexhip.Resampler(ch, i, o, q)
        ref = orc.Oracle(ch, i, o, q)
        worst = 0
        for k in range(5):
            got, used = r.process(x, frames * 4)
            want, wu = ref.process(x, frames * 4)
            assert used == wu and got.shape == want.shape
            worst = max(worst, int(np.abs(got.astype(np.int32) - want).max()))
        ts = []
        for k in range(300):
            t0 = time.perf_counter()
            r.process(x, frames * 4)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("ch=%d %d->%d q=%d %6d frames: median %.1f us, p10 %.1f, p90 %.1f   (max |diff| vs oracle %d LSB)" %
              (ch, i, o, q, frames, ts[150] * 1e6, ts[30] * 1e6, ts[270] * 1e6, worst), flush=True)
        r.close()
