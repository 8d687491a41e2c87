#!/usr/bin/env python3
"""Where a fresh state's first call goes: create / first call / later calls / close, per configuration
(host-buffer path, one MI355X).  python tools/init_cost.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
import speexhip

def ms(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r

cfgs = [(1, 24000, 48000, 5, 441022), (2, 24000, 24000, 5, 441011), (2, 24000, 48000, 10, 441011),
        (2, 44100, 48000, 7, 441011), (2, 44100, 48000, 10, 441011), (2, 44100, 24000, 5, 441011)]
speexhip.Resampler(2, 44100, 48000, 3).close()  # runtime + module load
for ch, i, o, q, frames in cfgs * 2:
    x = (np.random.RandomState(1).randn(frames, ch) * 3000).astype(np.int16)
    t_new, r = ms(lambda: speexhip.Resampler(ch, i, o, q))
    t_first, _ = ms(lambda: r.process(x, frames * 3))
    t_second, _ = ms(lambda: r.process(x, frames * 3))
    t_third, _ = ms(lambda: r.process(x, frames * 3))
    t_close, _ = ms(r.close)
    print("ch=%d %d->%d q=%d: create %.3f ms, first call %.3f, second %.3f, third %.3f, close %.3f" %
          (ch, i, o, q, t_new, t_first, t_second, t_third, t_close), flush=True)
