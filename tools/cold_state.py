#!/usr/bin/env python3
"""tools/cold_state.py -- what state a launch inherits from the one before (diagnostics, run through gpurun).
The same batch launch timed with HIP events around each launch, back to back and with another kernel between two
launches that (a) reads 256 MB (every L2 and most of the Infinity Cache replaced), (b) reads 8 MB, (c) does nothing
but sit between them.  Diagnostics switches (SPEEXHIP_SKIP, SPEEXHIP_TOUCH, SPEEXHIP_PP ...) need the diagnostics build: run it through tools/ab.sh.
usage: python tools/cold_state.py channels,in,out,q [streams] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import speexhip
import bench

ch, fi, fo, q = (int(v) for v in sys.argv[1].split(","))
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
F = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
cap = bench.wrapper_capacity(F * ch * 2, fi, fo, ch)
b = speexhip.Batch(S, ch, fi, fo, q)
x = torch.from_numpy(np.stack([bench.lcg_pcm(F * ch, 12345 + s).reshape(F, ch) for s in range(S)])).cuda()
y = torch.zeros((S, cap, ch), dtype=torch.int16, device="cuda")
big = torch.ones(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
small = torch.ones(2 * 1024 * 1024, dtype=torch.float32, device="cuda")
tiny = torch.ones(1024, dtype=torch.float32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
between = {"back to back": None, "a tiny kernel between": tiny, "8 MB read between": small, "256 MB read between": big}
if os.environ.get("SIZES_MB"):
    between = {"back to back": None}
    for mb in os.environ["SIZES_MB"].split(","):
        between["%s MB read between" % mb] = big[: int(mb) * 262144]
for rep in range(2):
    for name, buf in between.items():
        ts = []
        for i in range(40):
            if buf is not None:
                buf.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.process_device(x.data_ptr(), F * ch, F, y.data_ptr(), cap * ch, cap, sp, False)
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(c) * 1e3 for a, c in ts[8:])
        print("%s S=%d F=%d  %-24s median %8.1f us  min %8.1f" % (sys.argv[1], S, F, name, us[len(us) // 2], us[0]))
