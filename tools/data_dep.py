#!/usr/bin/env python3
"""tools/data_dep.py -- does a launch's time depend on the samples?  The same batch launch on random PCM, on
silence and on a constant, timed with HIP events over a train of launches (diagnostics, run through gpurun).
usage: python tools/data_dep.py channels,in,out,q [streams] [frames]"""
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
rnd = torch.from_numpy(np.stack([bench.lcg_pcm(F * ch, 12345 + s).reshape(F, ch) for s in range(S)])).cuda()
data = {"random": rnd, "silence": torch.zeros_like(rnd), "constant 1000": torch.full_like(rnd, 1000),
        "random >> 8": rnd >> 8}
y = torch.zeros((S, cap, ch), dtype=torch.int16, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for name, x in data.items():
        for _ in range(30):
            b.process_device(x.data_ptr(), F * ch, F, y.data_ptr(), cap * ch, cap, sp, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            b.process_device(x.data_ptr(), F * ch, F, y.data_ptr(), cap * ch, cap, sp, False)
        e1.record()
        torch.cuda.synchronize()
        print("%s  %-14s %8.1f us per launch" % (sys.argv[1], name, e0.elapsed_time(e1) * 10.0))
