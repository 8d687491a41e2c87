#!/usr/bin/env python3
"""tools/call_stamps.py -- VERDICT r4 #5: calls 1..6 of one state through the C ABI on preallocated, touched
buffers (no allocation, no first-touch page faults inside the timed call): copy-out call, owned-block call, and the
kernel alone (device pointers, HIP events).  profiles/r04_init_cost.txt showed call 2 at 0.79 ms for 24k->48k stereo q10."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
import torch
import speexhip
lib = speexhip.lib()
for (ch, fi, fo, q, frames) in [(2, 24000, 48000, 10, 441011), (2, 44100, 48000, 7, 441011), (1, 24000, 48000, 5, 441022)]:
    x = (np.random.RandomState(1).randn(frames, ch) * 3000).astype(np.int16)
    cap = frames * fo // fi + 64
    y = np.ones((cap, ch), np.int16)  # touched
    px, py = x.ctypes.data_as(C.POINTER(C.c_int16)), y.ctypes.data_as(C.POINTER(C.c_int16))
    for path in ("copy", "take", "device"):
        r = speexhip.Resampler(ch, fi, fo, q)
        ts = []
        if path == "device":
            d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((cap, ch), dtype=torch.int16, device="cuda")
            s = torch.cuda.current_stream()
        for call in range(6):
            if path == "copy":
                il, ol = C.c_uint32(frames), C.c_uint32(cap)
                t0 = time.perf_counter()
                rc = lib.speexhip_resampler_process_interleaved_int(r._h, px, C.byref(il), py, C.byref(ol))
                ts.append((time.perf_counter() - t0) * 1e3)
            elif path == "take":
                il, ol, blk = C.c_uint32(frames), C.c_uint32(cap), C.POINTER(C.c_int16)()
                t0 = time.perf_counter()
                rc = lib.speexhip_resampler_process_interleaved_int_take(r._h, px, C.byref(il), C.byref(ol), C.byref(blk))
                ts.append((time.perf_counter() - t0) * 1e3)
                lib.speexhip_block_release(C.cast(blk, C.c_void_p))
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                r.process_device(d_in.data_ptr(), frames, d_out.data_ptr(), cap, s.cuda_stream)
                e1.record(s); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1)); rc = 0
            assert rc == 0
        i = r.info()
        print("ch=%d %d->%d q=%d %-6s fast_path=%d: calls 1..6 ms: %s   (position after: last=%d frac=%d)" % (
            ch, fi, fo, q, path, i["fast_path"], " ".join("%.3f" % t for t in ts), i["last_sample"], i["samp_frac_num"]), flush=True)
        r.close()
