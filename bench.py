#!/usr/bin/env python3
"""bench.py -- throughput of the resampler hot path on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic input: every stream owned by
a rank resamples its next chunk.  Default workload = BASELINE.json configs[1]: 44100->48000 Hz,
2 ch int16, q=7, one 2^20-frame chunk per stream per step, ONE stream per GPU (weak scaling:
the per-GPU workload stays fixed as N grows).  `--total-streams T` is BASELINE.json configs[4]:
T independent streams in the whole job, stream s on rank s % N (strong scaling; T = 256 gives
32 streams per GPU at N = 8).  Inputs and outputs are resident in HBM when the timed region
starts; the streams are stateful (each step continues the previous one, as processChunk calls
do).  One process per GPU; no data-path collective (streams are independent): torch.distributed
(RCCL) only carries the barrier, the MAX of the elapsed times and a checksum/sample-count SUM.

Launch: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the
driver's form; WORLD_SIZE must equal N), or plainly as `python bench.py --gpus N`: with no
WORLD_SIZE in the environment and N > 1 this process starts N children itself, one per GPU,
BEFORE anything touches a GPU, relays rank 0's line and exits with their status.

Timing: W untimed warmup steps, then R repetitions (default 5) of EXACTLY K steps, each
repetition bracketed by a barrier + torch.cuda.synchronize() on both sides, MAX over ranks per
repetition; `ms_per_step`/`value` are the MEDIAN repetition (min and max are in the line too):
K steps of a 13 us kernel are over in a fraction of a millisecond, and one such region swings
by 10 % with launch-train edge effects.

Prints ONE JSON line on rank 0 (contract in the task description): value = whole-job input
Msamples/s, plus "roofline" (algorithmic HBM bytes per launch / launch time measured with HIP
events on the launch stream, vs the 8 TB/s HBM peak), "parity" (first chunk of up to four
streams checked against the CPU oracle) and, at N=1, "cpu_baseline" (the reference's own C
timed on the host cores) and the HOST-FED legs, reported beside `value`, never as it: "end_to_end"
(the host-buffer call a Node caller makes, PCIe included: pageable input, and -- round 6 -- a chunk
in a pinned block of the library, read in place), "end_to_end_streams" (32 states through one
many-states call), each with a "pcie" block {achieved_in_GBs, achieved_out_GBs, peak, frac}
against "pcie_peak", the link's own rate measured in the same run by plain pinned copies one way
and both ways at once.  The host-fed legs and the probe run in a CHILD process without torch: the
library then sits on /opt/rocm's HIP runtime, the one a Node or C caller of the drop-in loads
(behind torch it shares torch's bundled runtime, under which pinned copies of opposite directions
do not overlap -- profiles/r06_runtime_ab.txt).
"""
import argparse
import json
import math
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
# (the 32-stream pinned legs of end_to_end_streams hold 32 x (4.2 + 4.6) MB of the library's pinned blocks at once: more
#  than the default cap of 256 MiB on its slabs -- a documented limit of the library, include/speexhip_resampler.h)
os.environ.setdefault("SPEEXHIP_TAKE_MAX_MB", "1024")

CONFIGS = {
    # name: (channels, in_rate, out_rate, quality) -- BASELINE.json configs[1..3] (+ SURVEY F3)
    "cfg2": (2, 44100, 48000, 7),
    "cfg3": (1, 24000, 48000, 10),
    "cfg4": (8, 48000, 44100, 5),
    "f3": (1, 24000, 48000, 5),
}
CONFIG_LABEL = {"cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2]", "cfg4": "BASELINE configs[3]",
                "f3": "SURVEY F3 (direct_single)"}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak
VALU64_PEAK_TFLOPS = 78.6  # fp64 vector peak: v_fma_f64 issues at the rate of v_pk_fma_f32, half the FMAs each
                           # (tools/ubench_fma64.hip, profiles/r04_ubench_fma64.txt)
PMC_FILE = os.path.join("profiles", "pmc_traffic.json")


def wrapper_capacity(chunk_bytes, in_rate, out_rate, channels):
    """Frames of output room the JS wrapper grants a first call (reference src/index.ts:80-95)."""
    return int(math.ceil(chunk_bytes * out_rate / in_rate) / channels / 2)


def lcg_pcm(n, seed):
    import numpy as np
    with np.errstate(over="ignore"):
        a = np.cumprod(np.full(n, 1664525, np.uint32), dtype=np.uint32)
        geo = np.cumsum(np.concatenate(([np.uint32(1)], a[:-1])), dtype=np.uint32)
        s = a * np.uint32(seed) + np.uint32(1013904223) * geo
    return (s >> np.uint32(16)).astype(np.uint16).view(np.int16)


def shard_streams(total_streams, world, rank):
    """Global stream ids owned by `rank` (the rule of dist_util.shard_streams, restated here so
    that the launcher half of this file needs no torch import)."""
    return [s for s in range(total_streams) if s % world == rank]


def _oracle_engine(cfg):
    """The CPU checker (the ONLY place bench.py touches oracle/): oracle/_ref -- the reference's
    own C, kind "reference" -- when it is built, else the restatement ("port")."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    ch, fi, fo, q = cfg
    kind = "reference" if orc.have_reference() else "port"
    return (orc.Reference if kind == "reference" else orc.Oracle)(ch, fi, fo, q), kind


def parity_block(cfg, frames, first_chunks, float_io):
    """First chunk of each sampled stream against the CPU oracle: +-1 LSB (int16) and equal
    counters; the float entry point within 4e-6 of full scale.  first_chunks: list of
    (global stream id, output array, consumed, produced)."""
    import numpy as np
    ch, fi, fo, q = cfg
    worst, mism, n, counters, checked = 0.0, 0, 0, True, []
    for sid, got, used, made in first_chunks:
        eng, kind = _oracle_engine(cfg)
        x = lcg_pcm(frames * ch, 12345 + sid).reshape(frames, ch)
        cap = wrapper_capacity(x.size * 2, fi, fo, ch)
        if float_io:
            want, want_used = eng.process_float(x.astype(np.float32) / np.float32(32768.0), cap)
            diff = np.abs(got.astype(np.float64) - want.astype(np.float64)) if got.shape == want.shape else None
        else:
            want, want_used = eng.process(x, cap)
            diff = np.abs(got.astype(np.int32) - want.astype(np.int32)) if got.shape == want.shape else None
        counters = counters and used == want_used and made == want.shape[0] and diff is not None
        if diff is not None:
            worst = max(worst, float(diff.max()))
            mism += int((diff != 0).sum())
            n += diff.size
        checked.append(sid)
    blk = {"checker": "oracle/_ref (reference's native C)" if kind == "reference" else "oracle port",
           "streams_checked": checked, "checked_frames_per_stream": int(first_chunks[0][3]),
           "counters_equal": bool(counters)}
    if float_io:
        blk["max_abs_diff"] = worst
        blk["tolerance"] = 4e-6
        ok = counters and worst <= 4e-6
    else:
        blk["max_abs_diff_lsb"] = int(worst)
        blk["mismatch_rate"] = mism / max(n, 1)
        blk["tolerance_lsb"] = 1
        ok = counters and worst <= 1
    return blk, ok


def pcie_peak(speexhip, chunk_bytes):
    """The PCIe link's own rate, measured in this run on this box by the library's probe (plain pinned hipMemcpyAsync, one
    way each and both ways at once on two streams): at 64 MiB -- the link -- and at the size of the chunk the host-fed
    legs move, where a copy's fixed cost shows.  The roofline of end_to_end* (never of `value`): `peak` = bytes in + out per
    second with both directions busy, at the chunk's size."""
    big = speexhip.pcie_peak(64 << 20)
    at = speexhip.pcie_peak(max(int(chunk_bytes), 1 << 16))
    return {"h2d_GBs": round(at[0], 2), "d2h_GBs": round(at[1], 2), "both_ways_each_GBs": round(at[2], 2),
            "peak": round(2 * at[2], 2), "unit": "GB/s", "copy_bytes": int(chunk_bytes),
            "at_64MiB": {"h2d_GBs": round(big[0], 2), "d2h_GBs": round(big[1], 2), "both_ways_each_GBs": round(big[2], 2)},
            "what": "plain pinned hipMemcpyAsync measured in this run (speexhip_debug_pcie_peak): one way each, and both ways "
                    "at once on two streams, of %d bytes (one chunk of the workload) and of 64 MiB; peak = in + out with both "
                    "directions busy at the chunk's size" % chunk_bytes}


def pcie_block(peak, bytes_in, bytes_out, seconds, big=False):
    """{achieved_in_GBs, achieved_out_GBs, peak, frac} of one host-fed call against pcie_peak(): the rates at the chunk's
    size, or (big) at 64 MiB for the legs that move hundreds of MB per step"""
    ref = peak["at_64MiB"] if big else peak
    both = 2 * ref["both_ways_each_GBs"]
    ain, aout = bytes_in / seconds / 1e9, bytes_out / seconds / 1e9
    return {"achieved_in_GBs": round(ain, 2), "achieved_out_GBs": round(aout, 2), "peak": round(both, 2), "unit": "GB/s",
            "frac": round((ain + aout) / both, 4), "peak_at": "64 MiB copies" if big else "%d-byte copies" % peak["copy_bytes"],
            "h2d_GBs": ref["h2d_GBs"], "d2h_GBs": ref["d2h_GBs"], "both_ways_each_GBs": ref["both_ways_each_GBs"]}


def end_to_end(speexhip, cfg, frames, mode, float_io, base_stream, calls=30, peak=None):
    """The call a host-buffer caller makes (speexhip_resampler_process_interleaved_*: what index.js's
    processChunk runs): pageable buffers in and out, synchronous, PCIe both ways inside the timed call.
    Reported beside the kernel-resident `value`, never as it."""
    import numpy as np
    ch, fi, fo, q = cfg
    x = base_stream if not float_io else (base_stream.astype(np.float32) / np.float32(32768.0))
    import ctypes as C
    x = np.ascontiguousarray(x)
    r = speexhip.Resampler(ch, fi, fo, q, mode=mode)
    cap = wrapper_capacity(frames * ch * 2, fi, fo, ch)
    # the C call itself on preallocated buffers: no numpy allocation or page faults in the timed calls
    y = np.zeros((cap, ch), np.float32 if float_io else np.int16)
    ctype = C.c_float if float_io else C.c_int16
    fn = (speexhip.lib().speexhip_resampler_process_interleaved_float if float_io
          else speexhip.lib().speexhip_resampler_process_interleaved_int)
    px, py = x.ctypes.data_as(C.POINTER(ctype)), y.ctypes.data_as(C.POINTER(ctype))
    used = 0

    def call():
        il, ol = C.c_uint32(frames), C.c_uint32(cap)
        rc = fn(r._h, px, C.byref(il), py, C.byref(ol))
        assert rc == 0, rc
        return il.value

    for _ in range(3):
        call()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        used = call()
        ts.append(time.perf_counter() - t0)
    # ... and the call the N-API addon makes since round 4: the result stays in a pinned block of the library that the
    # caller then owns (an external Buffer in JavaScript): no copy out of it
    take = (speexhip.lib().speexhip_resampler_process_interleaved_float_take if float_io
            else speexhip.lib().speexhip_resampler_process_interleaved_int_take)

    def call_take():
        il, ol, blk = C.c_uint32(frames), C.c_uint32(cap), C.POINTER(ctype)()
        rc = take(r._h, px, C.byref(il), C.byref(ol), C.byref(blk))
        assert rc == 0 and blk, rc
        speexhip.lib().speexhip_block_release(C.cast(blk, C.c_void_p))

    for _ in range(3):
        call_take()
    tt = []
    for _ in range(calls):
        t0 = time.perf_counter()
        call_take()
        tt.append(time.perf_counter() - t0)
    tt.sort()
    # ... and (round 6) the same call on a chunk the caller left in a pinned block of the library (speexhip_block_acquire:
    # SpeexResampler.allocChunk in JavaScript): the kernel reads it through PCIe while it writes the result block through
    # PCIe -- one launch, both directions of the link at once
    made = [0]
    tp = []
    blk_in = lib_block = None
    try:
        lib_block = speexhip.PinnedBlock(x.nbytes)
        blk_in = lib_block.array(x.dtype, x.shape)
        blk_in[...] = x
        pin_ptr = C.c_void_p(blk_in.ctypes.data)

        def call_pinned():
            il, ol, blk = C.c_uint32(frames), C.c_uint32(cap), C.POINTER(ctype)()
            rc = take(r._h, pin_ptr, C.byref(il), C.byref(ol), C.byref(blk))
            assert rc == 0 and blk, rc
            made[0] = ol.value
            speexhip.lib().speexhip_block_release(C.cast(blk, C.c_void_p))

        for _ in range(3):
            call_pinned()
        for _ in range(calls):
            t0 = time.perf_counter()
            call_pinned()
            tp.append(time.perf_counter() - t0)
        tp.sort()
    except MemoryError:
        tp = []
    finally:
        if lib_block is not None:
            lib_block.close()
    r.close()
    ts.sort()
    med = ts[len(ts) // 2]
    res = {"ms_per_chunk": round(min(med, tt[len(tt) // 2]) * 1e3, 4), "ms_min": round(min(ts[0], tt[0]) * 1e3, 4),
           "ms_per_chunk_copy_out": round(med * 1e3, 4), "ms_per_chunk_owned_block": round(tt[len(tt) // 2] * 1e3, 4),
           "input_msamples_per_s": round(used * ch / med / 1e6, 1), "calls": calls,
           "what": "one stream, a host buffer in through the C ABI's synchronous calls, %d-frame chunk, "
                   "PCIe-inclusive (not `value`): copy_out = ..._process_interleaved_* from and into the caller's pageable "
                   "buffers (H2D + kernel + D2H + wait), owned_block = ..._take (the kernel writes a pinned block the caller "
                   "then owns: what processChunk returns as an external Buffer); ms_per_chunk = the faster of the two, "
                   "both on a PAGEABLE input; pinned_in = ..._take on a chunk in a pinned block of the library "
                   "(speexhip_block_acquire / SpeexResampler.allocChunk): read in place, one launch" % frames}
    if tp:
        pin = tp[len(tp) // 2]
        res["ms_per_chunk_pinned_in"] = round(pin * 1e3, 4)
        res["ms_min_pinned_in"] = round(tp[0] * 1e3, 4)
        res["input_msamples_per_s_pinned_in"] = round(used * ch / pin / 1e6, 1)
        if peak is not None:
            es = 4 if float_io else 2
            res["pcie"] = dict(pcie_block(peak, frames * ch * es, made[0] * ch * es, pin), leg="pinned_in")
            res["pcie_pageable"] = dict(pcie_block(peak, frames * ch * es, made[0] * ch * es, min(med, tt[len(tt) // 2])),
                                        leg="ms_per_chunk")
    return res


def end_to_end_streams(speexhip, cfg, frames, mode, streams=32, calls=8, peak=None):
    """BASELINE configs[4]'s per-GPU share as a HOST caller reaches it (round 5): `streams` independent states -- what
    `streams` SpeexResampler instances of one Node process hold -- fed pageable host buffers through ONE
    speexhip_resampler_process_many_int call per step (per GPU one transfer in, one launch per <= 32 states, one
    transfer out; index.js: SpeexResamplerBatch.processChunks and the same-tick coalescer of processChunkAsync).
    PCIe-inclusive, reported beside `value`, never as it; a streaming-size point (16 384 frames per stream, the
    64 KiB chunks of the reference's stream test) beside the full-size one."""
    import ctypes as C
    import numpy as np
    ch, fi, fo, q = cfg
    lib = speexhip.lib()
    out = {"streams": streams,
           "what": "%d single-stream states, pageable host buffers in and out, one speexhip_resampler_process_many_int "
                   "call per step (PCIe-inclusive; not `value`); separate_calls = the same states through %d "
                   "speexhip_resampler_process_interleaved_int calls" % (streams, streams)}
    for label, F, n_calls in (("full", frames, calls), ("streaming", 16384, 4 * calls)):
        states = [speexhip.Resampler(ch, fi, fo, q, mode=mode) for _ in range(streams)]
        cap = wrapper_capacity(F * ch * 2, fi, fo, ch)
        xs = [np.ascontiguousarray(lcg_pcm(F * ch, 12345 + s).reshape(F, ch)) for s in range(streams)]
        ys = [np.zeros((cap, ch), np.int16) for _ in range(streams)]
        n = streams
        hs = (C.c_void_p * n)(*[st._h for st in states])
        ins = (C.c_void_p * n)(*[x.ctypes.data for x in xs])
        outs = (C.c_void_p * n)(*[y.ctypes.data for y in ys])
        il, ol, codes = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_int * n)()

        def many():
            for i in range(n):
                il[i], ol[i] = F, cap
            rc = lib.speexhip_resampler_process_many_int(n, hs, ins, il, outs, ol, codes)
            assert rc == 0, rc

        def apart():
            fn = lib.speexhip_resampler_process_interleaved_int
            for i in range(n):
                a, b = C.c_uint32(F), C.c_uint32(cap)
                rc = fn(states[i]._h, xs[i].ctypes.data_as(C.POINTER(C.c_int16)), C.byref(a),
                        ys[i].ctypes.data_as(C.POINTER(C.c_int16)), C.byref(b))
                assert rc == 0, rc

        # round 6: the chunks in pinned blocks of the library (read in place), results into pageable buffers or into
        # pinned blocks as well (what the addon's external Buffers are)
        legs = [("many", many), ("separate_calls", apart)]
        blocks = []
        try:
            bis = [speexhip.PinnedBlock(xs[0].nbytes) for _ in range(streams)]
            blocks += bis
            bos = [speexhip.PinnedBlock(cap * ch * 2) for _ in range(streams)]
            blocks += bos
            for s_ in range(streams):
                bis[s_].array(np.int16, xs[s_].shape)[...] = xs[s_]
            pin_ins = (C.c_void_p * n)(*[b.ptr for b in bis])
            pin_outs = (C.c_void_p * n)(*[b.ptr for b in bos])

            def many_with(a, b):
                def f():
                    for i in range(n):
                        il[i], ol[i] = F, cap
                    rc = lib.speexhip_resampler_process_many_int(n, hs, a, il, b, ol, codes)
                    assert rc == 0, rc
                return f
            legs += [("pinned_in", many_with(pin_ins, outs)), ("pinned_in_pinned_out", many_with(pin_ins, pin_outs))]
        except MemoryError:
            pass
        res = {}
        for name, fn in legs:
            for _ in range(2):
                fn()
            ts = []
            for _ in range(n_calls):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            res[name] = ts[len(ts) // 2]
        made_bytes = sum(int(ol[i]) for i in range(n)) * ch * 2
        for st in states:
            st.close()
        for b in blocks:
            b.close()
        out[label] = {"frames_per_stream": F, "ms_per_step": round(res["many"] * 1e3, 4),
                      "ms_per_step_separate_calls": round(res["separate_calls"] * 1e3, 4),
                      "input_msamples_per_s": round(streams * F * ch / res["many"] / 1e6, 1)}
        if "pinned_in_pinned_out" in res:
            out[label]["ms_per_step_pinned_in"] = round(res["pinned_in"] * 1e3, 4)
            out[label]["ms_per_step_pinned_in_pinned_out"] = round(res["pinned_in_pinned_out"] * 1e3, 4)
            out[label]["input_msamples_per_s_pinned"] = round(streams * F * ch / res["pinned_in_pinned_out"] / 1e6, 1)
            if peak is not None:
                big = streams * F * ch * 2 >= (32 << 20)
                out[label]["pcie"] = dict(pcie_block(peak, streams * F * ch * 2, made_bytes, res["pinned_in_pinned_out"], big),
                                          leg="pinned_in_pinned_out")
                out[label]["pcie_pageable"] = dict(pcie_block(peak, streams * F * ch * 2, made_bytes, res["many"], big), leg="ms_per_step")
    out["what"] += ("; pinned_in = the chunks in pinned blocks of the library (speexhip_block_acquire), read in place; "
                    "pinned_in_pinned_out = the results into such blocks as well (one launch per 32 states, no copy)")
    return out


def cpu_baseline(cfg, frames, budget_s=12.0, max_chunks=32):
    """Time the CPU path on this box's host cores on a bounded sample of the same workload."""
    eng, kind = _oracle_engine(cfg)
    ch, fi, fo, q = cfg
    x = lcg_pcm(frames * ch, 12345).reshape(frames, ch)
    cap = wrapper_capacity(x.size * 2, fi, fo, ch)
    eng.process(x, cap)  # warm-up chunk
    t0 = time.perf_counter()
    chunks = 0
    while chunks < max_chunks and (time.perf_counter() - t0) < budget_s:
        eng.process(x, cap)
        chunks += 1
    dt = time.perf_counter() - t0
    return {"value": round(chunks * frames * ch / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "kind": kind,
            "what": ("the reference's deps/speex/resample.c compiled natively (gcc -O2, oracle/_ref), NOT the "
                     "shipped WASM build (SURVEY measured the WASM 4x slower)") if kind == "reference"
            else "oracle/speex_oracle.c (bit-exact C restatement)",
            "sample": "%d chunks of %d frames x %d ch, same rates/quality, 1 thread, %.1f s" % (
                chunks, frames, ch, dt)}


def usable_cores():
    """Cores this process may actually use: the affinity mask and the container's CPU quota (cgroup v2 cpu.max, v1
    cfs_quota) bound os.cpu_count() -- a box that shows 256 cores to a container limited to a dozen ran 256 workers at
    0.84 Msamples/s each against 11.2 for one alone (profiles/r05_bench_total256.json, first collection)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline_workers(cfg, frames, total_streams, budget_s=10.0):
    """SURVEY 8(d), configs[4]'s CPU column: the reference is one scalar thread per stream, so T streams on a host
    with C cores run min(C, T) at a time.  Starts that many fresh CHILD processes (this process has not touched a GPU
    yet), each timing the CPU path on its own stream for ~budget_s; value = samples all of them processed / the
    slowest one's time."""
    cores = usable_cores()
    workers = max(1, min(cores, total_streams))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                               "%d,%d,%d,%d,%d,%d,%f" % (cfg + (frames, 12345 + w, budget_s))],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for w in range(workers)]
    rows = []
    for p_ in procs:
        out, _ = p_.communicate()
        if p_.returncode == 0 and out.strip():
            rows.append(json.loads(out.decode().strip().splitlines()[-1]))
    if len(rows) != workers:
        return {"error": "%d of %d CPU workers failed" % (workers - len(rows), workers)}
    ch = cfg[0]
    total = sum(r["chunks"] for r in rows) * frames * ch
    slowest = max(r["seconds"] for r in rows)
    return {"value": round(total / slowest / 1e6, 3), "unit": "Msamples/s", "cores": cores, "workers": workers,
            "cpu_count": os.cpu_count(),
            "kind": rows[0]["kind"],
            "per_worker_msamples_per_s": round(sum(r["chunks"] * frames * ch / r["seconds"] for r in rows) / workers / 1e6, 3),
            "sample": "%d worker processes (min(%d host cores, %d streams)), one stream each, %d chunks of %d frames x %d ch "
                      "in all, %.1f s (slowest worker)" % (workers, cores, total_streams, sum(r["chunks"] for r in rows),
                                                           frames, ch, slowest)}


def cpu_worker(spec):
    """child of cpu_baseline_workers: `ch,in,out,q,frames,seed,budget` -> one JSON line"""
    v = spec.split(",")
    cfg, frames, seed, budget = tuple(int(x) for x in v[:4]), int(v[4]), int(v[5]), float(v[6])
    eng, kind = _oracle_engine(cfg)
    ch, fi, fo, q = cfg
    x = lcg_pcm(frames * ch, seed).reshape(frames, ch)
    cap = wrapper_capacity(x.size * 2, fi, fo, ch)
    eng.process(x, cap)
    t0 = time.perf_counter()
    chunks = 0
    while chunks < 64 and (time.perf_counter() - t0) < budget:
        eng.process(x, cap)
        chunks += 1
    print(json.dumps({"chunks": chunks, "seconds": time.perf_counter() - t0, "kind": kind}), flush=True)


def host_legs(args):
    """Child process of the N = 1 line: the HOST-FED legs (`end_to_end`, `end_to_end_streams`) and the link's own rate
    (`pcie_peak`), in a process WITHOUT torch -- the library then runs on /opt/rocm's HIP runtime, which is what a Node
    or C caller of the drop-in loads.  (Behind torch the library shares torch's bundled runtime, under which pinned
    copies of opposite directions do not overlap: 26 + 26 GB/s where /opt/rocm's gives 43 + 43 at 4 MiB,
    profiles/r06_runtime_ab.txt -- a probe of the link there would understate it, and the legs would measure a runtime no
    host caller of the product uses.)  Prints one JSON object."""
    os.environ["SPEEXHIP_PY_NO_TORCH"] = "1"
    import speexhip
    cfg = tuple(int(v) for v in args.custom.split(",")) if args.custom else CONFIGS[args.config]
    ch, fi, fo, q = cfg
    F = args.frames
    fio = args.io == "float"
    es = 4 if fio else 2
    mode = {"exact": speexhip.MODE_EXACT, "fast": speexhip.MODE_FAST, "fast_f32": speexhip.MODE_FAST_F32,
            "fast_fixed": speexhip.MODE_FAST_FIXED}[args.mode]
    base = lcg_pcm(F * ch, 12345).reshape(F, ch)
    peak = pcie_peak(speexhip, F * ch * es)
    out = {"pcie_peak": peak, "end_to_end": end_to_end(speexhip, cfg, F, mode, fio, base, peak=peak)}
    if not fio:
        out["end_to_end_streams"] = end_to_end_streams(speexhip, cfg, F, mode, peak=peak)
    out["runtime"] = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
    print(json.dumps(out), flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the K-step timed region; the median is reported")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--custom", default=None, help="channels,in_rate,out_rate,quality (overrides --config)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("BENCH_STREAMS", "1")),
                    help="independent streams PER GPU (weak scaling)")
    ap.add_argument("--total-streams", type=int, default=int(os.environ.get("BENCH_TOTAL_STREAMS", "0")),
                    help="independent streams in the WHOLE JOB, stream s on rank s %% N (strong scaling; "
                         "256 = BASELINE configs[4]); overrides --streams")
    ap.add_argument("--frames", type=int, default=1 << 20, help="input frames per stream per step")
    ap.add_argument("--mode", default="fast_fixed", choices=["fast", "exact", "fast_f32", "fast_fixed"],
                    help="fast_fixed: the library's default -- +-1 LSB, fp64 accumulator where the reference has one (q9, "
                         "q10), bytes independent of chunking / batch size; fast: the same with tap-range shares on small "
                         "launches (bytes may depend on chunking); fast_f32: one fp32 FMA chain for every filter "
                         "(rounds 1-3); exact: the reference's arithmetic order")
    ap.add_argument("--io", default="int16", choices=["int16", "float"],
                    help="sample type of the buffers: int16 = the BASELINE metric; float = the N2 entry point")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="run the hot path for this long before the W warmup steps: the GPU leaves its idle "
                         "clocks only after some milliseconds of load (measured: the first ~30 ms of a "
                         "launch train run 20-30 %% slower), and W steps of a 15 us kernel are over before that")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)  # (child of cpu_baseline_workers)
    ap.add_argument("--host-legs", action="store_true", help=argparse.SUPPRESS)  # (child: the host-fed legs, see host_legs)
    ap.add_argument("--no-parity", action="store_true")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks, one per
    GPU, as CHILD processes (this process has not touched a GPU and never will), relay rank 0's
    JSON line, fail if any rank fails."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    # rank 0's stdout goes to a temporary file, not a pipe: nobody reads a pipe until every rank has
    # exited, and a rank 0 that printed more than the pipe holds (library warnings) would block on it
    # while the others wait for it at the next barrier
    import tempfile
    rank0_out = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, WORLD_SIZE=str(args.gpus), RANK=str(r), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=rank0_out if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # the others would wait for it at the rendezvous forever
                    q.terminate()
        time.sleep(0.05)
    rank0_out.seek(0)
    out = rank0_out.read().decode()
    sys.stdout.write(out)
    sys.stdout.flush()
    if rc != 0:
        sys.stderr.write("bench.py: a rank exited with status %d (%d ranks requested)\n" % (rc, args.gpus))
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return
    if args.host_legs:
        host_legs(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    # configs[4]'s CPU column (strong-scaling line at N = 1): min(cores, T) worker processes, started -- and finished --
    # before anything in this process touches a GPU
    cpu_many = None
    if args.gpus == 1 and args.total_streams > 0 and not args.no_cpu_baseline and "WORLD_SIZE" not in os.environ:
        cfg0 = tuple(int(v) for v in args.custom.split(",")) if args.custom else CONFIGS[args.config]
        cpu_many = cpu_baseline_workers(cfg0, args.frames, args.total_streams)

    import numpy as np
    import torch
    import dist_util
    import speexhip

    world, rank, local = dist_util.env_world()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, "
                 "or without WORLD_SIZE (bench.py then starts the ranks itself)" % (args.gpus, world, args.gpus))
    n_dev = torch.cuda.device_count()
    # BENCH_SHARE_GPU=1 (tests only): all ranks on GPU 0 with the gloo backend, so that the N-rank code
    # path -- launcher, rendezvous, barriers, MAX / SUM reductions, rank-0 line -- can be exercised
    # end to end on a 1-GPU box.  The line then says so and its numbers mean nothing.
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if local >= n_dev and not (share and n_dev >= 1):
        sys.exit("bench.py: rank %d needs GPU %d but this node shows %d GPU(s)" % (rank, local, n_dev))
    world, rank, local = dist_util.init("gloo" if share else "nccl")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctl = torch.device("cpu") if share else dev  # where the control collectives run

    cfg = CONFIGS[args.config]
    if args.custom:
        cfg = tuple(int(v) for v in args.custom.split(","))
        CONFIG_LABEL[args.config] = "custom"
    ch, fi, fo, q = cfg
    F = args.frames
    strong = args.total_streams > 0
    mine = shard_streams(args.total_streams, world, rank) if strong else [rank * args.streams + s
                                                                         for s in range(args.streams)]
    S = len(mine)
    if S == 0:
        sys.exit("bench.py: rank %d owns no stream (--total-streams %d over %d ranks)" % (rank, args.total_streams, world))
    streams_total = args.total_streams if strong else args.streams * world
    cap = wrapper_capacity(F * ch * 2, fi, fo, ch)
    mode = {"exact": speexhip.MODE_EXACT, "fast": speexhip.MODE_FAST, "fast_f32": speexhip.MODE_FAST_F32,
            "fast_fixed": speexhip.MODE_FAST_FIXED}[args.mode]
    batch = speexhip.Batch(S, ch, fi, fo, q, mode=mode)
    info = batch.info()

    # Synthetic input: LCG white noise, seed 12345 + global stream id.  To keep the
    # inputs in HBM rather than in the 256 MiB Infinity Cache, steps rotate over enough distinct
    # buffer pairs that one rotation exceeds ~600 MB of traffic.
    fio = args.io == "float"
    es = 4 if fio else 2
    in_bytes, out_bytes = S * F * ch * es, S * cap * ch * es
    nbuf = max(2, min(256, int(math.ceil(600e6 / (in_bytes + out_bytes)))))
    base = np.stack([lcg_pcm(F * ch, 12345 + sid).reshape(F, ch) for sid in mine])
    d_base = torch.from_numpy(base).to(dev)
    if fio:
        d_base = d_base.to(torch.float32) / 32768.0
    d_in = [d_base] + [torch.roll(d_base, shifts=17 * i, dims=1).contiguous() for i in range(1, nbuf)]
    d_out = [torch.zeros((S, cap, ch), dtype=torch.float32 if fio else torch.int16, device=dev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def step(i):
        b = i % nbuf
        return batch.process_device(d_in[b].data_ptr(), F * ch, F, d_out[b].data_ptr(), cap * ch, cap, sp,
                                    float_io=fio)

    # first chunk (outside the timed region); up to four streams of rank 0 are kept for the checker
    used, made = step(0)
    torch.cuda.synchronize()
    first_chunks = []
    if rank == 0 and not args.no_parity:
        for j in sorted(set([0, S // 3, (2 * S) // 3, S - 1])):
            first_chunks.append((mine[j], d_out[0][j, : made[j]].cpu().numpy(), used[j], made[j]))
    # checksum of the first step's outputs (all streams of this rank): deterministic, unlike the
    # buffers after a time-based preheat
    first_sum = int((d_out[0] * (32768.0 if fio else 1)).to(torch.int64).sum().item())

    # clock ramp (untimed, before the warmup steps): same calls as the timed loop
    t_heat, i_heat = time.perf_counter(), 0
    while (time.perf_counter() - t_heat) * 1e3 < args.preheat_ms:
        for _ in range(64):
            step(i_heat)
            i_heat += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i + 1)
    torch.cuda.synchronize()

    # R repetitions of exactly K steps; each is bracketed by barrier + synchronize on both sides
    reps = max(1, args.reps)
    wall, gpu, issue = [], [], []
    consumed = produced = 0
    it = args.warmup + 1
    for r in range(reps):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist_util.barrier(ctl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(args.steps):
            u, m = step(it)
            it += 1
            if r == 0:
                consumed += sum(u)
                produced += sum(m)
        t_issued = time.perf_counter() - t0  # the K calls have returned; the device may still be running
        ev1.record(stream)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        issue.append(t_issued)
        dist_util.barrier(ctl)
        wall.append(dist_util.reduce_scalar(elapsed, "max", ctl))
        gpu.append(ev0.elapsed_time(ev1))

    total_in_samples = dist_util.reduce_int(args.steps * S * F * ch, ctl)
    total_out_samples = dist_util.reduce_int(produced * ch, ctl)
    checksum = dist_util.reduce_int(first_sum, ctl)

    rc = 0
    if rank == 0:
        try:
            box_ghz = speexhip.device_clock()  # (after the timed region)
        except AttributeError:                 # (an older library loaded through SPEEXHIP_LIB_PATH for an A/B)
            box_ghz = (0.0, 0.0)
        elapsed_med = statistics.median(wall)
        value = total_in_samples / elapsed_med / 1e6
        # dominant kernel = the one launch per step; algorithmic bytes per launch:
        # 2*ch*consumed read + 2*ch*produced written, per stream, per call (rank 0's launch)
        launch_ms = statistics.median(gpu) / args.steps
        alg_bytes = (consumed + produced) * ch * es / args.steps
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        flops = 2.0 * info["filt_len"] * produced * ch / args.steps  # minimal form, 2*N per output
        tfl = flops / (launch_ms * 1e-3) / 1e12
        # the vector peak of the arithmetic the launch runs: fp64 for the v_fma_f64 kernels and for the exact kernels of
        # the double kinds (fp32 multiplies + fp64 adds: priced against fp64 as well)
        valu_peak = VALU64_PEAK_TFLOPS if info["accumulate_bits"] == 64 else VALU_PEAK_TFLOPS
        traffic = None
        pmc = os.path.join(ROOT, PMC_FILE)
        if os.path.exists(pmc) and not args.custom and F == 1 << 20:  # (the file holds the named workloads only)
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get("%s_s%d_%s%s" % (args.config, S, args.mode, "_float" if fio else ""))
            except Exception:
                traffic = None
        ms_per_step = elapsed_med / args.steps * 1e3
        assert launch_ms <= ms_per_step * 1.001, (launch_ms, ms_per_step)  # the launches fit in the wall time
        # what the host spends per step planning the call and putting the launch into the stream (rank 0;
        # Python + ctypes + the library's planner + hipLaunchKernel): a step cannot be shorter than this, so
        # once it reaches the kernel's own time the line measures the host, not the GPU
        host_issue_us = statistics.median(issue) / args.steps * 1e6
        line = {
            "metric": "input Msamples/s int16 %d->%d q=%d %dch (whole job)" % (fi, fo, q, ch),
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64" if info["accumulate_bits"] == 64 else "f32", "data": "synthetic",
            "config": {"workload": "%s%s: %d->%d Hz, %dch %s, q=%d, %d-frame chunk per stream per step; %s" % (
                           CONFIG_LABEL[args.config],
                           " as BASELINE configs[4] (sharded streams)" if strong and args.config == "cfg2" else "",
                           fi, fo, ch, args.io, q, F,
                           ("%d streams in the job, stream s on rank s %% %d" % (streams_total, world)) if strong
                           else ("%d stream(s) per GPU" % args.streams)),
                       "streams_total": streams_total, "streams_per_gpu": S, "frames_per_chunk": F,
                       "mode": args.mode, "io": args.io, "preheat_ms": args.preheat_ms,
                       "kernel": speexhip.KERNEL_NAMES[info["kernel"]], "fast_path": info["fast_path"],
                       "filt_len": info["filt_len"],
                       "accumulate": ("as the reference" if args.mode == "exact" else
                                      "f64: v_fma_f64 chain, exact products (reference: f64 sums of f32 products)"
                                      if info["accumulate_bits"] == 64 else
                                      "f32 FMA chain (reference: %s)" % (
                                          "f64 sums of f32 products" if info["kernel"] in (1, 3) else "f32")),
                       "parallelism": "independent streams sharded over %d rank(s), no data-path collective" % world},
            "timing": {"reps": reps, "host_issue_us_per_step": round(host_issue_us, 3),
                       "host_bound": bool(host_issue_us >= 0.95 * launch_ms * 1e3),
                       "ms_per_step_median": round(ms_per_step, 5),
                       "ms_per_step_min": round(min(wall) / args.steps * 1e3, 5),
                       "ms_per_step_max": round(max(wall) / args.steps * 1e3, 5),
                       "region": "each repetition = exactly `steps` steps between barrier+synchronize pairs, "
                                 "MAX over ranks; value uses the median repetition"},
            "output_msamples_per_s": round(total_out_samples / elapsed_med / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": ("%s (static: rocprofv3 PMC passes of this workload, tools/gpu_profile.sh; "
                                            "not measured in this run)" % PMC_FILE) if traffic is not None else None,
                         "launch_us": round(launch_ms * 1e3, 3),
                         "launch_us_min": round(min(gpu) / args.steps * 1e3, 3),
                         "launch_us_max": round(max(gpu) / args.steps * 1e3, 3),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "read_only_frac": round(consumed * ch * es / args.steps / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "vector-ALU bound, not HBM bound (SURVEY F4); see valu"},
            "valu": {"achieved": round(tfl, 2), "peak": valu_peak, "unit": "TFLOP/s",
                     "frac": round(tfl / valu_peak, 4), "flops_per_launch": int(flops),
                     "arithmetic": "fp64 vector" if valu_peak == VALU64_PEAK_TFLOPS else "fp32 vector"},
            "checksum": checksum,
            # which kind of box this was: the shader clock held under an FIR-like load (the pool's boxes differ by
            # 4-6 %: lines of different leases compare at equal clocks only)
            "box": {"shader_ghz_under_load": round(box_ghz[0], 3), "slowest_workgroup_ghz": round(box_ghz[1], 3)},
        }
        if share:
            line["config"]["parallelism"] += "; TEST RUN: all ranks share GPU 0 (BENCH_SHARE_GPU=1, gloo) -- not a measurement"
        if first_chunks:
            line["parity"], ok = parity_block(cfg, F, first_chunks, fio)
            if not ok:
                rc = 3
        if world == 1 and not args.no_cpu_baseline:
            # the host-fed legs and the link probe: a child process without torch (host_legs: the runtime a Node / C caller
            # of the drop-in loads); this process is idle meanwhile
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-legs", "--config", args.config,
                                    "--frames", str(F), "--mode", args.mode, "--io", args.io] +
                                   (["--custom", args.custom] if args.custom else []),
                                   env=dict(os.environ, SPEEXHIP_PY_NO_TORCH="1"), capture_output=True, text=True, timeout=900)
            if child.returncode == 0 and child.stdout.strip():
                legs = json.loads(child.stdout.strip().splitlines()[-1])
                line["pcie_peak"] = legs["pcie_peak"]
                line["end_to_end"] = legs["end_to_end"]
                if "end_to_end_streams" in legs:
                    line["end_to_end_streams"] = legs["end_to_end_streams"]
                line["host_legs_runtime"] = {"libamdhip64": legs["runtime"],
                                             "note": "end_to_end*, pcie_peak: measured in a child process without torch, on the HIP "
                                                     "runtime a Node / C caller of the drop-in loads; `value` and `roofline`: this "
                                                     "process (torch's bundled runtime; device-resident, no PCIe inside)"}
            else:
                line["end_to_end"] = {"error": "host-legs child failed: " + (child.stderr or "")[-400:]}
            line["cpu_baseline"] = cpu_baseline(cfg, F)
            if cpu_many is not None:
                # the strong-scaling line: T streams against min(cores, T) CPU workers; the 1-core figure stays beside it
                one = line["cpu_baseline"]
                line["cpu_baseline"] = dict(cpu_many, one_core=one)
        print(json.dumps(line), flush=True)
        if rc:
            sys.stderr.write("bench.py: PARITY FAILED: %s\n" % json.dumps(line["parity"]))
    batch.close()
    dist_util.finish()
    sys.exit(rc)


if __name__ == "__main__":
    main()
