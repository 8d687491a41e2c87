#!/usr/bin/env python3
"""bench.py -- throughput of the resampler hot path on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch of synthetic input: every stream owned by
a rank resamples its next chunk (default: BASELINE.json configs[1] -- 44100->48000 Hz, 2 ch
int16, q=7, one 2^20-frame chunk per stream per step, ONE stream per GPU).  Inputs and outputs
are resident in HBM when the timed region starts; the streams are stateful (each step continues
the previous one, as processChunk calls do).  One process per GPU; no data-path collective
(streams are independent): torch.distributed (RCCL) only carries the barrier, the MAX of the
elapsed times and a checksum/sample-count SUM.

Prints ONE JSON line on rank 0 (contract in the task description): value = whole-job input
Msamples/s, plus "roofline" (algorithmic HBM bytes per launch / measured launch time vs the
8 TB/s HBM peak) and, at N=1, "cpu_baseline" (the reference's own C timed on the host cores).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))

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


def cpu_baseline(cfg, frames, gpu_first_chunk=None, budget_s=12.0, max_chunks=32):
    """The CPU leg (the ONLY place bench.py touches oracle/): time the CPU path on this box's host
    cores on a bounded sample of the same workload -- oracle/_ref (the reference's own C, kind
    "reference") when present, else the restatement ("port") -- and, since its warm-up chunk is
    the very chunk the GPU processed first, use it as the checker of that chunk (parity)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    ch, fi, fo, q = cfg
    kind = "reference" if orc.have_reference() else "port"
    eng = (orc.Reference if kind == "reference" else orc.Oracle)(ch, fi, fo, q)
    x = lcg_pcm(frames * ch, 12345).reshape(frames, ch)
    cap = wrapper_capacity(x.size * 2, fi, fo, ch)
    want, want_used = eng.process(x, cap)  # warm-up chunk == the GPU's first chunk
    parity = None
    if gpu_first_chunk is not None:
        got, used, made = gpu_first_chunk
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32)) if got.shape == want.shape else None
        parity = {"checked_frames": int(want.shape[0]),
                  "max_abs_diff_lsb": int(diff.max()) if diff is not None else -1,
                  "mismatch_rate": float((diff != 0).mean()) if diff is not None else 1.0,
                  "counters_equal": bool(used == want_used and made == want.shape[0])}
    t0 = time.perf_counter()
    chunks = 0
    while chunks < max_chunks and (time.perf_counter() - t0) < budget_s:
        eng.process(x, cap)
        chunks += 1
    dt = time.perf_counter() - t0
    return {"value": round(chunks * frames * ch / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "kind": kind,
            "sample": "%d chunks of %d frames x %d ch, same rates/quality, 1 thread, %.1f s" % (
                chunks, frames, ch, dt)}, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--custom", default=None, help="channels,in_rate,out_rate,quality (overrides --config)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("BENCH_STREAMS", "1")), help="independent streams per GPU (configs[4] uses 32)")
    ap.add_argument("--frames", type=int, default=1 << 20, help="input frames per stream per step")
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    ap.add_argument("--io", default="int16", choices=["int16", "float"],
                    help="sample type of the buffers: int16 = the BASELINE metric; float = the N2 entry point")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="run the hot path for this long before the W warmup steps: the GPU leaves its idle "
                         "clocks only after some milliseconds of load (measured: the first ~30 ms of a "
                         "launch train run 20-30 %% slower), and W steps of a 15 us kernel are over before that")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import dist_util
    import speexhip

    world, rank, local = dist_util.init("nccl")
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = CONFIGS[args.config]
    if args.custom:
        cfg = tuple(int(v) for v in args.custom.split(","))
        CONFIG_LABEL[args.config] = "custom"
    ch, fi, fo, q = cfg
    S, F = args.streams, args.frames
    cap = wrapper_capacity(F * ch * 2, fi, fo, ch)
    mode = speexhip.MODE_EXACT if args.mode == "exact" else speexhip.MODE_FAST
    batch = speexhip.Batch(S, ch, fi, fo, q, mode=mode)
    info = batch.info()

    # Synthetic input: LCG white noise, seed 12345 + global stream id.  To keep the
    # inputs in HBM rather than in the 256 MiB Infinity Cache, steps rotate over enough distinct
    # buffer pairs that one rotation exceeds ~600 MB of traffic.
    fio = args.io == "float"
    es = 4 if fio else 2
    in_bytes, out_bytes = S * F * ch * es, S * cap * ch * es
    nbuf = max(2, min(256, int(math.ceil(600e6 / (in_bytes + out_bytes)))))
    base = np.stack([lcg_pcm(F * ch, 12345 + rank * S + s).reshape(F, ch) for s in range(S)])
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

    # first chunk (outside the timed region); kept for the CPU leg's parity check
    used, made = step(0)
    torch.cuda.synchronize()
    first_chunk = (d_out[0][0, : made[0]].cpu().numpy(), used[0], made[0]) if rank == 0 and not fio else None
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
    dist_util.barrier(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    consumed = produced = 0
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        u, m = step(args.warmup + 1 + i)
        consumed += sum(u)
        produced += sum(m)
    ev1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dist_util.barrier(dev)
    gpu_ms = ev0.elapsed_time(ev1)

    elapsed_max = dist_util.reduce_scalar(elapsed, "max", dev)
    total_in_samples = dist_util.reduce_int(args.steps * S * F * ch, dev)
    total_out_samples = dist_util.reduce_int(produced * ch, dev)
    checksum = dist_util.reduce_int(first_sum, dev)

    if rank == 0:
        value = total_in_samples / elapsed_max / 1e6
        # dominant kernel = the one launch per step; algorithmic bytes per launch:
        # 2*ch*consumed read + 2*ch*produced written, per stream, per call
        launch_ms = gpu_ms / args.steps
        alg_bytes = (consumed + produced) * ch * es / args.steps
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        flops = 2.0 * info["filt_len"] * produced * ch / args.steps  # minimal form, 2*N per output
        tfl = flops / (launch_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get("%s_s%d_%s" % (args.config, S, args.mode))
            except Exception:
                traffic = None
        line = {
            "metric": "input Msamples/s int16 %d->%d q=%d %dch (whole job)" % (fi, fo, q, ch),
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s per GPU x %d stream(s): %d->%d Hz, %dch int16, "
                                   "q=%d, %d-frame chunk per stream per step" % (CONFIG_LABEL[args.config], S, fi,
                                                                                 fo, ch, q, F),
                       "streams_per_gpu": S, "frames_per_chunk": F, "mode": args.mode, "io": args.io,
                       "preheat_ms": args.preheat_ms,
                       "kernel": speexhip.KERNEL_NAMES[info["kernel"]], "fast_path": info["fast_path"],
                       "filt_len": info["filt_len"], "parallelism": "streams sharded, %d rank(s)" % world},
            "output_msamples_per_s": round(total_out_samples / elapsed_max / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launch_us": round(launch_ms * 1e3, 3), "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "fp32 vector-ALU bound, not HBM bound (SURVEY F4); see valu"},
            "valu": {"achieved": round(tfl, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tfl / VALU_PEAK_TFLOPS, 4), "flops_per_launch": int(flops)},
            "checksum": checksum,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], parity = cpu_baseline(cfg, F, None if args.no_parity else first_chunk)
            if parity is not None:
                line["parity"] = parity
                assert parity["counters_equal"] and 0 <= parity["max_abs_diff_lsb"] <= 1, parity
        print(json.dumps(line), flush=True)
    batch.close()
    dist_util.finish()


if __name__ == "__main__":
    main()
