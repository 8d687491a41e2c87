#!/usr/bin/env python3
"""tools/perf_floor.py -- the perf-regression gate's measurements (round 4).

The planners of the fast kernels are thresholds and a fitted cost model (launch_period_plan,
period_launch_prefers_w16, launch_slide): forced-variant tests keep them CORRECT, nothing kept them FAST when a
threshold moved.  profiles/perf_floor.json holds, per workload -- the BASELINE configs at 1 and 32 streams and the
rows of the channels x rate-pairs sweep that sit lowest (tools/perf_sweep.sh) -- the launch time measured on the
slowest box seen so far and a ceiling 10 % above it; tests/test_gpu_perf_gate.py re-measures (50 launches through
HIP events, best of 3 repetitions) and fails above the ceiling.

  python tools/perf_floor.py --measure            print this box's numbers (and its clock)
  python tools/perf_floor.py --measure --merge    ... and fold them into profiles/perf_floor.json: a workload's
                                                  `slowest_us` only ever grows (run on several leases)
  python tools/perf_floor.py --measure --reset    ... start the file over from this box's numbers
  ... --merge --forget a,b                        ... after dropping the records of workloads a and b (they got faster)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
FLOOR = os.path.join(ROOT, "profiles", "perf_floor.json")
MARGIN = 1.10

# name, (channels, in_rate, out_rate, quality), streams, frames per stream, mode, io
WORKLOADS = [
    ("cfg2_s1", (2, 44100, 48000, 7), 1, 1 << 20, "fast", "int16"),       # BASELINE configs[1]: the BENCH line
    ("cfg2_s32", (2, 44100, 48000, 7), 32, 1 << 20, "fast", "int16"),     # configs[4]'s per-GPU share
    ("cfg3_s1", (1, 24000, 48000, 10), 1, 1 << 20, "fast", "int16"),      # configs[2], fp64 accumulate
    ("cfg3_s32", (1, 24000, 48000, 10), 32, 1 << 20, "fast", "int16"),
    ("cfg3_f32chain_s32", (1, 24000, 48000, 10), 32, 1 << 20, "fast_f32", "int16"),
    ("cfg4_s1", (8, 48000, 44100, 5), 1, 1 << 20, "fast", "int16"),       # configs[3]
    ("cfg4_s32", (8, 48000, 44100, 5), 32, 1 << 20, "fast", "int16"),
    ("f3_s32", (1, 24000, 48000, 5), 32, 1 << 20, "fast", "int16"),
    ("cfg2_float_s32", (2, 44100, 48000, 7), 32, 1 << 20, "fast", "float"),
    ("q10_44k48k_s32", (2, 44100, 48000, 10), 32, 1 << 20, "fast", "int16"),  # period kernel, fp64 accumulate
    # the sweep's lowest rows (32 streams x 131072 frames, q7): where a planner rule gone wrong shows first
    ("sw_1ch_48k_11k", (1, 48000, 11025, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_48k_22k", (1, 48000, 22050, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_96k_44k", (1, 96000, 44100, 7), 32, 131072, "fast", "int16"),
    ("sw_3ch_48k_11k", (3, 48000, 11025, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_44k_32k", (1, 44100, 32000, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_44k_8k", (1, 44100, 8000, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_44k_16k", (1, 44100, 16000, 7), 32, 131072, "fast", "int16"),
    ("sw_3ch_44k_32k", (3, 44100, 32000, 7), 32, 131072, "fast", "int16"),
    ("sw_2ch_48k_11k", (2, 48000, 11025, 7), 32, 131072, "fast", "int16"),
    ("sw_4ch_48k_11k", (4, 48000, 11025, 7), 32, 131072, "fast", "int16"),
    ("sw_3ch_44k_16k", (3, 44100, 16000, 7), 32, 131072, "fast", "int16"),
    ("sw_2ch_44k_8k", (2, 44100, 8000, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_32k_44k", (1, 32000, 44100, 7), 32, 131072, "fast", "int16"),
    ("sw_3ch_44k_8k", (3, 44100, 8000, 7), 32, 131072, "fast", "int16"),
    ("sw_2ch_44k_32k", (2, 44100, 32000, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_96k_48k", (1, 96000, 48000, 7), 32, 131072, "fast", "int16"),
    ("sw_2ch_48k_22k", (2, 48000, 22050, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_48k_44k", (1, 48000, 44100, 7), 32, 131072, "fast", "int16"),
    ("sw_1ch_48k_8k", (1, 48000, 8000, 7), 32, 131072, "fast", "int16"),
    ("sw_2ch_48k_8k", (2, 48000, 8000, 7), 32, 131072, "fast", "int16"),
    ("sw_6ch_44k_8k", (6, 44100, 8000, 7), 32, 131072, "fast", "int16"),
    ("sw_3ch_48k_22k", (3, 48000, 22050, 7), 32, 131072, "fast", "int16"),
    ("sw_2ch_48k_16k", (2, 48000, 16000, 7), 32, 131072, "fast", "int16"),
    # several generations of wide windows: phase pairs with tap-range shares, rows fetched behind the window
    ("big_3ch_48k_11k", (3, 48000, 11025, 7), 32, 1 << 20, "fast", "int16"),
    ("big_2ch_48k_11k", (2, 48000, 11025, 7), 32, 1 << 20, "fast", "int16"),
    ("big_3ch_44k_16k", (3, 44100, 16000, 7), 32, 1 << 20, "fast", "int16"),
    # round 5: frames of five / seven / three channels on their ISA loops, the wide window that left the exact kernel
    ("sw_5ch_48k_11k", (5, 48000, 11025, 7), 32, 131072, "fast", "int16"),
    ("sw_7ch_48k_11k", (7, 48000, 11025, 7), 32, 131072, "fast", "int16"),
    ("sw_7ch_44k_48k", (7, 44100, 48000, 7), 32, 131072, "fast", "int16"),
    ("big_3ch_44k_48k", (3, 44100, 48000, 7), 32, 1 << 20, "fast", "int16"),
    ("sw_1ch_96k_11k", (1, 96000, 11025, 7), 32, 131072, "fast", "int16"),
    ("q10_5ch_44k48k_s32", (5, 44100, 48000, 10), 32, 262144, "fast", "int16"),
    # launches of one generation with their own plans (tap-range shares, r = 5 shares)
    ("one_48k_11k_2ch", (2, 48000, 11025, 7), 1, 441000, "fast", "int16"),
    ("one_48k_8k_2ch", (2, 48000, 8000, 7), 1, 441000, "fast", "int16"),
    ("one_44k_48k_mono", (1, 44100, 48000, 7), 1, 1 << 20, "fast", "int16"),
    # round 6: the DEFAULT mode (fast_fixed: no tap-range shares) on the BASELINE configs and on the one-stream decimator
    # that pays for it
    ("cfg2_s1_default", (2, 44100, 48000, 7), 1, 1 << 20, "fast_fixed", "int16"),
    ("cfg2_s32_default", (2, 44100, 48000, 7), 32, 1 << 20, "fast_fixed", "int16"),
    ("cfg3_s1_default", (1, 24000, 48000, 10), 1, 1 << 20, "fast_fixed", "int16"),
    ("cfg4_s1_default", (8, 48000, 44100, 5), 1, 1 << 20, "fast_fixed", "int16"),
    ("cfg4_s32_default", (8, 48000, 44100, 5), 32, 1 << 20, "fast_fixed", "int16"),
    ("one_48k_11k_2ch_default", (2, 48000, 11025, 7), 1, 441000, "fast_fixed", "int16"),
]


def measure(workload, launches=50, reps=3, preheat_ms=150.0):
    """best-of-`reps` average launch time (us) of `launches` back-to-back launches, HIP events on the launch stream"""
    import numpy as np
    import torch
    import speexhip
    name, (ch, fi, fo, q), S, F, mode, io = workload
    fio = io == "float"
    cap = int(F * fo / fi) + 1024
    b = speexhip.Batch(S, ch, fi, fo, q, mode={"fast": speexhip.MODE_FAST, "fast_f32": speexhip.MODE_FAST_F32,
                                              "exact": speexhip.MODE_EXACT, "fast_fixed": speexhip.MODE_FAST_FIXED}[mode])
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randint(-20000, 20000, (S, F, ch), generator=g, dtype=torch.int16)
    nbuf = 3
    d_in = [(x.roll(7 * i, 1).cuda().float() / 32768.0) if fio else x.roll(7 * i, 1).cuda() for i in range(nbuf)]
    d_out = [torch.zeros((S, cap, ch), dtype=torch.float32 if fio else torch.int16, device="cuda") for _ in range(nbuf)]
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def step(i):
        k = i % nbuf
        b.process_device(d_in[k].data_ptr(), F * ch, F, d_out[k].data_ptr(), cap * ch, cap, sp, float_io=fio)

    t0, i = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < preheat_ms:
        for _ in range(16):
            step(i)
            i += 1
        torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(launches):
            step(i)
            i += 1
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / launches
        best = us if best is None else min(best, us)
    info = b.info()
    b.close()
    del d_in, d_out
    return best, info["fast_path"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--measure", action="store_true")
    ap.add_argument("--merge", action="store_true")
    ap.add_argument("--reset", action="store_true")
    ap.add_argument("--only", default=None, help="substring of the workload names to run")
    ap.add_argument("--forget", default=None,
                    help="comma-separated workload names whose recorded numbers are dropped first (a kernel got faster: "
                         "its old slowest time is no floor any more)")
    ap.add_argument("--forget-faster", type=float, default=None, metavar="RATIO",
                    help="with --merge: a workload measured below RATIO x its recorded slowest time starts over from this "
                         "box's number (a kernel or a launch rule got faster; e.g. 0.85)")
    ap.add_argument("--out", default=FLOOR, help="where to write (gpurun brings back gpurun_out/ only)")
    args = ap.parse_args()
    if not args.measure:
        ap.error("nothing to do")
    import speexhip
    ghz, ghz_min = speexhip.device_clock()
    print("# box: shader clock under load %.3f GHz (slowest workgroup %.3f)" % (ghz, ghz_min))
    floor = {"margin": MARGIN, "boxes": [], "workloads": {}}
    if os.path.exists(FLOOR) and not args.reset:
        floor = json.load(open(FLOOR))
    for name in (args.forget or "").split(","):
        floor["workloads"].pop(name.strip(), None)
    rows = {}
    for w in WORKLOADS:
        if args.only and args.only not in w[0]:
            continue
        us, path = measure(w)
        rows[w[0]] = us
        old = floor["workloads"].get(w[0], {})
        print("%-22s %9.2f us   path %d   (slowest so far %s, ceiling %s)" % (
            w[0], us, path, old.get("slowest_us"), old.get("ceiling_us")))
        if args.merge or args.reset:
            if args.forget_faster and old.get("slowest_us") and us < args.forget_faster * old["slowest_us"]:
                print("   (faster than %.2f x the record: starting over)" % args.forget_faster)
                old = {}
            slowest = max(us, old.get("slowest_us", 0.0))
            floor["workloads"][w[0]] = {"config": list(w[1]), "streams": w[2], "frames": w[3], "mode": w[4], "io": w[5],
                                        "fast_path": path, "slowest_us": round(slowest, 2),
                                        "fastest_us": round(min(us, old.get("fastest_us", 1e30)), 2),
                                        "ceiling_us": round(slowest * MARGIN, 2)}
    if args.merge or args.reset:
        floor["boxes"].append({"ghz": round(ghz, 3), "ghz_min": round(ghz_min, 3),
                               "commit": os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip()})
        json.dump(floor, open(args.out, "w"), indent=1, sort_keys=True)
        print("# wrote", args.out)


if __name__ == "__main__":
    main()
