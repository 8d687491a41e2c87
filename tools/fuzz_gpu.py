#!/usr/bin/env python3
"""Differential fuzz of libspeexhip against the oracle on one MI355X (test infrastructure: this is a
checker run, the oracle is never part of the product path).

Random (channels, in_rate, out_rate, quality) -- common audio rates, rates that reduce to small
ratios (the slide kernel's shapes), near-unity and coprime oddballs (long periods, the exact kernel) --
driven through a random call sequence: chunk sizes from 0 to a few hundred thousand frames, output
capacities that are sometimes too small (unconsumed input, reference resample.c:1061-1082), int16 and
float calls mixed on one state, per-channel calls with strides (after which the interleaved calls run
channel by channel), runs of coalesced chunks, occasional set_rate / set_quality / skip_zeros / reset_mem
in between.
EXACT mode must be bit-identical (samples, counters, positions); FAST mode within +-1 LSB (int16) /
the float bound of tests/test_gpu_parity.py, counters and positions identical.  A trial ends at a
set_rate that returns RESAMPLER_ERR_OVERFLOW in both (the one documented deviation: DESIGN 3.5).

With --batch every third trial drives 1-48 ragged streams of one configuration through the batched
device-pointer call instead.

usage: python tools/fuzz_gpu.py [--seconds 240] [--seed 1] [--max-frames 300000] [--batch]
Prints one line per failing trial (with the seed that reproduces it) and a summary; exit code 1 on
any failure.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
import oracle as orc  # noqa: E402
import speexhip  # noqa: E402

COMMON = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000]


def pick_rates(rng):
    kind = rng.randint(0, 10)
    if kind < 5:
        return int(rng.choice(COMMON)), int(rng.choice(COMMON))
    if kind < 7:  # small ratios a:b times a base
        a, b = int(rng.choice(list(range(1, 13)) + [16, 20, 24])), int(rng.randint(1, 13))
        base = int(rng.choice([1000, 4000, 8000, 11025]))
        return a * base, b * base
    if kind < 9:  # near unity / oddballs with long periods
        r = int(rng.choice(COMMON))
        return r, r + int(rng.randint(-300, 301)) or 1
    return int(rng.randint(2000, 200000)), int(rng.randint(2000, 200000))


def signal(rng, frames, ch, as_float):
    kind = rng.randint(0, 4)
    if kind == 0:
        x = rng.randint(-32768, 32768, size=(frames, ch)).astype(np.int16)  # full-scale noise (saturates)
    elif kind == 1:
        t = np.arange(frames)[:, None] * (0.01 + 0.3 * rng.rand(1, ch))
        x = (np.sin(t) * rng.randint(100, 32767)).astype(np.int16)
    elif kind == 2:
        x = np.zeros((frames, ch), np.int16)
        if frames:
            x[rng.randint(0, frames, size=max(1, frames // 50))] = rng.choice([-32768, 32767])
    else:
        x = (rng.randn(frames, ch) * 3000).clip(-32768, 32767).astype(np.int16)
    if as_float:
        return (x.astype(np.float32) * np.float32(1.0 + rng.rand())).astype(np.float32)
    return x


def one_trial(seed, max_frames, many_channels=False):
    rng = np.random.RandomState(seed)
    ch = int(rng.choice([1, 1, 2, 2, 2, 3, 4, 5, 6, 7, 8] + ([9, 10, 10, 12, 12, 16, 16, 24, 32, 64, 65, 100] if many_channels else [])))
    i, o = pick_rates(rng)
    q = int(rng.randint(0, 11))
    mode = speexhip.MODE_EXACT if rng.rand() < 0.35 else speexhip.MODE_FAST
    what = "seed=%d ch=%d %d->%d q=%d %s" % (seed, ch, i, o, q, "exact" if mode == speexhip.MODE_EXACT else "fast")
    try:
        ref = orc.Oracle(ch, i, o, q)
    except Exception:
        ref = None
    try:
        got_r = speexhip.Resampler(ch, i, o, q, mode=mode)
    except Exception as e:
        if ref is None:
            return None, what + " (both refuse)"
        return "library refused a configuration the oracle accepts: %r" % (e,), what
    if ref is None:
        got_r.close()
        return "library accepted a configuration the oracle refuses", what
    n_calls = int(rng.randint(1, 7))
    frames_done = 0
    peak_in = 1.0  # largest input magnitude so far (what the filter memory may hold)
    for call in range(n_calls):
        ctl = rng.rand()
        if call and ctl < 0.08:
            q2 = int(rng.randint(0, 11))
            if ref.set_quality(q2) != got_r.set_quality(q2):
                return "set_quality(%d): return codes differ" % q2, what
        elif call and ctl < 0.16:
            i2, o2 = pick_rates(rng)
            rc_ref, rc_got = ref.set_rate(i2, o2), got_r.set_rate(i2, o2)
            if rc_ref != rc_got:
                return "set_rate(%d,%d): oracle rc %d, library rc %d" % (i2, o2, rc_ref, rc_got), what
            if rc_ref == 5:
                # RESAMPLER_ERR_OVERFLOW: the reference returns with its state half-updated (new rates, old
                # filter: resample.c:1130-1136 return before update_filter), the library with its state
                # untouched (documented deviation) -- nothing comparable after this point
                break
        elif call and ctl < 0.20:
            ref.skip_zeros()
            got_r.skip_zeros()
        elif call and ctl < 0.23:
            ref.reset_mem()
            got_r.reset_mem()
        as_float = rng.rand() < 0.3
        size_kind = rng.randint(0, 6)
        frames = [0, 1, int(rng.randint(2, 200)), int(rng.randint(200, 5000)), int(rng.randint(5000, 60000)) if max_frames >= 60000 else int(rng.randint(200, max_frames + 1)),
                  int(rng.randint(min(60000, max_frames), max_frames + 1))][size_kind]
        frames = min(frames, max(0, max_frames * 2 - frames_done))
        frames_done += frames
        x = signal(rng, frames, ch, as_float)
        if x.size:
            peak_in = max(peak_in, float(np.abs(x.astype(np.float64)).max()))
        in_r, out_r = ref.rate()
        full = int(frames * out_r / max(in_r, 1)) + 64
        cap = full if rng.rand() < 0.7 else int(rng.randint(0, full + 1))
        kind = "float" if as_float else "int"
        uniform = len(set(ref.positions())) == 1
        op = rng.rand()
        if op < 0.15 and frames <= 40000:
            # one channel through the per-channel entry point with strides (resample.c:927-1036, 1170-1188):
            # the exact kernel in either mode, so the whole sentinel-filled buffers must be identical
            c = int(rng.randint(0, ch))
            ins, outs = int(rng.randint(1, 4)), int(rng.randint(1, 4))
            xc = x[:, c] if frames else x.reshape(-1)[:0]
            a = ref.channel_call(kind, c, xc, cap, ins, outs)
            b = got_r.channel_call(kind, c, xc, cap, ins, outs)
            tag = "call %d (channel %d, %s, %d frames, cap %d, strides %d/%d)" % (call, c, kind, frames, cap, ins, outs)
            if a[:3] != b[:3]:
                return "%s: rc/used/made %s, oracle %s" % (tag, b[:3], a[:3]), what
            if not np.array_equal(a[3], b[3]):
                return "%s: buffers differ in %d places" % (tag, int((a[3] != b[3]).sum())), what
            if got_r.positions() != ref.positions():
                return "%s: positions %s, oracle %s" % (tag, got_r.positions(), ref.positions()), what
            continue
        if op < 0.27 and uniform and not as_float and frames >= 4:
            # 2-4 consecutive calls as one launch (speexhip_resampler_process_chunks_int) against the
            # oracle's separate calls, each with a capacity of its own
            n = int(rng.randint(2, 5))
            cuts = sorted(set(int(v) for v in rng.randint(0, frames + 1, size=n - 1)))
            parts = np.split(x, cuts)
            caps = [int(len(p_) * out_r / max(in_r, 1)) + 8 if rng.rand() < 0.7 else int(rng.randint(0, len(p_) + 9))
                    for p_ in parts]
            wants = [ref.process(p_, c_) for p_, c_ in zip(parts, caps)]
            gouts, gused = got_r.process_chunks(parts, caps)
            tag = "call %d (%d coalesced chunks of %s frames, caps %s)" % (call, len(parts), [len(p_) for p_ in parts], caps)
            for i, ((w, wu), g, gu) in enumerate(zip(wants, gouts, gused)):
                if gu != wu or g.shape != w.shape:
                    return "%s: chunk %d consumed/produced %d/%d, oracle %d/%d" % (tag, i, gu, g.shape[0], wu, w.shape[0]), what
                if g.size == 0:
                    continue
                d = np.abs(g.astype(np.int32) - w.astype(np.int32))
                if d.max() > (0 if mode == speexhip.MODE_EXACT else 1):
                    return "%s: chunk %d off by %d LSB" % (tag, i, d.max()), what
            if tuple(got_r.position()) != tuple(ref.position()):
                return "%s: position %s, oracle %s" % (tag, got_r.position(), ref.position()), what
            continue
        if not uniform:
            # channels stand apart (per-channel calls before): interleaved calls go channel by channel on
            # the exact kernel -- whole buffers identical in either mode (resample.c:1061-1082)
            a = ref.raw_call(kind, x, cap)
            b = got_r.raw_call(kind, x, cap)
            tag = "call %d (split %s, %d frames, cap %d)" % (call, kind, frames, cap)
            if a[:3] != b[:3]:
                return "%s: rc/used/made %s, oracle %s" % (tag, b[:3], a[:3]), what
            if not np.array_equal(a[3], b[3]):
                return "%s: buffers differ in %d places" % (tag, int((a[3] != b[3]).sum())), what
            if got_r.positions() != ref.positions():
                return "%s: positions %s, oracle %s" % (tag, got_r.positions(), ref.positions()), what
            continue
        if as_float:
            want, wu = ref.process_float(x, cap)
            got, gu = got_r.process_float(x, cap)
        else:
            want, wu = ref.process(x, cap)
            got, gu = got_r.process(x, cap)
        tag = "call %d (%s, %d frames, cap %d)" % (call, "float" if as_float else "int16", frames, cap)
        if gu != wu or got.shape != want.shape:
            return "%s: consumed/produced %d/%d, oracle %d/%d" % (tag, gu, got.shape[0], wu, want.shape[0]), what
        if tuple(got_r.position()) != tuple(ref.position()):
            return "%s: position %s, oracle %s" % (tag, got_r.position(), ref.position()), what
        if got.size == 0:
            continue
        if mode == speexhip.MODE_EXACT:
            if not np.array_equal(got, want):
                bad = np.argwhere(got != want)
                return "%s: EXACT differs in %d samples, first at %s" % (tag, len(bad), bad[0]), what
        elif as_float:
            scale = max(1.0, float(np.abs(want).max()), peak_in)
            err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
            # 8x the bound measured on unit-scale signals (inputs here reach 65534), growing like sqrt(taps)
            if err > 4e-6 * scale * 8 * max(1.0, (ref.taps / 256.0) ** 0.5):
                return "%s: float error %.3g at scale %.3g" % (tag, err, scale), what
        else:
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            if d.max() > 1:
                return "%s: %d LSB at %s" % (tag, d.max(), np.unravel_index(np.argmax(d), d.shape)), what
            # share of samples that differ at all (each by exactly 1): ~E|difference of the two fp32 rounding
            # errors| in LSB, growing like sqrt(taps): measured up to 3.1e-3 for filters of <= 256 taps,
            # 5-8e-3 for decimation filters of 560-1416 taps, 1.3e-2 at 2824 taps (full-scale noise)
            if got.size >= 20000 and (d != 0).mean() > 5e-3 * max(1.0, 1.2 * (ref.taps / 256.0) ** 0.5):
                return "%s: %.2e of samples differ (%d taps)" % (tag, (d != 0).mean(), ref.taps), what
    got_r.close()
    return None, what


def batch_trial(seed, max_frames):
    """Many independent streams of one configuration through speexhip.Batch.process_device (device
    pointers, one launch per call): ragged lengths and capacities, 1-3 consecutive calls, every stream
    checked against an oracle state of its own."""
    import torch
    rng = np.random.RandomState(seed)
    ch = int(rng.choice([1, 2, 2, 3, 4, 6, 8]))
    i, o = pick_rates(rng)
    q = int(rng.randint(0, 11))
    S = int(rng.choice([1, 2, 5, 8, 9, 17, 33, 48]))
    mode = speexhip.MODE_EXACT if rng.rand() < 0.3 else speexhip.MODE_FAST
    as_float = rng.rand() < 0.25
    what = "seed=%d BATCH S=%d ch=%d %d->%d q=%d %s %s" % (seed, S, ch, i, o, q, "exact" if mode == speexhip.MODE_EXACT else "fast",
                                                           "float" if as_float else "int16")
    try:
        refs = [orc.Oracle(ch, i, o, q) for _ in range(S)]
    except Exception:
        return None, what + " (oracle refuses)"
    if refs[0].den > 4000 and refs[0].taps > 300:
        return None, what + " (skipped: slow on the oracle)"
    b = speexhip.Batch(S, ch, i, o, q, mode=mode)
    # (one trial in eight is big enough for launches of several generations of workgroups: other tile
    #  shapes, the mono image stores, eight-wave slide workgroups)
    budget = 24e6 if rng.rand() < 0.125 else 3e6
    fmax = max(16, min(max_frames if budget < 4e6 else 1 << 20, int(budget / (S * ch))))
    dt = np.float32 if as_float else np.int16
    peak = [1.0] * S  # largest input magnitude a stream has seen: what its filter memory may hold
    for call in range(int(rng.randint(1, 4))):
        F = int(rng.randint(1, fmax + 1))
        lens = [F if rng.rand() < 0.5 else int(rng.randint(0, F + 1)) for _ in range(S)]
        full = int(F * refs[0].rate()[1] / max(refs[0].rate()[0], 1)) + 64
        caps = [full if rng.rand() < 0.8 else int(rng.randint(0, full + 1)) for _ in range(S)]
        x = np.stack([signal(rng, F, ch, as_float) for _ in range(S)])
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros((S, full, ch), dtype=torch.float32 if as_float else torch.int16, device="cuda")
        used, made = b.process_device(d_in.data_ptr(), F * ch, lens, d_out.data_ptr(), full * ch, caps,
                                      torch.cuda.current_stream().cuda_stream, float_io=as_float)
        torch.cuda.synchronize()
        out = d_out.cpu().numpy()
        for s_ in range(S):
            want, wu = (refs[s_].process_float if as_float else refs[s_].process)(x[s_, : lens[s_]].astype(dt), caps[s_])
            if lens[s_]:
                peak[s_] = max(peak[s_], float(np.abs(x[s_, : lens[s_]]).max()))
            tag = "call %d stream %d (%d of %d frames, cap %d)" % (call, s_, lens[s_], F, caps[s_])
            if used[s_] != wu or made[s_] != want.shape[0]:
                return "%s: consumed/produced %d/%d, oracle %d/%d" % (tag, used[s_], made[s_], wu, want.shape[0]), what
            got = out[s_, : made[s_]]
            if got.size == 0:
                continue
            if mode == speexhip.MODE_EXACT:
                if not np.array_equal(got, want):
                    return "%s: EXACT differs in %d samples" % (tag, int((got != want).sum())), what
            elif as_float:
                # (scale: the output, or what is still in the filter's memory -- a quiet call after a loud one
                #  is the difference of large terms; the bound grows like sqrt(taps) as the int16 rate does)
                scale = max(float(np.abs(want).max()), peak[s_])
                err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
                if err > 4e-6 * scale * 8 * max(1.0, (refs[s_].taps / 256.0) ** 0.5):
                    return "%s: float error %.3g at scale %.3g" % (tag, err, scale), what
            else:
                d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                if d.max() > 1:
                    return "%s: %d LSB" % (tag, d.max()), what
    b.close()
    return None, what


def many_trial(seed, max_frames):
    """Round 5: 2-40 single-stream states of 1-3 configurations through speexhip_resampler_process_many_int / _float
    (host buffers, one call per step: per device one transfer in, one launch per <= 32 states of one filter, one
    transfer out), modes mixed between the states, 1-4 steps of ragged lengths and capacities, NULL inputs, now and then
    a state named twice and a set_quality between two steps; every state against an oracle state of its own."""
    rng = np.random.RandomState(seed)
    as_float = rng.rand() < 0.25
    kinds = []
    for _ in range(int(rng.randint(1, 4))):
        i, o = pick_rates(rng)
        kinds.append((int(rng.choice([1, 2, 2, 3, 4, 8])), i, o, int(rng.randint(0, 11))))
    S = int(rng.choice([2, 3, 7, 16, 33, 40]))
    cfg = [kinds[int(rng.randint(0, len(kinds)))] for _ in range(S)]
    modes = [speexhip.MODE_EXACT if rng.rand() < 0.3 else (speexhip.MODE_FAST_FIXED if rng.rand() < 0.3 else speexhip.MODE_FAST) for _ in range(S)]
    what = "seed=%d MANY S=%d kinds=%s %s" % (seed, S, kinds, "float" if as_float else "int16")
    try:
        refs = [orc.Oracle(*c) for c in cfg]
    except Exception:
        return None, what + " (oracle refuses)"
    if any(r.den > 4000 and r.taps > 300 for r in refs):
        return None, what + " (skipped: slow on the oracle)"
    states = [speexhip.Resampler(*c, mode=m) for c, m in zip(cfg, modes)]
    dt = np.float32 if as_float else np.int16
    big = rng.rand() < 0.1   # one trial in ten crosses 32 MB: the pipelined large path
    fmax = max(16, min(max_frames, int((48e6 if big else 4e6) / (S * 2))))
    peak = [1.0] * S
    for step in range(int(rng.randint(1, 5))):
        order = list(range(S))
        if rng.rand() < 0.2:
            order.append(int(rng.randint(0, S)))   # a state twice in one call
        chunks, caps, frames = [], [], []
        for s_ in order:
            ch, i, o, q = cfg[s_]
            F = int(rng.randint(0, fmax + 1)) if rng.rand() < 0.8 else int(rng.choice([0, 1, 160, 480]))
            full = int(F * o / max(i, 1)) + 64
            cap = full if rng.rand() < 0.8 else int(rng.randint(0, full + 1))
            if rng.rand() < 0.05:
                chunks.append(None)
                caps.append((F, cap))
            else:
                chunks.append(signal(rng, F, ch, as_float))
                caps.append(cap)
            frames.append(F)
        outs, used, codes = speexhip.process_many([states[s_] for s_ in order], chunks, caps, dtype=dt)
        for j, s_ in enumerate(order):
            ref = refs[s_]
            if chunks[j] is None:
                want, wu = (ref.process_float if as_float else ref.process)(None, caps[j][1], null_frames=caps[j][0])
            else:
                want, wu = (ref.process_float if as_float else ref.process)(chunks[j], caps[j])
                if frames[j]:
                    peak[s_] = max(peak[s_], float(np.abs(chunks[j]).max()))
            tag = "step %d entry %d state %d %s mode %d (%d frames)" % (step, j, s_, cfg[s_], modes[s_], frames[j])
            if codes[j] != 0 or used[j] != wu or outs[j].shape[0] != want.shape[0]:
                return "%s: code %d consumed/produced %d/%d, oracle %d/%d" % (tag, codes[j], used[j], outs[j].shape[0], wu, want.shape[0]), what
            got = outs[j]
            if got.size == 0:
                continue
            if modes[s_] == speexhip.MODE_EXACT:
                if not np.array_equal(got, want):
                    return "%s: EXACT differs in %d samples" % (tag, int((got != want).sum())), what
            elif as_float:
                scale = max(float(np.abs(want).max()), peak[s_])
                err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
                if err > 4e-6 * scale * 8 * max(1.0, (ref.taps / 256.0) ** 0.5):
                    return "%s: float error %.3g at scale %.3g" % (tag, err, scale), what
            else:
                d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                if d.max() > 1:
                    return "%s: %d LSB" % (tag, d.max()), what
        for s_ in range(S):
            if states[s_].position() != refs[s_].position():
                return "step %d state %d: position %s, oracle %s" % (step, s_, states[s_].position(), refs[s_].position()), what
        if rng.rand() < 0.2:
            s_ = int(rng.randint(0, S))
            q2 = int(rng.randint(0, 11))
            if states[s_].set_quality(q2) != refs[s_].set_quality(q2):
                return "set_quality(%d) on state %d: codes differ" % (q2, s_), what
    for st in states:
        st.close()
    return None, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-frames", type=int, default=300000)
    ap.add_argument("--only", type=int, default=None, help="run exactly this trial seed")
    ap.add_argument("--many-channels", action="store_true", help="channel counts up to 100 (shorter calls)")
    ap.add_argument("--only-batch", action="store_true", help="with --only: the seed was a batch trial")
    ap.add_argument("--batch", action="store_true", help="every third trial: many streams through Batch.process_device")
    ap.add_argument("--many", action="store_true", help="every trial: many single-stream states through the many-states host call")
    args = ap.parse_args()
    orc.build()
    t0 = time.time()
    trials = fails = 0
    seed = (args.seed * 1000003) % (2 ** 32 - 10 ** 7)  # (trial seeds = seed + n must stay below numpy's 2^32)
    while time.time() - t0 < args.seconds:
        s = args.only if args.only is not None else seed + trials
        try:
            if args.many:
                err, what = many_trial(s, args.max_frames)
            elif (args.batch or args.only is not None) and (s % 3 == 0) and (args.batch or args.only_batch):
                err, what = batch_trial(s, args.max_frames)
            else:
                err, what = one_trial(s, args.max_frames if not args.many_channels else min(args.max_frames, 30000), args.many_channels)
        except RuntimeError as e:  # (an error code out of the library is a failure of the trial, with its seed)
            err, what = "exception: %s" % e, "seed %d" % s
            if args.only is not None:
                import traceback
                traceback.print_exc()
        trials += 1
        if err:
            fails += 1
            print("FAIL %s: %s" % (what, err), flush=True)
        if args.only is not None:
            print(what, "ok" if not err else "")
            break
    print("fuzz: %d trials, %d failures, %.0f s" % (trials, fails, time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
