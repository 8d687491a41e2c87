#!/usr/bin/env python3
"""Generate tests/golden/golden_channels.json from the REFERENCE ITSELF (dev container only):
scripts over the per-channel entry points and the failed-filter fallback.

  * speex_resampler_process_int / _process_float with input / output strides (reference
    deps/speex/resample.c:927-1036, 1170-1188): channels of one state advanced unevenly, then
    interleaved calls on the uneven state (resample.c:1061-1082);
  * filter changes that cannot build their filter (a ratio so large that the filter length
    overflows, resample.c:620-621): RESAMPLER_ERR_ALLOC_FAILED, resampler_basic_zero installed
    (resample.c:561-591, 785-791), zeros out with moving counters, then recovery.

Run on oracle/_ref/libspeexref.so (the reference's own resample.c, recipe: oracle/Makefile).  Stored
per op: return code, counters, sha1 of the WHOLE sentinel-filled output buffer (so samples a call
must not touch are pinned too), every channel's (last_sample, samp_frac_num, magic_samples), filter
length, rates and ratio.  DATA only.
Run:  python tests/golden/make_golden_channels.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

RATES = [8000, 16000, 22050, 32000, 44100, 48000, 96000]
BASE_TAPS = [8, 16, 32, 48, 64, 80, 96, 128, 160, 192, 256]  # quality -> base filter length (resample.c:226-238)


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def channel_input(kind, frames, seed):
    pcm = orc.lcg_pcm(frames, seed)
    return pcm if kind == "int" else (pcm.astype(np.float32) / np.float32(32768.0))


def apply_channel_op(eng, op, ch):
    """Run one op on `eng` (Reference, Oracle or the HIP mirror).  Returns the stored row."""
    kind = op[0]
    if kind in ("int_ch", "float_ch"):
        _, c, frames, cap, seed, istr, ostr = op
        k = kind[:-3]
        x = channel_input(k, frames, seed) if seed >= 0 else None
        rc, used, made, out = eng.channel_call(k, c, x, cap, istr, ostr, null_frames=frames)
        res = [int(rc), int(used), int(made), sha1(out)[:16]]
    elif kind in ("int", "float"):
        _, frames, cap, seed = op
        x = channel_input(kind, frames * ch, seed).reshape(frames, ch)
        rc, used, made, out = eng.raw_call(kind, x, cap)
        res = [int(rc), int(used), int(made), sha1(out)[:16]]
    elif kind == "rate":
        res = [int(eng.set_rate(op[1], op[2]))]
    elif kind == "ratefrac":
        res = [int(eng.set_rate_frac(op[1], op[2], op[3], op[4]))]
    elif kind == "quality":
        res = [int(eng.set_quality(op[1]))]
    elif kind == "skip":
        res = [int(eng.skip_zeros())]
    elif kind == "reset":
        res = [int(eng.reset_mem())]
    else:
        raise ValueError(kind)
    state = [list(map(int, p)) for p in eng.positions()]
    return res + [state, int(eng.taps)] + [int(v) for v in eng.rate()] + [int(v) for v in eng.ratio()]


def scripts(n_scripts=24, steps=26):
    rows = []
    for k in range(n_scripts):
        r = np.random.RandomState(7000 + k)
        ch = int(r.choice([1, 2, 2, 3, 4, 6]))
        i, o = int(r.choice(RATES)), int(r.choice(RATES))
        q = int(r.randint(0, 11))
        ref = orc.Reference(ch, i, o, q)
        ops, results = [], []
        failed = False
        for step in range(steps):
            pick = r.randint(0, 20)
            frames = int(r.choice([0, 1, 9, 160, 161, 700, 3000]))
            full = int(np.ceil(frames * ref.den / max(ref.num, 1))) + 2 if not failed else 4
            cap = [full, max(full // 2, 1), int(r.randint(0, full + 5)), 0][int(r.choice([0, 0, 0, 1, 2, 3]))]
            seed = int(r.randint(1, 1 << 30))
            if pick < 9:      # one channel, strided
                kind = "int_ch" if r.rand() < 0.6 else "float_ch"
                op = [kind, int(r.randint(0, ch)), frames, cap, seed if r.rand() > 0.06 else -1,
                      int(r.choice([1, 1, 2, 3, ch])), int(r.choice([1, 1, 2, ch, 5]))]
            elif pick < 13:   # interleaved, possibly on an uneven state
                op = ["int" if r.rand() < 0.6 else "float", frames, cap, seed]
            elif pick < 15:
                op = ["rate", int(r.choice(RATES)), int(r.choice(RATES))]
            elif pick == 15:  # a ratio whose filter length overflows 32 bits: the filter cannot be built
                q_now = ref.quality()
                num = (1 << 32) // BASE_TAPS[q_now] + int(r.randint(1000, 100000))
                op = ["ratefrac", int(num), 1, 48000, 16]
            elif pick == 16:
                op = ["quality", int(r.randint(0, 11))]
            elif pick == 17:
                op = ["skip"]
            elif pick == 18:
                op = ["reset"]
            else:
                n, d = int(r.randint(1, 9)), int(r.randint(1, 9))
                op = ["ratefrac", n, d, n * 1000, d * 1000]
            row = apply_channel_op(ref, op, ch)
            failed = failed or (op[0] in ("rate", "ratefrac", "quality") and row[0] == 1)
            if op[0] in ("rate", "ratefrac", "quality") and row[0] == 0:
                failed = False
            ops.append(op)
            results.append(row)
        rows.append(dict(name="chan_%02d" % k, channels=ch, in_rate=i, out_rate=o, quality=q, ops=ops,
                         results=results))
    return rows


def main():
    orc.build()
    assert orc.have_reference(), "oracle/_ref/libspeexref.so missing (need /root/reference)"
    rows = scripts()
    n_ops = sum(len(c["ops"]) for c in rows)
    n_fail = sum(1 for c in rows for op, res in zip(c["ops"], c["results"]) if op[0] in ("rate", "ratefrac", "quality") and res[0] == 1)
    n_zero = sum(1 for c in rows for op, res in zip(c["ops"], c["results"]) if op[0] in ("int", "float", "int_ch", "float_ch") and res[0] == 1)
    n_uneven = sum(1 for c in rows for op, res in zip(c["ops"], c["results"]) if op[0] in ("int", "float") and len(set(map(tuple, res[-6]))) > 1)
    doc = dict(generator="tests/golden/make_golden_channels.py",
               source="oracle/_ref/libspeexref.so (reference deps/speex/resample.c, -DFLOATING_POINT -DOUTSIDE_SPEEX)",
               sentinels=[orc.SENTINEL_I16, orc.SENTINEL_F32], scripts=rows)
    with open(os.path.join(HERE, "golden_channels.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("scripts %d, ops %d, failed filter changes %d, processing ops in the zero fallback %d, "
          "interleaved ops ending on an uneven state %d" % (len(rows), n_ops, n_fail, n_zero, n_uneven))


if __name__ == "__main__":
    main()
