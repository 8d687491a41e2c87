#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REFERENCE ITSELF (dev container only).

Sources of truth, both run here and required to agree bit-for-bit:
  * oracle/_ref/libspeexref.so -- /root/reference/deps/speex/resample.c compiled with the
    -D flags of scripts/build_emscripten.sh (recipe: oracle/Makefile)
  * the shipped WASM -- require('/root/reference/app/index.js') under node (tests/golden/wasm_sha1.js)

What is stored is DATA only: seeds/parameters, expected lengths, counters, sha1 digests, short
excerpts and a few complete small vectors.  No reference source, no music file is copied.
Run:  python tests/golden/make_golden.py      (rewrites golden.json next to this script)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

REF_ROOT = "/root/reference"


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_input(spec):
    ch, frames = spec["channels"], spec["frames"]
    if spec["input"] == "lcg":
        return orc.lcg_pcm(frames * ch, spec["seed"]).reshape(frames, ch)
    if spec["input"] == "tone":
        return orc.tone_pcm(frames, ch, spec["seed"])
    if spec["input"] == "edge":  # saturation / silence / impulses
        x = np.zeros((frames, ch), np.int16)
        x[: frames // 4] = 32767
        x[frames // 4: frames // 2] = -32768
        x[frames // 2 + 3] = 32767
        x[frames // 2 + 200] = -32768
        x[3 * frames // 4:] = orc.lcg_pcm((frames - 3 * frames // 4) * ch, spec["seed"]).reshape(-1, ch)
        return x
    raise ValueError(spec["input"])


def chunk_plan(spec, total_bytes):
    """Byte sizes of successive processChunk calls."""
    mode = spec.get("chunks", "whole")
    fb = 2 * spec["channels"]
    if mode == "whole":
        return [total_bytes]
    if isinstance(mode, int):
        sizes = [mode] * (total_bytes // mode)
        if total_bytes % mode:
            sizes.append(total_bytes % mode)
        return sizes
    if mode == "ragged":
        rng = np.random.RandomState(spec["seed"] + 99)
        sizes, left = [], total_bytes
        while left:
            n = min(left, int(rng.choice([0, 1, 2, 7, 160, 161, 500, 1024, 4096, 10000])) * fb)
            sizes.append(n)
            left -= n
        return sizes
    raise ValueError(mode)


def run_reference(spec, x):
    """Drive the native reference exactly like src/index.ts does (capacity rule, dropped frames)."""
    ch = spec["channels"]
    r = orc.Reference(ch, spec["in_rate"], spec["out_rate"], spec["quality"])
    outs, per_call = [], []
    out_buf_size, off = -1, 0
    for nbytes in chunk_plan(spec, x.size * 2):
        f = nbytes // (2 * ch)
        cap, out_buf_size = orc.wrapper_capacity(nbytes, spec["in_rate"], spec["out_rate"], ch,
                                                 out_buf_size)
        o, used = r.process(x[off: off + f], cap)
        pos, ph = r.position()
        per_call.append([f, cap, used, int(o.shape[0]), int(pos), int(ph)])
        outs.append(o)
        off += f
    out = np.concatenate(outs) if outs else np.zeros((0, ch), np.int16)
    return r, out, per_call


def wasm_sha1(spec, x):
    """sha1 of the shipped WASM's output for the same call sequence (None if node/ref missing)."""
    js = os.path.join(HERE, "wasm_sha1.js")
    if not os.path.exists(os.path.join(REF_ROOT, "app", "index.js")):
        return None
    with tempfile.NamedTemporaryFile(suffix=".pcm", delete=False) as f:
        f.write(x.tobytes())
        path = f.name
    try:
        arg = json.dumps({"file": path, "channels": spec["channels"], "in_rate": spec["in_rate"],
                          "out_rate": spec["out_rate"], "quality": spec["quality"],
                          "chunks": chunk_plan(spec, x.size * 2)})
        res = subprocess.run(["node", js, arg], capture_output=True, text=True, timeout=600)
        if res.returncode != 0:
            raise RuntimeError(res.stderr)
        return json.loads(res.stdout)
    finally:
        os.unlink(path)


CASES = []


def case(name, ch, i, o, q, frames, input="lcg", seed=12345, chunks="whole", full=False, wasm=True):
    CASES.append(dict(name=name, channels=ch, in_rate=i, out_rate=o, quality=q, frames=frames,
                      input=input, seed=seed, chunks=chunks, full=full, wasm=wasm))


# A. SURVEY section 4 synthetic 2^20-frame rows (BASELINE configs 2,3,4 + direct_single)
case("cfg2_44k1_48k_2ch_q7_1M", 2, 44100, 48000, 7, 1 << 20)
case("cfg3_24k_48k_1ch_q10_1M", 1, 24000, 48000, 10, 1 << 20)
case("cfg4_48k_44k1_8ch_q5_1M", 8, 48000, 44100, 5, 1 << 20)
case("f3_24k_48k_1ch_q5_1M", 1, 24000, 48000, 5, 1 << 20)
# B. the reference test's seven (rates, ch, q) tuples (src/test.ts:14-22) on a deterministic
#    music-like signal: whole buffer and 64 KiB stream chunks (createReadStream default)
REF_TUPLES = [(1, 24000, 48000, 5), (2, 24000, 24000, 5), (2, 24000, 48000, 10), (2, 44100, 48000, 7),
              (2, 44100, 48000, 10), (2, 44100, 48000, 1), (2, 44100, 24000, 5)]
for (ch, i, o, q) in REF_TUPLES:
    case("t_%d_%d_%dch_q%d_whole" % (i, o, ch, q), ch, i, o, q, 100000, input="tone", seed=3)
    case("t_%d_%d_%dch_q%d_64k" % (i, o, ch, q), ch, i, o, q, 100000, input="tone", seed=3, chunks=65536)
# C. complete small vectors, one per kernel kind + edge amplitudes
case("full_interp_single", 2, 44100, 48000, 7, 2048, full=True)
case("full_direct_single", 1, 24000, 48000, 5, 2048, full=True)
case("full_direct_double", 1, 24000, 48000, 10, 2048, full=True)
case("full_interp_double", 2, 44100, 48000, 10, 2048, full=True)
case("full_down_interp", 3, 48000, 44100, 5, 2048, full=True)
case("full_edge_sat", 2, 44100, 48000, 7, 4096, input="edge", full=True)
case("full_q0", 1, 8000, 11025, 0, 1024, full=True)
# D. capacity rule / dropped frames and ragged call sequences
case("f5_640B_chunks", 2, 44100, 48000, 7, 16000, chunks=640)
case("f5_ragged_up", 2, 44100, 48000, 7, 60000, chunks="ragged", seed=5)
case("f5_ragged_down6", 1, 48000, 8000, 6, 60000, chunks="ragged", seed=6)
case("f5_ragged_down_8ch", 8, 48000, 44100, 5, 30000, chunks="ragged", seed=7)
case("f5_ragged_direct_dn", 2, 48000, 24000, 8, 30000, chunks="ragged", seed=8)
case("f5_ragged_q10", 2, 44100, 48000, 10, 30000, chunks="ragged", seed=9)
case("odd_rates_big_den", 1, 44101, 48000, 4, 50000, chunks=8192)
case("big_ratio_down", 2, 192000, 8000, 9, 100000, chunks=20000)
case("big_ratio_up", 1, 8000, 96000, 8, 20000, chunks=5000)


def planner_cases():
    """Random (F, capacity) call sequences: pins the integer bookkeeping
    (in_used, out_len, last_sample, samp_frac_num) without any audio arithmetic."""
    out = []
    rng = np.random.RandomState(2024)
    for (i, o, q) in [(44100, 48000, 7), (48000, 44100, 5), (48000, 8000, 3), (8000, 48000, 4),
                      (24000, 48000, 5), (48000, 24000, 5), (44100, 32000, 2), (32000, 32000, 1),
                      (96000, 44100, 6), (11025, 192000, 0)]:
        r = orc.Reference(1, i, o, q)
        calls = []
        for _ in range(120):
            f = int(rng.choice([0, 1, 2, 3, 50, 159, 160, 161, 333, 1000, 4097]))
            cap = int(rng.choice([0, 1, 2, 54, 500, 1023, 1024, 1025, 3000, 100000]))
            x = np.zeros((f, 1), np.int16)
            oo, used = r.process(x, cap)
            pos, ph = r.position()
            calls.append([f, cap, used, int(oo.shape[0]), int(pos), int(ph)])
        out.append(dict(in_rate=i, out_rate=o, quality=q, calls=calls))
    return out


FLOAT_CASES = [  # (name, channels, in_rate, out_rate, quality, frames, call plan [(frames, capacity)...])
    ("f_interp_single", 2, 44100, 48000, 7, 6000), ("f_interp_double", 2, 44100, 48000, 10, 6000),
    ("f_direct_single", 1, 24000, 48000, 5, 6000), ("f_direct_double", 1, 24000, 48000, 10, 6000),
    ("f_down_8ch", 8, 48000, 44100, 5, 4000), ("f_decim3", 2, 48000, 16000, 6, 6000),
    ("f_up6_blockcap", 1, 8000, 48000, 3, 6000),  # > 1024 outputs per 160-frame block: float path only
]


def float_input(frames, ch, seed):
    """deterministic float32 in [-1, 1): the LCG samples scaled by 2^-15"""
    return (orc.lcg_pcm(frames * ch, seed).astype(np.float32) / np.float32(32768.0)).reshape(frames, ch)


def float_cases():
    """speex_resampler_process_interleaved_float on the native reference (the WASM build does not
    export it): mixed call sizes incl. capacity-bound ones; digests of the float32 bytes."""
    rows = []
    for k, (name, ch, i, o, q, frames) in enumerate(FLOAT_CASES):
        x = float_input(frames, ch, 4000 + k)
        r = orc.Reference(ch, i, o, q)
        outs, calls, off = [], [], 0
        for n, cap in [(1, 1 << 20), (999, 1 << 20), (2000, 1500), (frames, 1 << 20)]:
            part = x[off: off + n]
            y, used = r.process_float(part, cap)
            pos, ph = r.position()
            calls.append([int(part.shape[0]), cap, used, int(y.shape[0]), int(pos), int(ph)])
            outs.append(y)
            off += used
        out = np.concatenate(outs)
        rows.append(dict(name=name, channels=ch, in_rate=i, out_rate=o, quality=q, frames=frames, seed=4000 + k,
                         kind=r.kind, calls=calls, out_frames=int(out.shape[0]), out_sha1=sha1(out),
                         head=[float(v) for v in out[:4].reshape(-1)]))
        print("%-34s %-20s out=%8d %s (float)" % (name, r.kind, out.shape[0], rows[-1]["out_sha1"][:12]))
    return rows


CONTROL_RATES = [8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000]


def control_input(op, ch):
    """Input block of a control-script processing op: LCG int16, or the same scaled to [-1, 1)."""
    kind, frames, _cap, seed = op[:4]
    if kind in ("int_null", "float_null"):
        return None
    pcm = orc.lcg_pcm(frames * ch, seed).reshape(frames, ch)
    if kind == "int":
        return pcm
    return (pcm.astype(np.float32) / np.float32(32768.0)).reshape(frames, ch)


def apply_op(eng, op, ch):
    """Run one op of a control script on `eng` (Reference, Oracle or the HIP mirror: same method
    names).  Returns (row, output-or-None); row = what the golden file stores for the op."""
    kind = op[0]
    out = None
    if kind in ("int", "float", "int_null", "float_null"):
        x = control_input(op, ch)
        fn = eng.process if kind.startswith("int") else eng.process_float
        if x is None:
            out, used = fn(None, op[2], null_frames=op[1])
        else:
            out, used = fn(x, op[2])
        res = [int(used), int(out.shape[0]), sha1(out)[:16]]
    elif kind == "rate":
        res = [eng.set_rate(op[1], op[2])]
    elif kind == "ratefrac":
        res = [eng.set_rate_frac(op[1], op[2], op[3], op[4])]
    elif kind == "quality":
        res = [eng.set_quality(op[1])]
    elif kind == "skip":
        res = [eng.skip_zeros()]
    elif kind == "reset":
        res = [eng.reset_mem()]
    else:
        raise ValueError(kind)
    pos, ph = eng.position()
    state = [int(pos), int(ph), int(len(eng.pending())), int(eng.taps), int(eng.input_latency()),
             int(eng.output_latency())] + [int(v) for v in eng.rate()] + [int(v) for v in eng.ratio()]
    return res + state, out


def control_cases(n_scripts=40):
    """Mid-stream control (reference deps/speex/resample.c:703-782, 904-922, 1084-1220): random
    scripts of process / set_rate / set_rate_frac / set_quality / skip_zeros / reset_mem ops run
    on the native reference.  Stored: the ops and, per op, return code or (used, produced,
    digest) plus the visible state (position, phase, pending "magic" frames, filter length,
    latencies, rates, ratio)."""
    rows = []
    for k in range(n_scripts):
        r = np.random.RandomState(9000 + k)
        ch = int(r.choice([1, 2, 2, 3, 8]))
        i, o = int(r.choice(CONTROL_RATES)), int(r.choice(CONTROL_RATES))
        q = int(r.randint(0, 11))
        ref = orc.Reference(ch, i, o, q)
        ops, results = [], []
        for step in range(24):
            pick = r.randint(0, 12)
            if pick < 6 or step == 0:
                frames = int(r.choice([0, 1, 7, 100, 160, 161, 500, 2000, 5000]))
                full = int(np.ceil(frames * ref.den / ref.num)) + 2
                cap = [full, full // 2, int(r.randint(0, full + 5)), 0][int(r.choice([0, 0, 1, 2, 2, 3]))]
                kind = "int" if r.rand() < 0.6 else "float"
                if r.rand() < 0.08:
                    kind += "_null"
                op = [kind, frames, int(cap), int(r.randint(1, 1 << 30))]
            elif pick < 8:
                op = ["rate", int(r.choice(CONTROL_RATES)), int(r.choice(CONTROL_RATES))]
            elif pick == 8:
                n, d = (160, 147) if r.rand() < 0.3 else (int(r.randint(1, 13)), int(r.randint(1, 13)))
                op = ["ratefrac", n, d, n * 1000, d * 1000]
            elif pick == 9:
                op = ["quality", int(r.randint(0, 11))]
            elif pick == 10:
                op = ["skip"]
            else:
                op = ["reset"]
            row, _ = apply_op(ref, op, ch)
            ops.append(op)
            results.append(row)
        rows.append(dict(name="ctl_%02d" % k, channels=ch, in_rate=i, out_rate=o, quality=q, ops=ops,
                         results=results))
    n_pending = sum(1 for c in rows for res in c["results"] if res[-8] > 0)
    print("control scripts: %d, ops with pending frames afterwards: %d" % (len(rows), n_pending))
    return rows


def resource_cases():
    """The reference's own fixtures (resources/*.pcm; read whole, header and all, like
    src/test.ts:29).  Only digests are stored; the test needs /root/reference to re-run them."""
    rows = []
    files = {24000: {1: "24000hz_mono_test.pcm", 2: "24000hz_test.pcm"}, 44100: {2: "44100hz_test.pcm"}}
    for (ch, i, o, q) in REF_TUPLES:
        path = os.path.join(REF_ROOT, "resources", files[i][ch])
        if not os.path.exists(path):
            continue
        raw = np.fromfile(path, dtype=np.uint8)
        raw = raw[: raw.size - raw.size % (2 * ch)]
        x = raw.view(np.int16).reshape(-1, ch)
        spec = dict(channels=ch, in_rate=i, out_rate=o, quality=q, chunks="whole")
        _, out, per_call = run_reference(spec, x)
        w = wasm_sha1(spec, x)
        assert w is None or w["sha1"] == sha1(out), "native reference != WASM on %s" % path
        rows.append(dict(file=files[i][ch], channels=ch, in_rate=i, out_rate=o, quality=q,
                         out_frames=int(out.shape[0]), sha1=sha1(out), wasm_agrees=w is not None))
    return rows


def main():
    orc.build()
    assert orc.have_reference(), "oracle/_ref/libspeexref.so missing (need /root/reference)"
    cases = []
    for spec in CASES:
        x = make_input(spec)
        r, out, per_call = run_reference(spec, x)
        row = dict(spec)
        row.update(num=r.num, den=r.den, taps=r.taps, oversample=r.oversample, kind=r.kind,
                   table_len=r.table_len, table_sha1=sha1(r.table()), input_sha1=sha1(x),
                   out_frames=int(out.shape[0]), out_sha1=sha1(out),
                   head=out[:8].reshape(-1).tolist(), tail=out[-8:].reshape(-1).tolist(),
                   calls=per_call if len(per_call) <= 400 else per_call[:400],
                   n_calls=len(per_call), total_in_used=int(sum(c[2] for c in per_call)),
                   final_pos=per_call[-1][4:6] if per_call else [0, 0],
                   out_min=int(out.min()) if out.size else 0, out_max=int(out.max()) if out.size else 0)
        if spec["full"]:
            row["out_full"] = out.reshape(-1).tolist()
        if spec["wasm"]:
            w = wasm_sha1(spec, x)
            if w is not None:
                assert w["sha1"] == row["out_sha1"], "native reference != WASM on " + spec["name"]
                assert w["out_frames"] == row["out_frames"]
            row["wasm_agrees"] = w is not None
        cases.append(row)
        print("%-34s %-20s out=%8d used=%8d %s wasm=%s" % (spec["name"], r.kind, row["out_frames"],
                                                          row["total_in_used"], row["out_sha1"][:12],
                                                          row.get("wasm_agrees")))
    doc = dict(
        generator="tests/golden/make_golden.py",
        sources=["oracle/_ref/libspeexref.so (reference deps/speex/resample.c, -DFLOATING_POINT "
                 "-DOUTSIDE_SPEEX)", "reference app/speex_wasm.js via node"],
        lcg="s=s*1664525+1013904223 mod 2^32; sample=int16(s>>16); seed per case",
        cases=cases, planner=planner_cases(), resources=resource_cases(), float_cases=float_cases(),
        control_cases=control_cases())
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote golden.json: %d cases, %d planner sets, %d resource rows" %
          (len(cases), len(doc["planner"]), len(doc["resources"])))


if __name__ == "__main__":
    main()
