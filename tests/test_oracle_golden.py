"""Pin the oracle (oracle/speex_oracle.c) to the reference: every golden vector produced by
the reference itself (native C + shipped WASM, tests/golden/make_golden.py) must be reproduced
BIT-FOR-BIT by the restatement -- lengths, counters, table bits and output bytes."""
import os

import numpy as np
import pytest

import oracle as orc
from golden_util import ROOT, drive, make_input, sha1


def test_lcg_generator_matches_scalar_definition():
    s, ref = 12345, []
    for _ in range(1000):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        v = s >> 16
        ref.append(v - 65536 if v >= 32768 else v)
    assert orc.lcg_pcm(1000, 12345).tolist() == ref


def test_golden_inputs_regenerate(golden):
    for c in golden["cases"]:
        if c["frames"] <= 100000:
            assert sha1(make_input(c)) == c["input_sha1"], c["name"]


def test_oracle_reproduces_every_golden_case(golden):
    for c in golden["cases"]:
        x = make_input(c)
        assert sha1(x) == c["input_sha1"], c["name"]
        o = orc.Oracle(c["channels"], c["in_rate"], c["out_rate"], c["quality"])
        assert (o.num, o.den, o.taps, o.oversample, o.kind, o.table_len) == (
            c["num"], c["den"], c["taps"], c["oversample"], c["kind"], c["table_len"]), c["name"]
        assert sha1(o.table()) == c["table_sha1"], c["name"] + ": filter table bits differ"
        out, calls = drive(o, c, x)
        assert out.shape[0] == c["out_frames"], c["name"]
        assert calls[: len(c["calls"])] == c["calls"], c["name"] + ": per-call counters differ"
        assert sha1(out) == c["out_sha1"], c["name"] + ": output bytes differ"
        if "out_full" in c:
            assert out.reshape(-1).tolist() == c["out_full"], c["name"]
        assert out[:8].reshape(-1).tolist() == c["head"] and out[-8:].reshape(-1).tolist() == c["tail"]


def test_oracle_bookkeeping_matches_planner_goldens(golden):
    for p in golden["planner"]:
        o = orc.Oracle(1, p["in_rate"], p["out_rate"], p["quality"])
        for (f, cap, used, n_out, pos, ph) in p["calls"]:
            out, u = o.process(np.zeros((f, 1), np.int16), cap)
            assert (u, out.shape[0]) + o.position() == (used, n_out, pos, ph), (p["in_rate"], p["out_rate"])


def test_oracle_error_paths():
    for args in [(0, 44100, 48000, 7), (2, 0, 48000, 7), (2, 44100, 0, 7), (2, 44100, 48000, 11),
                 (2, 44100, 48000, -1)]:
        with pytest.raises(ValueError, match="Invalid argument."):
            orc.Oracle(*args)
    orc.Oracle(1, 8000, 8000, 0)  # quality 0 is accepted (reference resample.c:804)


@pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref not built")
def test_oracle_equals_reference_build_on_random_configs():
    rng = np.random.RandomState(1)
    rates = [8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 96000]
    for trial in range(40):
        ch = int(rng.randint(1, 5))
        i, o = int(rng.choice(rates)), int(rng.choice(rates))
        q = int(rng.randint(0, 11))
        a, r = orc.Oracle(ch, i, o, q), orc.Reference(ch, i, o, q)
        assert np.array_equal(a.table().view(np.uint32), r.table().view(np.uint32))
        for call in range(4):
            f = int(rng.choice([0, 1, 100, 777, 3000]))
            cap = int(rng.choice([0, 10, 1000, 100000]))
            x = orc.lcg_pcm(f * ch, trial * 10 + call).reshape(f, ch)
            oa, ua = a.process(x, cap)
            orr, ur = r.process(x, cap)
            assert ua == ur and np.array_equal(oa, orr) and a.position() == r.position()
            for c in range(ch):
                assert np.array_equal(a.history(c), r.history(c))


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources"), reason="reference fixtures absent")
def test_oracle_on_the_reference_fixture_files(golden):
    """The reference's own resources/*.pcm (not copied into this repo): digests from SURVEY section 4."""
    for r in golden["resources"]:
        raw = np.fromfile(os.path.join("/root/reference/resources", r["file"]), dtype=np.uint8)
        raw = raw[: raw.size - raw.size % (2 * r["channels"])]
        x = raw.view(np.int16).reshape(-1, r["channels"])
        spec = dict(channels=r["channels"], in_rate=r["in_rate"], out_rate=r["out_rate"], chunks="whole")
        out, _ = drive(orc.Oracle(r["channels"], r["in_rate"], r["out_rate"], r["quality"]), spec, x)
        assert out.shape[0] == r["out_frames"] and sha1(out) == r["sha1"], r["file"]
        # the reference test's only assertion (src/test.ts:40): durations agree within 10 ms
        assert abs(x.shape[0] / r["in_rate"] - out.shape[0] / r["out_rate"]) < 0.01


def test_oracle_float_entry_point_matches_the_reference_goldens(golden):
    """speex_resampler_process_interleaved_float (native reference; the WASM build does not export
    it): bit-identical float32 bytes, counters and positions, incl. capacity-bound calls and the
    8k->48k case where one 160-frame block emits more than 1024 outputs (float path only)."""
    from make_golden import float_input
    for c in golden["float_cases"]:
        x = float_input(c["frames"], c["channels"], c["seed"])
        o = orc.Oracle(c["channels"], c["in_rate"], c["out_rate"], c["quality"])
        assert o.kind == c["kind"]
        outs, off = [], 0
        for (n, cap, used, made, pos, ph) in c["calls"]:
            y, u = o.process_float(x[off: off + n], cap)
            assert (u, y.shape[0]) + o.position() == (used, made, pos, ph), c["name"]
            outs.append(y)
            off += u
        out = np.concatenate(outs)
        assert out.shape[0] == c["out_frames"] and sha1(out) == c["out_sha1"], c["name"]


def test_oracle_mid_stream_control_matches_the_reference_goldens(golden):
    """set_rate / set_rate_frac / set_quality / skip_zeros / reset_mem between processing calls
    (reference deps/speex/resample.c:703-782, 904-922, 1084-1220): 40 scripts x 24 ops recorded
    on the native reference; the restatement must reproduce every return code, counter,
    output digest and visible state, incl. the pending ("magic") frames."""
    from make_golden import apply_op
    n_pending = 0
    for c in golden["control_cases"]:
        o = orc.Oracle(c["channels"], c["in_rate"], c["out_rate"], c["quality"])
        for k, (op, want) in enumerate(zip(c["ops"], c["results"])):
            row, _ = apply_op(o, op, c["channels"])
            assert row == want, (c["name"], k, op)
            n_pending += row[-8] > 0
    assert n_pending > 100  # the scripts really exercise the pending-frame paths


@pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref not built")
def test_oracle_equals_reference_build_on_random_control_scripts():
    """Same ops on the reference build and on the restatement, beyond the committed scripts
    (full output bytes and histories compared, not digests)."""
    from make_golden import apply_op
    for seed in range(25):
        r = np.random.RandomState(seed)
        ch = int(r.choice([1, 2, 5]))
        args = (ch, int(r.choice([8000, 11025, 44100, 48000])), int(r.choice([8000, 16000, 48000])),
                int(r.randint(0, 11)))
        a, b = orc.Reference(*args), orc.Oracle(*args)
        for _ in range(20):
            pick = r.randint(0, 6)
            if pick < 3:
                op = ["int" if r.rand() < 0.5 else "float", int(r.randint(0, 1500)), int(r.randint(0, 3000)),
                      int(r.randint(1, 1 << 30))]
            elif pick == 3:
                op = ["rate", int(r.choice([8000, 11025, 44100, 48000])), int(r.choice([8000, 16000, 48000]))]
            elif pick == 4:
                op = ["quality", int(r.randint(0, 11))]
            else:
                op = [["skip"], ["reset"]][int(r.randint(0, 2))]
            ra, ya = apply_op(a, op, ch)
            rb, yb = apply_op(b, op, ch)
            assert ra == rb, (seed, op)
            if ya is not None:
                assert ya.tobytes() == yb.tobytes()
            for c in range(ch):
                assert a.history(c).tobytes() == b.history(c).tobytes()
                assert a.pending(c).tobytes() == b.pending(c).tobytes()


def test_restatement_reproduces_the_per_channel_and_failed_filter_scripts():
    """the per-channel entry points with strides (resample.c:927-1036,
    1170-1188), interleaved calls on states whose channels stand apart (:1061-1082) and the
    resampler_basic_zero fallback after a filter change that cannot build its filter (:561-591,
    785-791) -- 624 ops recorded from the reference (tests/golden/make_golden_channels.py): return
    codes, counters, digests of the whole sentinel-filled output buffers, every channel's position."""
    import json
    from make_golden_channels import apply_channel_op
    with open(os.path.join(ROOT, "tests", "golden", "golden_channels.json")) as f:
        doc = json.load(f)
    assert doc["sentinels"] == [orc.SENTINEL_I16, orc.SENTINEL_F32]
    ops = zero_ops = 0
    for c in doc["scripts"]:
        eng = orc.Oracle(c["channels"], c["in_rate"], c["out_rate"], c["quality"])
        for k, (op, want) in enumerate(zip(c["ops"], c["results"])):
            got = apply_channel_op(eng, op, c["channels"])
            assert got == want, (c["name"], k, op, got, want)
            ops += 1
            zero_ops += op[0] in ("int", "float", "int_ch", "float_ch") and want[0] == 1
    assert ops == 624 and zero_ops >= 50
    if orc.have_reference():  # dev container / prebuilt _ref: the recorded rows ARE the reference's
        c = doc["scripts"][3]
        eng = orc.Reference(c["channels"], c["in_rate"], c["out_rate"], c["quality"])
        for op, want in zip(c["ops"], c["results"]):
            assert apply_channel_op(eng, op, c["channels"]) == want
