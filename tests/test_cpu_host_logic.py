"""CPU tests of the PRODUCT's host side (no GPU, no compute calls): the C-ABI library loads and
exports every symbol include/*.h declares; its filter design reproduces the reference's table
bits; its closed-form stream planner reproduces the reference's per-call counters."""
import hashlib
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import speexhip
from golden_util import ROOT


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_library_exports_every_declared_symbol():
    lib = speexhip.lib()
    header = open(os.path.join(ROOT, "include", "speexhip_resampler.h")).read()
    declared = set(re.findall(r"\b(speexhip_\w+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(speexhip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    out = subprocess.run(["nm", "-D", "--defined-only", speexhip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (\w+)", out))
    assert declared <= exported
    assert b"gfx950" in lib.speexhip_version()


def test_strerror_matches_reference_strings():
    # reference resample.c:1222-1239 (5 = OVERFLOW has no case there -> "Unknown error...")
    assert speexhip.strerror(0) == "Success."
    assert speexhip.strerror(1) == "Memory allocation failed."
    assert speexhip.strerror(2) == "Bad resampler state."
    assert speexhip.strerror(3) == "Invalid argument."
    assert speexhip.strerror(4) == "Input and output buffers overlap."
    assert speexhip.strerror(5) == "Unknown error. Bad error code or strange version mismatch."
    assert speexhip.strerror(99) == "Unknown error. Bad error code or strange version mismatch."


def test_filter_design_bits_match_reference_goldens(golden):
    seen = set()
    for c in golden["cases"]:
        key = (c["in_rate"], c["out_rate"], c["quality"])
        if key in seen:
            continue
        seen.add(key)
        info, table = speexhip.design_filter(*key)
        assert (info["num_rate"], info["den_rate"], info["filt_len"], info["oversample"],
                speexhip.KERNEL_NAMES[info["kernel"]], info["sinc_table_length"]) == (
            c["num"], c["den"], c["taps"], c["oversample"], c["kind"], c["table_len"]), c["name"]
        assert sha1(table) == c["table_sha1"], c["name"] + ": product table differs from the reference"
    assert len(seen) >= 12


def test_filter_design_equals_oracle_on_a_rate_grid():
    import oracle as orc
    rates = [8000, 11025, 12000, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000]
    for i in rates:
        for o in rates:
            for q in (0, 3, 5, 8, 9, 10):
                info, table = speexhip.design_filter(i, o, q)
                ref = orc.Oracle(1, i, o, q)
                assert (info["filt_len"], info["oversample"], speexhip.KERNEL_NAMES[info["kernel"]]) == (
                    ref.taps, ref.oversample, ref.kind)
                assert np.array_equal(table.view(np.uint32), ref.table().view(np.uint32)), (i, o, q)


def test_design_filter_rejects_bad_arguments():
    for args in [(0, 48000, 7), (44100, 0, 7), (44100, 48000, 11), (44100, 48000, -1)]:
        with pytest.raises(ValueError, match="Invalid argument."):
            speexhip.design_filter(*args)


def test_planner_matches_reference_counters(golden):
    for p in golden["planner"]:
        info, _ = speexhip.design_filter(p["in_rate"], p["out_rate"], p["quality"], want_table=False)
        last, frac = 0, 0
        for (f, cap, used, n_out, pos, ph) in p["calls"]:
            c, n, last, frac = speexhip.plan_call(info["num_rate"], info["den_rate"], f, cap, last, frac)
            assert (c, n, last, frac) == (used, n_out, pos, ph), (p["in_rate"], p["out_rate"], f, cap)


def test_planner_matches_every_golden_call_sequence(golden):
    for c in golden["cases"]:
        last, frac = 0, 0
        for (f, cap, used, n_out, pos, ph) in c["calls"]:
            got = speexhip.plan_call(c["num"], c["den"], f, cap, last, frac)
            assert got == (used, n_out, pos, ph), c["name"]
            last, frac = pos, ph


def test_planner_equals_oracle_on_random_calls():
    import oracle as orc
    rng = np.random.RandomState(7)
    for (i, o) in [(44100, 48000), (48000, 44100), (192000, 8000), (8000, 192000), (44101, 48000),
                   (48000, 16000), (22050, 22050), (11025, 48000)]:
        ref = orc.Oracle(1, i, o, 2)
        last, frac = 0, 0
        for _ in range(300):
            f = int(rng.randint(0, 3000))
            cap = int(rng.choice([0, 1, 7, 100, 1024, 5000, 1 << 20]))
            out, used = ref.process(np.zeros((f, 1), np.int16), cap)
            c, n, last, frac = speexhip.plan_call(ref.num, ref.den, f, cap, last, frac)
            assert (c, n, last, frac) == (used, out.shape[0]) + ref.position()


def test_init_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="HIP device error"):
        speexhip.Resampler(2, 44100, 48000, 7)
    # argument errors still win over the device check (reference order, resample.c:804-809)
    with pytest.raises(ValueError, match="Invalid argument."):
        speexhip.Resampler(2, 44100, 48000, 11)


def test_python_mirror_keeps_the_wrapper_contract_without_touching_the_gpu():
    r = speexhip.SpeexResampler(2, 44100, 48000)
    assert (r.channels, r.inRate, r.outRate, r.quality) == (2, 44100, 48000, 7)
    with pytest.raises(ValueError, match="Chunk length should be a multiple of channels \\* 2 bytes"):
        r.processChunk(b"\0" * 7)
    with pytest.raises(ValueError, match="Chunk length should be a multiple"):
        speexhip.SpeexResampler(0, 44100, 48000).processChunk(b"\0" * 8)


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_addon_loads_and_mirrors_reference_errors():
    addon = os.path.join(ROOT, "node-speex-resampler_amd", "speex_hip_napi.node")
    if not os.path.exists(addon):
        pytest.skip("addon not built")
    js = r"""
const R = require(process.argv[1]);
const out = {};
const grab = (k, f) => { try { f(); out[k] = 'no throw'; } catch (e) { out[k] = e.message; } };
grab('early', () => new R.default(2, 44100, 48000).processChunk(Buffer.alloc(8)));
R.default.initPromise.then(() => {
  grab('len', () => new R.default(2, 44100, 48000).processChunk(Buffer.alloc(7)));
  grab('ch0', () => new R.default(0, 44100, 48000).processChunk(Buffer.alloc(8)));
  grab('q11', () => new R.default(2, 44100, 48000, 11).processChunk(Buffer.alloc(8)));
  grab('rate0', () => new R.default(2, 0, 48000).processChunk(Buffer.alloc(8)));
  const r = new R.default(2, 44100, 48000);
  out.fields = [r.channels, r.inRate, r.outRate, r.quality];
  out.exports = [typeof R.default, typeof R.SpeexResamplerTransform, typeof R.default.initPromise.then];
  console.log(JSON.stringify(out));
});
"""
    res = subprocess.run(["node", "-e", js, os.path.join(ROOT, "node-speex-resampler_amd", "index.js")],
                         capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    import json
    out = json.loads(res.stdout)
    assert out["early"] == "You need to wait for SpeexResampler.initPromise before calling this method"
    assert out["len"] == "Chunk length should be a multiple of channels * 2 bytes"
    assert out["ch0"] == "Chunk length should be a multiple of channels * 2 bytes"
    assert out["q11"] == "Invalid argument." and out["rate0"] == "Invalid argument."
    assert out["fields"] == [2, 44100, 48000, 7]
    assert out["exports"] == ["function", "function", "function"]


class _BookkeepingModel:
    """The integer side of a stream, driven ONLY through the library's host-only entry points
    (design_filter_frac, plan_call_ex, plan_filter_change) -- i.e. the product's planner and
    re-alignment rules without a GPU.  Mirrors what engine.cpp keeps per stream."""

    def __init__(self, in_rate, out_rate, quality):
        self.quality = quality
        self.last = self.frac = self.magic = 0
        self.started = False
        self.line = 0
        self._adopt(speexhip.design_filter_frac(in_rate, out_rate, in_rate, out_rate, quality))

    def _adopt(self, info):
        self.info = info
        self.num, self.den, self.taps = info["num_rate"], info["den_rate"], info["filt_len"]
        self.line = max(self.line, self.taps - 1 + 160)

    def _change(self, info, rescale):
        frac = self.frac
        if rescale:
            rc, _, _, _, frac = speexhip.plan_filter_change(self.taps, info["filt_len"], self.magic, self.frac,
                                                            self.den, info["den_rate"])
            if rc:
                return rc
        if self.started:
            rc, _shift, magic, delta, _ = speexhip.plan_filter_change(self.taps, info["filt_len"], self.magic)
            assert rc == 0
            self.magic, self.last = magic, self.last + delta
        self.frac = frac
        self._adopt(info)
        return 0

    def op(self, op):
        kind = op[0]
        if kind in ("int", "float", "int_null", "float_null"):
            used, made, self.last, self.frac, self.magic = speexhip.plan_call_ex(
                self.num, self.den, op[1], op[2], kind.startswith("float"), self.line - (self.taps - 1),
                self.last, self.frac, self.magic)
            self.started = self.started or (op[1] > 0 and op[2] > 0)
            res = [used, made]
        elif kind in ("rate", "ratefrac"):
            n, d, i, o = (op[1], op[2], op[1], op[2]) if kind == "rate" else op[1:5]
            if (self.info["in_rate"], self.info["out_rate"], self.num, self.den) == (i, o, n, d):
                res = [0]
            else:
                res = [self._change(speexhip.design_filter_frac(n, d, i, o, self.quality), True)]
        elif kind == "quality":
            if op[1] != self.quality:
                self.quality = op[1]
                self._change(speexhip.design_filter_frac(self.num, self.den, self.info["in_rate"],
                                                         self.info["out_rate"], op[1]), False)
            res = [0]
        elif kind == "skip":
            self.last = self.taps // 2
            res = [0]
        else:
            self.last = self.frac = self.magic = 0
            res = [0]
        return res + [self.last, self.frac, self.magic, self.taps, self.taps // 2,
                      ((self.taps // 2) * self.den + (self.num >> 1)) // self.num,
                      self.info["in_rate"], self.info["out_rate"], self.num, self.den]


def test_planner_and_realignment_match_the_reference_control_scripts(golden):
    """SURVEY 8(f) row N3 on the CPU: the product's host logic (planner with pending frames and
    either entry point's block rules, filter-change re-alignment, phase rescaling) reproduces the
    reference's counters and state over the 40 recorded control scripts (960 ops)."""
    checked = 0
    for c in golden["control_cases"]:
        m = _BookkeepingModel(c["in_rate"], c["out_rate"], c["quality"])
        for k, (op, want) in enumerate(zip(c["ops"], c["results"])):
            got = m.op(op)
            want = [w for w in want if not isinstance(w, str)]  # digests need the GPU
            assert got == want, (c["name"], k, op, got, want)
            checked += 1
    assert checked == 960


def test_bench_launcher_fails_loudly_without_the_gpus_it_was_asked_for():
    """`python bench.py --gpus 2` with no rendezvous in the environment starts two ranks itself
    (before anything touches a GPU).  On a box without two GPUs every rank must exit non-zero with
    a message, the launcher must report it, and NO JSON line may appear -- never a silent 1-GPU
    number labelled as N GPUs.  A WORLD_SIZE that contradicts --gpus is refused too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "metric" not in r.stdout
    assert "needs GPU" in r.stderr and "2 ranks requested" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "metric" not in r.stdout


def test_bench_stream_sharding_matches_dist_util():
    """bench.py restates the sharding rule for its torch-free launcher half; BASELINE configs[4]:
    256 streams over 8 ranks = 32 each."""
    import importlib.util
    import dist_util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for total, world in ((256, 8), (256, 4), (256, 2), (256, 1), (5, 2), (7, 8)):
        parts = [bench.shard_streams(total, world, r) for r in range(world)]
        assert parts == [dist_util.shard_streams(total, world, r) for r in range(world)]
    assert all(len(bench.shard_streams(256, 8, r)) == 32 for r in range(8))
    a = bench.parse_args(["--gpus", "8", "--total-streams", "256"])
    assert (a.gpus, a.total_streams, a.steps, a.reps) == (8, 256, 200, 5)


class _ChannelModel:
    """Per-channel bookkeeping of one state, driven only through the library's host-only entry
    points -- what engine.cpp keeps per (stream, channel), incl. the failed-filter fallback
    (reference resample.c:785-791: new rates and advances, old filter length, zeros out)."""

    def __init__(self, ch, in_rate, out_rate, quality):
        self.ch, self.quality = ch, quality
        self.pos = [[0, 0, 0] for _ in range(ch)]  # last_sample, samp_frac_num, magic_samples
        self.started = self.zero = False
        self.line = 0
        info = speexhip.design_filter_frac(in_rate, out_rate, in_rate, out_rate, quality)
        self.rates = (in_rate, out_rate)
        self.num, self.den, self.taps = info["num_rate"], info["den_rate"], info["filt_len"]
        self.line = self.taps - 1 + 160

    def _plan(self, c, frames, cap, float_entry):
        used, made, *self.pos[c] = speexhip.plan_call_ex(self.num, self.den, frames, cap, float_entry,
                                                         self.line - (self.taps - 1), *self.pos[c])
        self.started = self.started or (frames > 0 and cap > 0)
        return used, made

    def _change(self, n, d, i, o, quality):
        from math import gcd
        info_rc = 0
        try:
            info = speexhip.design_filter_frac(n, d, i, o, quality)
        except ValueError as e:  # strerror text of the code
            assert str(e) == "Memory allocation failed.", e
            info_rc, info = 1, None
        g = gcd(n, d)
        new_num, new_den = n // g, d // g
        fracs = []
        for p in self.pos:
            rc, _, _, _, f = speexhip.plan_filter_change(self.taps, self.taps, 0, p[1], self.den, new_den)
            if rc:
                return rc
            fracs.append(f)
        if info is not None and self.started:
            for p in self.pos:
                rc, _shift, magic, delta, _ = speexhip.plan_filter_change(self.taps, info["filt_len"], p[2])
                p[2], p[0] = magic, p[0] + delta
        for p, f in zip(self.pos, fracs):
            p[1] = f
        self.rates, self.num, self.den, self.quality = (i, o), new_num, new_den, quality
        if info is None:
            self.zero = True
            return 1
        self.zero = False
        self.taps = info["filt_len"]
        self.line = max(self.line, self.taps - 1 + 160)
        return 0

    def op(self, op):
        kind = op[0]
        if kind in ("int_ch", "float_ch"):
            used, made = self._plan(op[1], op[2], op[3], kind == "float_ch")
            res = [1 if self.zero else 0, used, made]
        elif kind in ("int", "float"):
            for c in range(self.ch):  # resample.c:1061-1082: the last channel's counters are reported
                used, made = self._plan(c, op[1], op[2], kind == "float")
            res = [1 if self.zero else 0, used, made]
        elif kind in ("rate", "ratefrac"):
            n, d, i, o = (op[1], op[2], op[1], op[2]) if kind == "rate" else op[1:5]
            same = (self.rates, self.num, self.den) == ((i, o), n, d)
            res = [0 if same else self._change(n, d, i, o, self.quality)]
        elif kind == "quality":
            res = [0 if op[1] == self.quality else self._change(self.num, self.den, *self.rates, op[1])]
        elif kind == "skip":
            for p in self.pos:
                p[0] = self.taps // 2
            res = [0]
        else:
            self.pos = [[0, 0, 0] for _ in range(self.ch)]
            res = [0]
        return res + [[list(p) for p in self.pos], self.taps, *self.rates, self.num, self.den]


def test_per_channel_planner_and_failed_filter_counters_match_the_reference_scripts():
    """SURVEY 8 rows a6 / N2 on the CPU: per-channel positions through the product's planner (incl.
    channels advanced unevenly, interleaved calls on the uneven state) and the counters of the
    resampler_basic_zero fallback -- new ratio, old filter length -- over the 624 recorded ops of
    tests/golden/golden_channels.json (digests need the GPU)."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden_channels.json")) as f:
        scripts = json.load(f)["scripts"]
    checked = 0
    for c in scripts:
        m = _ChannelModel(c["channels"], c["in_rate"], c["out_rate"], c["quality"])
        for k, (op, want) in enumerate(zip(c["ops"], c["results"])):
            got = m.op(op)
            want = [w for w in want if not isinstance(w, str)]
            assert got == want, (c["name"], k, op, got, want)
            checked += 1
    assert checked == 624


def test_planner_invariants_over_rates_qualities_and_channel_counts():
    """speexhip_debug_plan (host-only): whatever fast kernel a configuration gets, its geometry must fit
    the machine -- the period kernel's window inside the LDS budget, the slide kernel's smallest
    workgroup inside a CU's 160 KiB (a plan that did not was found on the GPU in round 2: 12:1 q10 on 8
    channels), its rows an even number of iterations (the carry loop runs two per trip)."""
    rates = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 40000, 44100, 48000, 56000, 64000, 72000, 80000,
             88200, 96000, 128000, 160000, 176400, 192000]
    seen = {0: 0, 2: 0, 3: 0}
    for i in rates:
        for o in rates:
            for q in (0, 5, 10):
                for ch in (1, 2, 3, 4, 8):
                    try:
                        t = speexhip.debug_plan(i, o, q, ch)
                    except ValueError:
                        continue  # a filter the reference refuses too (length overflow)
                    seen[t["fast_path"]] += 1
                    if t["fast_path"] == 2:
                        assert t["r_or_p"] in (5, 10) and t["lane_periods"] >= 1, (i, o, q, ch, t)
                        assert t["lds_bytes"] <= 150 * 1024, (i, o, q, ch, t)
                        assert t["pad"] % 4 == 0 and not (t["r_or_p"] == 5 and t["pad"]), (i, o, q, ch, t)
                        # an int16 window is planned only where it holds clearly more periods, never more than
                        # the lanes can take (64 per wave; single-channel lanes carry two periods each)
                        w = t["w16_lane_periods"]
                        assert w == 0 or (4 * w >= 5 * t["lane_periods"] and w <= (128 if ch % 2 else 64 // (ch // 2))), (i, o, q, ch, t)
                    elif t["fast_path"] == 3:
                        assert t["lds_bytes"] <= 160 * 1024, (i, o, q, ch, t)
                        assert t["row_len"] % (2 * t["steps_per_iteration"]) == 0, (i, o, q, ch, t)
                        assert t["r_or_p"] in (1, 2, 4, 8), (i, o, q, ch, t)
    assert min(seen.values()) > 0, seen  # the sweep reached all three outcomes


def test_planner_choices_for_the_named_configurations():
    """The BASELINE configurations and the rules added in round 2."""
    plan = speexhip.debug_plan
    cfg2 = plan(44100, 48000, 7, 2)
    assert (cfg2["fast_path"], cfg2["r_or_p"], cfg2["fine_plan"], cfg2["pad"]) == (2, 10, True, 0)
    cfg4 = plan(48000, 44100, 5, 8)
    assert (cfg4["fast_path"], cfg4["r_or_p"], cfg4["lane_periods"]) == (2, 10, 15) and cfg4["pad"] != 0
    assert plan(24000, 48000, 10, 1)["fast_path"] == 3 and plan(24000, 48000, 5, 1)["r_or_p"] == 8
    # few phases (den <= 80): groups of 5 in every launch
    for i, o in ((44100, 8000), (88200, 48000), (88200, 16000), (176400, 8000)):
        assert plan(i, o, 7, 2)["r_or_p"] == 5, (i, o)
    # wide windows (round 3): int16 calls run over an int16 LDS window with twice the periods per tile; not where
    # the float window already fills the waves (round 6: the layouts without an ISA loop, 9 channels and more, have one too)
    assert plan(48000, 11025, 7, 2)["w16_lane_periods"] >= 2 * plan(48000, 11025, 7, 2)["lane_periods"]
    assert plan(44100, 16000, 7, 2)["w16_lane_periods"] == 64 and plan(48000, 11025, 7, 4)["w16_lane_periods"] == 28
    assert cfg2["w16_lane_periods"] == 0 and cfg4["w16_lane_periods"] == 0
    # (round 5: three channels too -- 18 -> 38 of a tile's 42 periods)
    assert plan(48000, 11025, 7, 3)["w16_lane_periods"] == 38 and plan(48000, 11025, 7, 3)["lane_periods"] == 18
    # n:1 shapes: one period per lane from 16:1 on
    assert plan(192000, 8000, 7, 2)["r_or_p"] == 1 and plan(96000, 8000, 7, 2)["r_or_p"] == 2
    assert plan(48000, 8000, 7, 2)["r_or_p"] == 4 and plan(48000, 24000, 7, 2)["r_or_p"] == 8
    # round 6: 11:1, 7:6, 9:2, 16:3, 25:1 -- den <= 6 outside the slide kernel's shapes -- plan the period kernel on a
    # folded view (110:10, 35:30, 45:10, 80:15, 250:10: whole groups of five phases); a filter no LDS holds stays exact
    for (i, o, q, ch) in ((88000, 8000, 5, 1), (56000, 48000, 4, 2), (72000, 16000, 7, 2), (200000, 8000, 5, 2)):
        got = plan(i, o, q, ch)
        assert (got["fast_path"], got["r_or_p"]) == (2, 5), ((i, o, q, ch), got)
    assert plan(64000, 12000, 7, 1)["fast_path"] == 2 and plan(192000, 1000, 10, 1)["fast_path"] == 0
    assert speexhip.debug_plan64(7, 6, 10, 1)["fast_path"] == 5
    # a filter too long for even a two-wave workgroup falls back to the exact kernel
    # (24:1 of 6 144 taps x 8 channels fits no slide workgroup; since round 6 it runs the period kernel folded to 240:10, int16
    #  calls over the int16 window -- 14 of a tile's 16 periods; 192:1 at quality 10 fits nothing)
    p24 = plan(192000, 8000, 10, 8)
    assert p24["fast_path"] == 2 and p24["w16_lane_periods"] >= 8
    assert plan(192000, 1000, 10, 8)["fast_path"] == 0
    assert plan(192000, 1000, 10, 1)["fast_path"] == 0


def test_generated_fir_loop_is_in_step_with_its_generator():
    """csrc/fir_loop_asm.inc (the period kernel's FIR loop as gfx950 ISA) is committed generator output: it must be
    what csrc/gen_fir_loop.py writes today, and every variant must hold exactly the instructions its shape
    implies -- R x steps packed FMAs per bank, one sample read (two for single-channel lanes) per step, one
    lgkmcnt(0) in front of each bank."""
    import importlib.util
    import re
    path = os.path.join(ROOT, "node-speex-resampler_amd", "csrc", "gen_fir_loop.py")
    spec = importlib.util.spec_from_file_location("gen_fir_loop", path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(os.path.dirname(path), "fir_loop_asm.inc")).read()
    assert committed == (gen.HEAD + "\n".join(v.function() for v in gen.variants()) + gen.HEAD64 +
                         "\n".join(v.function() for v in gen.variants64()) + gen.HEADPP +
                         "\n".join(v.function() for v in gen.variants_pp())), "run python csrc/gen_fir_loop.py"
    for v in gen.variants():
        lines = v.lines()
        loops = 3 if v.R == 10 else 1
        fma = sum(1 for l in lines if l.startswith("v_pk_fma_f32"))
        rows = (5 + 10 + 5) if v.R == 10 else v.R            # rows touched by the head / main / tail copies
        assert fma == 2 * v.S * rows, (v.name, fma)
        reads = sum(1 for l in lines if l.startswith("ds_read"))
        assert reads == (loops * 2 + 1) * v.S * (2 if v.CT == 1 else 1), (v.name, reads)
        assert sum(1 for l in lines if l == "s_waitcnt lgkmcnt(0)") == 2 * loops + 1, v.name
        cvts = sum(1 for l in lines if l.startswith("v_cvt"))
        assert cvts == (loops * 2 * v.S * 2 if v.w16 else 0), (v.name, cvts)
        # taps live in s4..s73 minus the reserved s32, samples in VGPRs the launch bounds leave room for
        used = set(gen.bank_regs(v.banks["A"]) + gen.bank_regs(v.banks["B"]))
        named = set(int(r) for l in lines for pair in re.findall(r"s\[(\d+):(\d+)\]", l) for r in range(int(pair[0]), int(pair[1]) + 1))
        assert named <= used and len(used) == 4 * v.S * v.R // 2 * 1 and 32 not in used and 4 <= min(used) and max(used) <= 73, v.name
        assert max(v.vgprs()) < (64 if v.R == 10 else 128), v.name
    # the fp64-accumulate variants (round 4): one v_fma_f64 per tap and half of the lane's pair, one conversion per
    # half of every sample read, doubles in aligned SGPR pairs of the same homes, aligned VGPR pairs
    assert len(gen.variants64()) == 48   # (round 5: + frames of 3 / 5 / 7 channels, x int16 window)
    for v in gen.variants64():
        lines = v.lines()
        loops = 3 if v.R == 10 else 1
        rows = (5 + 10 + 5) if v.R == 10 else v.R
        assert sum(1 for l in lines if l.startswith("v_fma_f64")) == 2 * v.S * rows * 2, v.name
        assert not any(l.startswith("v_pk_fma_f32") for l in lines), v.name
        assert sum(1 for l in lines if l.startswith("ds_read")) == (loops * 2 + 1) * v.S * (2 if v.CT == 1 else 1), v.name
        # samples widened behind the wait: from floats, or (int16 window, round 5) sign-extended and widened from ints --
        # channel pairs then pay a shift and a bit-field extract per frame, single-channel lanes nothing extra
        widen = "v_cvt_f64_i32" if v.w16 else "v_cvt_f64_f32"
        assert sum(1 for l in lines if l.startswith(widen)) == loops * 2 * v.S * 2, v.name
        assert sum(1 for l in lines if l.startswith("v_cvt")) == loops * 2 * v.S * 2, v.name
        assert sum(1 for l in lines if l.startswith(("v_bfe_i32", "v_ashrrev_i32"))) == (loops * 2 * v.S * 2 if v.w16 and v.CT == 2 else 0), v.name
        assert sum(1 for l in lines if l == "s_waitcnt lgkmcnt(0)") == 2 * loops + 1, v.name
        used = set(gen.bank_regs(v.banks["A"]) + gen.bank_regs(v.banks["B"]))
        assert len(used) == 2 * (2 * v.S * v.R) and 32 not in used and max(used) <= 73, v.name   # two dwords per tap
        for l in lines:
            if l.startswith("v_fma_f64"):
                m = re.match(r"v_fma_f64 %\[a\d+[xy]\], s\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", l)
                assert m and int(m.group(1)) % 2 == 0 and int(m.group(3)) % 2 == 0, (v.name, l)
                assert int(m.group(1)) in used and int(m.group(2)) in used, (v.name, l)
        assert max(v.vgprs()) < (64 if v.R == 10 else 128) and min(v.vgprs()) % 2 == 0, v.name
    # the phase-pair variants (round 4, mono): one packed FMA per PAIR of phases, the tap pair in an aligned SGPR
    # pair, the sample broadcast from the low half of an aligned VGPR pair; one sample read per step
    assert len(gen.variants_pp()) == 18 and sorted(set(v.CF for v in gen.variants_pp())) == [1, 2, 3]
    for v in gen.variants_pp():
        lines = v.lines()
        loops = 3 if v.R == 10 else 1
        rows = (5 + 10 + 5) if v.R == 10 else v.R
        fmas = [l for l in lines if l.startswith("v_pk_fma_f32")]
        assert len(fmas) == 2 * v.S * rows, v.name
        assert all(l.endswith("op_sel:[0,0,0] op_sel_hi:[1,0,1]") for l in fmas), v.name
        for l in fmas:
            m = re.match(r"v_pk_fma_f32 %\[a\d+\], s\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", l)
            assert m and int(m.group(1)) % 2 == 0 and int(m.group(3)) % 2 == 0, (v.name, l)
        assert sum(1 for l in lines if l.startswith("ds_read")) == (loops * 2 + 1) * v.S, v.name
        assert sum(1 for l in lines if l.startswith("v_cvt")) == (loops * 2 * v.S if v.w16 else 0), v.name
        assert sum(1 for l in lines if l == "s_waitcnt lgkmcnt(0)") == 2 * loops + 1, v.name
        assert max(v.vgprs()) < (64 if v.R == 10 else 128), v.name


def test_round4_planners_fp64_accumulate_and_phase_pairs():
    """The plans added in round 4, without a GPU (speexhip_debug_plan64): quality 9 and 10 run the fp64-accumulate twins
    of the fast kernels wherever the fp32 ones run (slide: every shape; period: the ISA loop's layouts), within the
    SGPR / LDS bounds their kernels are built for; mono filters with wide windows (num >= 320) get phase-pair plans
    of 64 periods per tile whose int16 window is taken when it makes room for a second workgroup."""
    rates = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 40000, 44100, 48000, 56000, 64000, 72000, 80000,
             88200, 96000, 128000, 160000, 176400, 192000]
    seen = {0: 0, 4: 0, 5: 0, 6: 0}
    for i in rates:
        for o in rates:
            for q, ch in ((10, 1), (9, 2), (10, 3), (9, 8), (7, 1), (5, 2), (6, 3), (7, 4)):
                try:
                    base = speexhip.debug_plan(i, o, q, ch)
                    t = speexhip.debug_plan64(i, o, q, ch)
                except ValueError:
                    continue
                seen[t["fast_path"]] += 1
                if q >= 9:
                    if base["fast_path"] == 3:
                        assert t["fast_path"] == 4, (i, o, q, ch, base, t)       # every slide shape has its fp64 twin
                        steps = t["last"]
                        assert t["row_len"] % (2 * steps) == 0 and t["lds_bytes"] <= 160 * 1024, (i, o, q, ch, t)
                        den = o // np.gcd(i, o)
                        assert steps * den <= 30, (i, o, q, ch, t)                # one bank of tap doubles in SGPR pairs
                    elif base["fast_path"] == 2:
                        # (round 5: 3 / 5 / 7 channels too; and the fp32 plans run from a ninth of the lanes where the fp64
                        #  plans keep the quarter: 1280:N at quality 9 / 10 rides the fp32 chain, accumulate_bits says 32)
                        full = 64 // (ch // 2 if ch % 2 == 0 else ch) * (1 if ch % 2 == 0 else 2)
                        quarter = 4 * base["lane_periods"] >= full
                        assert t["fast_path"] == (5 if ch in (1, 2, 3, 4, 5, 6, 7, 8) and (quarter or t["fast_path"] == 5) else 0), (i, o, q, ch, base, t)
                        if t["fast_path"] == 5:
                            assert t["r_or_p"] in (5, 10) and t["lds_bytes"] <= 150 * 1024, (i, o, q, ch, t)
                            assert t["row_len"] == t["trips"] * (2 if t["r_or_p"] == 10 else 6), (i, o, q, ch, t)
                            assert not (t["r_or_p"] == 5 and t["pad_or_stride"]), (i, o, q, ch, t)
                    else:
                        assert t["fast_path"] == 0
                else:
                    num = i // np.gcd(i, o)
                    wide = base["fast_path"] == 2 and num >= 320 and ch <= 3    # lane = (period, channel): up to 3 channels
                    assert (t["fast_path"] == 6) == wide, (i, o, q, ch, base, t)
                    if wide:
                        # (from a ninth of a tile, like the other fp32 plans: 64k -> 11.025k mono, 14 of 64 periods)
                        assert 64 // ch // 9 <= t["lane_periods"] <= 64 // ch and t["lds_bytes"] <= 150 * 1024, (i, o, q, ch, t)
                        assert t["last"] in (0,) or t["last"] <= 64, (i, o, q, ch, t)
    assert min(seen.values()) > 0, seen
    p64 = speexhip.debug_plan64
    assert p64(24000, 48000, 10, 1)["fast_path"] == 4 and p64(24000, 48000, 10, 1)["r_or_p"] == 8    # BASELINE configs[2]
    assert p64(44100, 48000, 10, 2)["fast_path"] == 5 and p64(44100, 48000, 10, 3)["fast_path"] == 5   # (3 channels: round 5)
    assert p64(44100, 48000, 10, 9)["fast_path"] == 0                                                      # (no ISA loop for 9)
    assert p64(44100, 48000, 7, 1)["fast_path"] == 0 and p64(48000, 22050, 7, 1)["fast_path"] == 6
    assert 28 <= p64(48000, 22050, 7, 2)["lane_periods"] <= 32 and p64(44100, 32000, 7, 3)["lane_periods"] <= 21
    assert p64(48000, 11025, 7, 1)["last"] >= 60      # its int16 window: two workgroups per CU instead of one


def test_round4_launch_rules_shares_fetch_and_phase_pairs_by_launch():
    """The per-launch rules of the period kernel that round 4 added, without a GPU (speexhip_debug_launch_shape; a process
    without a device plans for 256 CUs): tap-range shares on UNSPLIT launches whose workgroups have <= 8 waves (R = 10
    only), the tap rows fetched behind the window where the launch moves >= 24 MB with >= 128 KB of rows or runs
    unsplit phase pairs with shares over >= 256 KB of rows, and phase pairs for stereo where the other plan has to
    split its tiles.  The figures behind every threshold: DESIGN.md 3.3, profiles/r04_ks_unsplit_ab.txt,
    r04_touch_ab3.txt, r04_rule4_ab.txt; tests/test_gpu_perf_gate.py holds the times."""
    from math import gcd

    def shape(ch, i, o, streams, frames, q=7, float_io=False):
        g = gcd(i, o)
        return speexhip.debug_launch_shape(i // g, o // g, q, ch, streams, frames, float_io)

    # three channels 48k -> 11.025k: phase pairs, 8 groups of 20 phases on 8 waves -> two shares each, rows fetched
    t = shape(3, 48000, 11025, 32, 131072)
    assert t["phase_pairs"] and t["r"] == 10 and t["splits"] == 1 and t["wave_groups"] == 8, t
    assert t["shares"] == 2 and t["threads"] == 1024 and t["touch"], t
    # (round 5: launches of several generations run the two-period plan over its int16 window, 38 of 42 periods per tile --
    #  until then that plan had only a float window, 18 periods, and phase pairs took these too: profiles/r05_w16_3ch.txt)
    t = shape(3, 48000, 11025, 32, 1 << 20)
    assert not t["phase_pairs"] and t["int16_window"] and t["lane_periods"] == 38 and t["wave_groups"] == 15 and t["touch"], t
    # stereo 48k -> 11.025k: the other plan would split its 150 KB tiles -> phase pairs, shares, rows (415 KB) fetched
    t = shape(2, 48000, 11025, 32, 131072)
    assert t["phase_pairs"] and t["splits"] == 1 and t["shares"] == 2 and t["touch"], t
    # ... but one stream of it is a split launch of the other plan (phase pairs only in batches)
    assert not shape(2, 48000, 11025, 1, 441000)["phase_pairs"]
    # stereo 48k -> 22.05k: phase pairs with shares, 210 KB of rows and 21 MB moved -> no fetch
    t = shape(2, 48000, 22050, 32, 131072)
    assert t["phase_pairs"] and t["shares"] == 2 and not t["touch"], t
    # R = 5 plans take no shares on unsplit launches (their instances with shares need 76 VGPRs)
    t = shape(2, 44100, 8000, 32, 131072)
    assert t["r"] == 5 and t["splits"] == 1 and t["shares"] == 1, t
    # 4 and 6 channels: channel pairs; 33 / 50 MB moved -> rows fetched; 6 channels 44.1k -> 8k has 8 waves -> shares
    t = shape(4, 48000, 11025, 32, 131072)
    assert not t["phase_pairs"] and t["wave_groups"] == 15 and t["shares"] == 1 and t["touch"], t
    t = shape(6, 44100, 8000, 32, 131072)
    assert not t["phase_pairs"] and t["wave_groups"] == 8 and t["shares"] == 2 and t["touch"], t
    # mono 48k -> 22.05k: unsplit, 8 waves -> shares; 10 MB moved -> no fetch
    t = shape(1, 48000, 22050, 32, 131072)
    assert t["splits"] == 1 and t["shares"] == 2 and not t["touch"], t
    # the BASELINE launches are what they were: no shares on the 16-wave workgroups of cfg2, no fetch of 90 KB of rows
    for streams in (1, 32):
        t = shape(2, 44100, 48000, streams, 1 << 20)
        assert not t["phase_pairs"] and t["shares"] == 1 and not t["touch"], t
    t = shape(8, 48000, 44100, 32, 1 << 20, q=5)
    assert t["shares"] == 1 and not t["touch"] and t["wave_groups"] == 15, t
    # configurations that do not run the fp32 period kernel answer with zeros
    assert shape(1, 24000, 48000, 1, 1 << 20, q=10)["r"] == 0 and shape(2, 44100, 48000, 1, 4096, q=10)["r"] == 0


def test_round5_shares_of_wide_window_launches_count_generations():
    """Round 5: a window of more than half the LDS means one workgroup per CU, and the history-roll block of every
    (stream, share) is a workgroup too.  Where the doubling rule's count -- which neither sees those blocks nor knows
    odd counts -- does not fit one generation (256 CUs without a device), the shares come from a model of the
    generations (launch_period_plan; profiles/r05_wide_grid.txt, r05_ab_split_model.txt); inside one generation the
    launch is what rounds 3-4 fitted."""
    from math import gcd

    def shape(ch, i, o, streams, frames, q=7):
        g = gcd(i, o)
        return speexhip.debug_launch_shape(i // g, o // g, q, ch, streams, frames, False)

    # 4 channels 32k -> 11.025k, 8 x 131 072 frames: 8 tiles x 8 streams x 4 shares = 256 workgroups + 8 that roll histories -> 3 shares
    t = shape(4, 32000, 11025, 8, 131072)
    assert t["tiles"] == 8 and t["splits"] == 3 and t["wave_groups"] == 15, t
    # ... 32 streams: 256 + 32 workgroups unsplit (two generations, the second nearly empty) -> 2 shares
    t = shape(4, 32000, 11025, 32, 131072)
    assert t["splits"] == 2, t
    # mono, 32 streams in 2 tiles each: 4 shares were 256 tile workgroups + 32 that roll histories -> 3 (192 + 32)
    t = shape(1, 32000, 11025, 32, 131072)
    assert t["phase_pairs"] and t["tiles"] == 2 and t["splits"] == 3, t
    # one generation: untouched (one stream of 48k -> 11.025k mono: 29 tiles + 1 in 8 shares = 240 workgroups, tap-range shares)
    t = shape(1, 48000, 11025, 1, 1 << 20)
    assert t["splits"] == 8 and t["shares"] == 8, t
    # narrow windows (two workgroups per CU) never take this path: BASELINE configs[1]
    assert shape(2, 44100, 48000, 1, 1 << 20)["splits"] == 2 and shape(2, 44100, 48000, 32, 1 << 20)["splits"] == 1


def test_round5_plans_for_wide_frames_and_plans_that_stand_for_their_int16_window():
    """Late in round 5 (host-only planner checks): frames of 10 / 12 / 16 channels have ISA loops (kernels_period_frames.hip) and
    with them an int16-window plan; a float-window plan under the fill rule stands when its int16-window plan passes it; and
    where not even one period of the float window fits (lds_bytes reported as 0) the plan stands for its int16 plan alone --
    float calls of such a state run the exact kernel (tests/test_gpu_parity.py walks that)."""
    plan = speexhip.debug_plan
    # 10 / 12 / 16 channels, a wide window: twice the periods per tile over int16
    for ch, lp, lp16 in ((10, 5, 11), (12, 4, 9), (16, 2, 6)):
        t = plan(640, 147, 7, ch)
        assert (t["fast_path"], t["r_or_p"], t["lane_periods"], t["w16_lane_periods"]) == (2, 10, lp, lp16), (ch, t)
        assert plan(147, 160, 7, ch)["fast_path"] == 2 and plan(147, 160, 7, ch)["w16_lane_periods"] == 0, ch  # narrow: float window
    # 9 channels (no ISA loop): no int16 window until round 6; now the C++ loop reads one (kernels_period_w16g.hip):
    # twice the periods per tile on wide windows, none where the float window already fills the waves
    for ch in (9, 11, 13, 14, 15, 17, 20):
        t = plan(640, 147, 7, ch)
        assert t["fast_path"] == 2 and t["w16_lane_periods"] >= 2 * t["lane_periods"] - 1 > 0, (ch, t)
        assert plan(147, 160, 7, ch)["w16_lane_periods"] == 0, ch
    # 8 channels of 2 232 taps at num = 1280: one period of the float window (a sixteenth of a tile), five of the int16 one
    t = plan(1280, 147, 10, 8)
    assert (t["fast_path"], t["lane_periods"], t["w16_lane_periods"]) == (2, 1, 5) and t["lds_bytes"] > 0, t
    # 16 channels at num = 1280: no float window at all, two periods of the int16 one
    t = plan(1280, 147, 7, 16)
    assert (t["fast_path"], t["lane_periods"], t["w16_lane_periods"], t["lds_bytes"]) == (2, 1, 2, 0), t
    sh = speexhip.debug_launch_shape(1280, 147, 7, 16, 32, 131072)
    assert sh["int16_window"] and sh["lane_periods"] == 2, sh
    # ... and a float call of it has nothing to launch from that plan: the shape hook answers for int16 calls only
    assert speexhip.debug_launch_shape(1280, 147, 7, 16, 32, 131072, True)["r"] == 0


def test_device_placement_rule():
    """Round 5: which GPU a new state lives on (csrc/devices.cpp) as a pure function of the device count, the two
    environment variables and the state's number in its process -- SPEEXHIP_DEVICES=all is BASELINE configs[4]'s
    "stream s on GPU s mod 8" inside ONE process (the reference's model: many instances, one module)."""
    P = speexhip.placement
    assert [P(8, None, "all", k) for k in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert [P(8, None, "0,2,5", k) for k in range(7)] == [0, 2, 5, 0, 2, 5, 0]
    assert [P(8, None, " 1, 0", k) for k in range(3)] == [1, 0, 1]
    assert [P(8, "3", "all", k) for k in range(4)] == [3, 3, 3, 3]        # SPEEXHIP_DEVICE wins
    assert [P(8, None, None, k, current=6) for k in range(3)] == [6, 6, 6]  # neither: the thread's current device
    assert P(8, "", "", 4, current=2) == 2                                  # empty = unset
    # what the node does not have, or cannot be read: -1 (init then fails with SPEEXHIP_ERR_DEVICE)
    for bad in [(8, "8", None), (8, "-1", None), (8, "x", None), (8, None, "0,8"), (8, None, "0,,1"), (8, None, "all,1"),
                (0, None, "all"), (-1, None, None)]:
        assert P(bad[0], bad[1], bad[2], 0) == -1, bad
    # 256 streams over 8 GPUs: 32 each
    counts = [0] * 8
    for k in range(256):
        counts[P(8, None, "all", k)] += 1
    assert counts == [32] * 8


def test_device_placement_by_live_state_count_with_create_destroy_churn():
    """Round 6: SPEEXHIP_DEVICES=all places a new state on the device with the fewest LIVE states --
    destroying a state frees its slot -- with ties broken so that a fresh process still deals round-robin; a list and a
    single device keep the counter rule; the no-environment default stays on the thread's current device.  Simulated
    process: a server whose connections open and close at random."""
    PL = speexhip.placement_live
    n = 8
    live = [0] * n
    # a fresh process: exactly the round-robin of the counter rule
    order = []
    for k in range(2 * n):
        d = PL(n, None, "all", k, 0, live)
        live[d] += 1
        order.append(d)
    assert order == [k % n for k in range(2 * n)]
    # churn: 3000 events, 40 % closes of a random live state; the spread never exceeds what the closes themselves opened
    rng = np.random.RandomState(5)
    owner, k = [], 2 * n
    for d in order:
        owner.append(d)
    worst_after_open = 0
    for _ in range(3000):
        if owner and rng.rand() < 0.4:
            d = owner.pop(rng.randint(len(owner)))
            live[d] -= 1
        else:
            before_min = min(live)
            d = PL(n, None, "all", k, 3, live)
            assert live[d] == before_min, "a new state must land on a least-loaded device"
            live[d] += 1
            owner.append(d)
            k += 1
            worst_after_open = max(worst_after_open, max(live) - min(live))
    # only opens for a while: the holes are filled before anything piles up
    for _ in range(4 * n):
        d = PL(n, None, "all", k, 3, live)
        live[d] += 1
        k += 1
    assert max(live) - min(live) <= 1, live
    # the counter rule where the caller asked for determinism, whatever the load
    skew = [100, 0, 0, 0, 0, 0, 0, 0]
    assert [PL(n, None, "0,2,5", j, 0, skew) for j in range(4)] == [0, 2, 5, 0]
    assert [PL(n, "0", "all", j, 4, skew) for j in range(3)] == [0, 0, 0]
    assert [PL(n, None, None, j, 6, skew) for j in range(3)] == [6, 6, 6]
    assert PL(n, None, "all", 0, 0, skew) == 1 and PL(n, None, "all", 5, 0, skew) == 5  # ties: k mod n first
    assert PL(0, None, "all", 0, 0, []) == -1


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_batch_class_and_many_call_check_their_arguments_without_a_gpu():
    """Round 5's JS surface on a box without a GPU: the argument checks of SpeexResamplerBatch and of the addon's
    processMany, the reference's messages through the batch, the init failure surfacing as Error(strerror) -- and the
    addon loading in a worker_threads Worker (NAPI_MODULE_INIT), with initPromise resolving behind a warm-up that fails
    quietly (no GPU is reported by the first real call, like every error of the reference)."""
    index = os.path.join(ROOT, "node-speex-resampler_amd", "index.js")
    if not os.path.exists(os.path.join(ROOT, "node-speex-resampler_amd", "speex_hip_napi.node")):
        pytest.skip("addon not built")
    js = r"""
const R = require(process.argv[1]);
const addon = require(require('path').join(require('path').dirname(process.argv[1]), 'speex_hip_napi.node'));
const out = {};
const grab = (k, f) => { try { const v = f(); out[k] = v === undefined ? 'no throw' : v; } catch (e) { out[k] = e.constructor.name + ': ' + e.message; } };
R.default.initPromise.then(async () => {
  grab('n0', () => new R.SpeexResamplerBatch(0, 2, 44100, 48000));
  const b = new R.SpeexResamplerBatch(3, 2, 44100, 48000, 7, { devices: [0] });
  out.fields = [b.length, b.channels, b.inRate, b.outRate, b.quality, b.streams.length];
  grab('count', () => b.processChunks([Buffer.alloc(8)]));
  grab('align', () => b.processChunks([Buffer.alloc(8), Buffer.alloc(7), Buffer.alloc(8)]));
  grab('nogpu', () => b.processChunks([Buffer.alloc(8), null, Buffer.alloc(8)]));
  out.asyncNoGpu = await b.processChunksAsync([Buffer.alloc(8), Buffer.alloc(8), Buffer.alloc(8)]).then(() => 'resolved', (e) => e.message);
  grab('allNull', () => JSON.stringify(b.processChunks([null, null, null])));
  grab('manyTypes', () => addon.processMany(1, 2, 3, 4));
  grab('manyHandle', () => addon.processMany([{}], [Buffer.alloc(4)], [1], [2]));
  grab('manyEmpty', () => addon.processMany([], [], [], []).length);
  grab('deviceCount', () => typeof R.default.deviceCount());
  grab('mode', () => new R.default(2, 44100, 48000).setMode('fastest'));
  const { Worker } = require('worker_threads');
  out.worker = await new Promise((res) => {
    const w = new Worker(`
      const { parentPort, workerData } = require('worker_threads');
      const R = require(workerData);
      R.default.initPromise.then(() => {
        let m; try { new R.default(2, 44100, 48000).processChunk(Buffer.alloc(7)); } catch (e) { m = e.message; }
        parentPort.postMessage(m);
      });`, { eval: true, workerData: process.argv[1] });
    w.on('message', res); w.on('error', (e) => res('worker error: ' + e.message));
  });
  console.log(JSON.stringify(out));
});
"""
    res = subprocess.run(["node", "-e", js, index], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    import json
    out = json.loads(res.stdout)
    assert out["n0"].endswith("nStreams must be a positive integer")
    assert out["fields"] == [3, 2, 44100, 48000, 7, 3]
    assert "one chunk (or null) per stream: 3" in out["count"]
    assert out["align"] == "Error: Chunk length should be a multiple of channels * 2 bytes"
    assert out["nogpu"].startswith("Error: HIP device error"), out["nogpu"]     # no CPU fallback, through the batch too
    assert out["asyncNoGpu"].startswith("HIP device error"), out["asyncNoGpu"]
    assert out["allNull"] == "[null,null,null]"
    assert out["manyTypes"].startswith("TypeError") and out["manyHandle"].startswith("TypeError")
    assert out["manyEmpty"] == 0 and out["deviceCount"] == "number"
    assert "mode must be" in out["mode"]
    assert out["worker"] == "Chunk length should be a multiple of channels * 2 bytes"


def test_product_library_reads_only_the_documented_environment():
    """a stray environment variable must not be able to change a drop-in's bytes.  The experiment
    switches of rounds 2-5 (SPEEXHIP_R, _KSPLIT, _SKIP, _PIECES ...) exist only in the diagnostics build
    (csrc/diag.h, ab/libspeexhip_diag.so); the shipped library, the addon and index.js read exactly the list below --
    placement, memory limits, start-up: nothing that changes a sample except SPEEXHIP_MODE, which is the documented
    way to pick the numerical contract."""
    pkg = os.path.join(ROOT, "node-speex-resampler_amd")

    def names(path):
        return set(m.decode() for m in re.findall(rb"SPEEXHIP_[A-Z0-9_]+", open(path, "rb").read()))

    constants = re.compile(r"SPEEXHIP_(ERR|MODE|KERNEL)_\w+|SPEEXHIP_API|SPEEXHIP_RESAMPLER_H")
    lib = {n for n in names(os.path.join(pkg, "libspeexhip.so")) if not constants.fullmatch(n)}
    assert lib == {"SPEEXHIP_DEVICE", "SPEEXHIP_DEVICES", "SPEEXHIP_ALIAS_DEVICES", "SPEEXHIP_MODE", "SPEEXHIP_POOL_MB",
                   "SPEEXHIP_TAKE_MB", "SPEEXHIP_TAKE_MAX_MB"}, sorted(lib)
    getenv = re.compile(r'getenv\("(SPEEXHIP_\w+)"\)|process\.env\.(SPEEXHIP_\w+)')
    read = set()
    for rel in ("napi/speex_hip_napi.c", "index.js"):
        for m in getenv.finditer(open(os.path.join(pkg, rel)).read()):
            read.add(m.group(1) or m.group(2))
    assert read <= {"SPEEXHIP_NAPI_COPY", "SPEEXHIP_NO_WARMUP", "SPEEXHIP_MODE", "SPEEXHIP_DEVICES"}, sorted(read)
    # ... and the diagnostics build is where the switches went
    diag = os.path.join(pkg, "ab", "libspeexhip_diag.so")
    assert os.path.exists(diag), "make diag"
    assert {"SPEEXHIP_SKIP", "SPEEXHIP_KSPLIT", "SPEEXHIP_PIECES", "SPEEXHIP_R"} <= names(diag)
