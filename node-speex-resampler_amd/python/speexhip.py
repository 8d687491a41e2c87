"""ctypes binding of libspeexhip.so (include/speexhip_resampler.h) plus a Python mirror of the
reference's host class (``SpeexResampler.processChunk``, reference src/index.ts:21-117) so that
the parity tests and bench.py can drive the HIP path without Node.

There is NO fallback: if the library is missing or no MI355X is usable, calls raise.
"""
import ctypes as C
import math
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (SPEEXHIP_LIB_PATH: same-box A/B of two builds of the library, tools/ab.sh and tools/lease.sh lib-ab)
LIB_PATH = os.environ.get("SPEEXHIP_LIB_PATH") or os.path.join(PKG_DIR, "libspeexhip.so")

MODE_FAST, MODE_EXACT, MODE_FAST_F32, MODE_FAST_FIXED = 0, 1, 2, 3
KERNEL_NAMES = ("direct_single", "direct_double", "interpolate_single", "interpolate_double")
# reference codes (deps/speex/speex_resampler.h:104-113) + 6 = HIP failure
ERR_SUCCESS, ERR_ALLOC_FAILED, ERR_BAD_STATE, ERR_INVALID_ARG, ERR_PTR_OVERLAP, ERR_OVERFLOW = 0, 1, 2, 3, 4, 5
ERR_DEVICE = 6
ERR_NO_BLOCK = 7  # the ..._take calls: no pinned result block free right now, state untouched

EXPORTS = [
    "speexhip_resampler_init", "speexhip_resampler_destroy",
    "speexhip_resampler_process_interleaved_int", "speexhip_resampler_process_interleaved_float",
    "speexhip_resampler_process_interleaved_float_device",
    "speexhip_batch_process_interleaved_float_device", "speexhip_resampler_get_rate",
    "speexhip_resampler_strerror", "speexhip_resampler_process_interleaved_int_device",
    "speexhip_resampler_set_mode", "speexhip_resampler_get_info", "speexhip_resampler_get_history",
    "speexhip_batch_init", "speexhip_batch_destroy", "speexhip_batch_set_mode",
    "speexhip_batch_get_info", "speexhip_batch_process_interleaved_int_device",
    "speexhip_design_filter", "speexhip_plan_call", "speexhip_version",
    # mid-stream control
    "speexhip_resampler_init_frac", "speexhip_resampler_set_rate", "speexhip_resampler_set_rate_frac",
    "speexhip_resampler_get_ratio", "speexhip_resampler_set_quality", "speexhip_resampler_get_quality",
    "speexhip_resampler_get_input_latency", "speexhip_resampler_get_output_latency",
    "speexhip_resampler_skip_zeros", "speexhip_resampler_reset_mem",
    "speexhip_batch_set_rate_frac", "speexhip_batch_set_quality", "speexhip_batch_skip_zeros",
    "speexhip_batch_reset_mem", "speexhip_batch_get_history",
    "speexhip_design_filter_frac", "speexhip_plan_call_ex", "speexhip_plan_filter_change",
    # chunk coalescing
    "speexhip_resampler_process_chunks_int", "speexhip_resampler_process_chunks_float",
    "speexhip_resampler_peek",
    # per-channel entry points + strides (rest of row N2), the zero fallback's test hook (row a6)
    "speexhip_resampler_process_int", "speexhip_resampler_process_float",
    "speexhip_resampler_set_input_stride", "speexhip_resampler_get_input_stride",
    "speexhip_resampler_set_output_stride", "speexhip_resampler_get_output_stride",
    "speexhip_resampler_get_channel_position", "speexhip_debug_fail_device_allocs",
    "speexhip_release_cached_memory", "speexhip_debug_plan",
    "speexhip_resampler_release_stream", "speexhip_batch_release_stream", "speexhip_debug_device_clock",
    "speexhip_resampler_process_interleaved_int_take", "speexhip_resampler_process_interleaved_float_take",
    "speexhip_block_release", "speexhip_debug_plan64", "speexhip_debug_launch_shape",
    # round 5: device placement, many states per call
    "speexhip_device_count", "speexhip_resampler_init_on", "speexhip_batch_init_on",
    "speexhip_resampler_process_many_int", "speexhip_resampler_process_many_float",
    "speexhip_resampler_get_info2", "speexhip_debug_placement", "speexhip_warmup",
    # round 6: pinned blocks the caller fills (inputs used in place)
    "speexhip_block_acquire", "speexhip_debug_pcie_peak", "speexhip_debug_placement_live", "speexhip_debug_live_states",
]


class Info(C.Structure):
    _fields_ = [("in_rate", C.c_uint32), ("out_rate", C.c_uint32), ("num_rate", C.c_uint32),
                ("den_rate", C.c_uint32), ("nb_channels", C.c_uint32), ("quality", C.c_int32),
                ("filt_len", C.c_uint32), ("oversample", C.c_uint32),
                ("sinc_table_length", C.c_uint32), ("kernel", C.c_int32), ("mode", C.c_int32),
                ("fast_path", C.c_int32), ("last_sample", C.c_int32), ("samp_frac_num", C.c_uint32),
                ("device", C.c_int32), ("magic_samples", C.c_uint32), ("block_in", C.c_uint32),
                ("accumulate_bits", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """Load libspeexhip.so or raise (the product path never degrades to a CPU implementation)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libspeexhip.so not built: run `python __graft_entry__.py` or "
                               "`make -C node-speex-resampler_amd` (no CPU fallback exists)")
        # One HIP runtime per process: PyTorch-ROCm preloads its bundled libamdhip64.so.7 by path;
        # if ours pulled in /opt/rocm's copy first there would be two runtimes and the second
        # to initialise sees no device.  Loading torch first makes both share one copy.
        # (SPEEXHIP_PY_NO_TORCH=1, tools only: load the library behind /opt/rocm's runtime, as the Node addon does --
        #  the two runtimes differ, e.g. in whether pinned copies of opposite directions overlap, profiles/r06_runtime_ab.txt)
        if os.environ.get("SPEEXHIP_PY_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        u32, i32, p = C.c_uint32, C.c_int, C.c_void_p
        pu32, pi16 = C.POINTER(C.c_uint32), C.POINTER(C.c_int16)
        L.speexhip_resampler_init.restype = p
        L.speexhip_resampler_init.argtypes = [u32, u32, u32, i32, C.POINTER(C.c_int)]
        L.speexhip_resampler_destroy.argtypes = [p]
        L.speexhip_resampler_process_interleaved_int.restype = i32
        L.speexhip_resampler_process_interleaved_int.argtypes = [p, pi16, pu32, pi16, pu32]
        L.speexhip_resampler_process_interleaved_int_device.restype = i32
        L.speexhip_resampler_process_interleaved_int_device.argtypes = [p, p, pu32, p, pu32, p]
        pf32 = C.POINTER(C.c_float)
        L.speexhip_resampler_process_interleaved_float.restype = i32
        L.speexhip_resampler_process_interleaved_float.argtypes = [p, pf32, pu32, pf32, pu32]
        L.speexhip_resampler_process_interleaved_float_device.restype = i32
        L.speexhip_resampler_process_interleaved_float_device.argtypes = [p, p, pu32, p, pu32, p]
        L.speexhip_batch_process_interleaved_float_device.restype = i32
        L.speexhip_batch_process_interleaved_float_device.argtypes = [p, p, C.c_uint64, pu32, p,
                                                                      C.c_uint64, pu32, p]
        L.speexhip_resampler_get_rate.argtypes = [p, pu32, pu32]
        L.speexhip_resampler_strerror.restype = C.c_char_p
        L.speexhip_resampler_strerror.argtypes = [i32]
        L.speexhip_resampler_set_mode.restype = i32
        L.speexhip_resampler_set_mode.argtypes = [p, i32]
        L.speexhip_resampler_get_info.restype = i32
        L.speexhip_resampler_get_info.argtypes = [p, C.POINTER(Info)]
        L.speexhip_resampler_get_history.restype = i32
        L.speexhip_resampler_get_history.argtypes = [p, C.POINTER(C.c_float)]
        L.speexhip_batch_init.restype = p
        L.speexhip_batch_init.argtypes = [u32, u32, u32, u32, i32, C.POINTER(C.c_int)]
        L.speexhip_batch_destroy.argtypes = [p]
        L.speexhip_batch_set_mode.restype = i32
        L.speexhip_batch_set_mode.argtypes = [p, i32]
        # (an older build of the library loaded through SPEEXHIP_LIB_PATH for a same-box A/B lacks the entry points
        #  of later rounds: they stay unbound there)
        if hasattr(L, "speexhip_block_release") or "SPEEXHIP_LIB_PATH" not in os.environ:
            for fn, st in ((L.speexhip_resampler_process_interleaved_int_take, C.c_int16),
                           (L.speexhip_resampler_process_interleaved_float_take, C.c_float)):
                fn.restype = i32
                fn.argtypes = [p, C.c_void_p, pu32, pu32, C.POINTER(C.POINTER(st))]
            L.speexhip_block_release.restype = None
            L.speexhip_block_release.argtypes = [C.c_void_p]
            L.speexhip_debug_device_clock.restype = i32
            L.speexhip_debug_device_clock.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
            L.speexhip_resampler_release_stream.restype = i32
            L.speexhip_resampler_release_stream.argtypes = [p]
            L.speexhip_batch_release_stream.restype = i32
            L.speexhip_batch_release_stream.argtypes = [p]
        L.speexhip_batch_get_info.restype = i32
        L.speexhip_batch_get_info.argtypes = [p, u32, C.POINTER(Info)]
        L.speexhip_batch_process_interleaved_int_device.restype = i32
        L.speexhip_batch_process_interleaved_int_device.argtypes = [p, p, C.c_uint64, pu32, p,
                                                                    C.c_uint64, pu32, p]
        L.speexhip_design_filter.restype = i32
        L.speexhip_design_filter.argtypes = [u32, u32, i32, C.POINTER(Info), C.POINTER(C.c_float), u32]
        L.speexhip_plan_call.restype = i32
        L.speexhip_plan_call.argtypes = [u32, u32, u32, u32, C.POINTER(C.c_int32), pu32, pu32, pu32]
        L.speexhip_version.restype = C.c_char_p
        pi32 = C.POINTER(C.c_int32)
        L.speexhip_resampler_init_frac.restype = p
        L.speexhip_resampler_init_frac.argtypes = [u32, u32, u32, u32, u32, i32, C.POINTER(C.c_int)]
        L.speexhip_resampler_set_rate.argtypes = [p, u32, u32]
        L.speexhip_resampler_set_rate_frac.argtypes = [p, u32, u32, u32, u32]
        L.speexhip_resampler_get_ratio.argtypes = [p, pu32, pu32]
        L.speexhip_resampler_set_quality.argtypes = [p, i32]
        L.speexhip_resampler_get_quality.argtypes = [p, C.POINTER(C.c_int)]
        for f in (L.speexhip_resampler_get_input_latency, L.speexhip_resampler_get_output_latency,
                  L.speexhip_resampler_skip_zeros, L.speexhip_resampler_reset_mem,
                  L.speexhip_batch_skip_zeros, L.speexhip_batch_reset_mem):
            f.restype = i32
            f.argtypes = [p]
        L.speexhip_batch_set_rate_frac.argtypes = [p, u32, u32, u32, u32]
        L.speexhip_batch_set_quality.argtypes = [p, i32]
        L.speexhip_batch_get_history.argtypes = [p, u32, C.POINTER(C.c_float)]
        L.speexhip_design_filter_frac.restype = i32
        L.speexhip_design_filter_frac.argtypes = [u32, u32, u32, u32, i32, C.POINTER(Info),
                                                  C.POINTER(C.c_float), u32]
        L.speexhip_plan_call_ex.restype = i32
        L.speexhip_plan_call_ex.argtypes = [u32, u32, u32, u32, i32, u32, pi32, pu32, pu32, pu32, pu32]
        L.speexhip_plan_filter_change.restype = i32
        L.speexhip_plan_filter_change.argtypes = [u32, u32, u32, C.POINTER(C.c_int64), pu32, pi32, pu32, u32, u32]
        for f in (L.speexhip_resampler_process_chunks_int, L.speexhip_resampler_process_chunks_float):
            f.restype = i32
            f.argtypes = [p, u32, C.POINTER(C.c_void_p), pu32, p, pu32]
        L.speexhip_resampler_peek.restype = i32
        L.speexhip_resampler_peek.argtypes = [p, u32, u32, i32, pu32, pu32]
        L.speexhip_resampler_process_int.restype = i32
        L.speexhip_resampler_process_int.argtypes = [p, u32, pi16, pu32, pi16, pu32]
        L.speexhip_resampler_process_float.restype = i32
        L.speexhip_resampler_process_float.argtypes = [p, u32, pf32, pu32, pf32, pu32]
        for f in (L.speexhip_resampler_set_input_stride, L.speexhip_resampler_set_output_stride):
            f.restype = None
            f.argtypes = [p, u32]
        for f in (L.speexhip_resampler_get_input_stride, L.speexhip_resampler_get_output_stride):
            f.restype = None
            f.argtypes = [p, pu32]
        L.speexhip_resampler_get_channel_position.restype = i32
        L.speexhip_resampler_get_channel_position.argtypes = [p, u32, pi32, pu32, pu32]
        L.speexhip_debug_fail_device_allocs.restype = None
        L.speexhip_debug_fail_device_allocs.argtypes = [i32]
        L.speexhip_debug_plan.restype = C.c_int
        L.speexhip_debug_plan.argtypes = [u32, u32, i32, u32, C.POINTER(u32)]
        L.speexhip_release_cached_memory.restype = C.c_uint64
        L.speexhip_release_cached_memory.argtypes = []
        if hasattr(L, "speexhip_resampler_process_many_int") or "SPEEXHIP_LIB_PATH" not in os.environ:
            L.speexhip_device_count.restype = i32
            L.speexhip_device_count.argtypes = []
            L.speexhip_resampler_init_on.restype = p
            L.speexhip_resampler_init_on.argtypes = [i32, u32, u32, u32, i32, C.POINTER(C.c_int)]
            L.speexhip_batch_init_on.restype = p
            L.speexhip_batch_init_on.argtypes = [i32, u32, u32, u32, u32, i32, C.POINTER(C.c_int)]
            for f in (L.speexhip_resampler_process_many_int, L.speexhip_resampler_process_many_float):
                f.restype = i32
                f.argtypes = [u32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), pu32, C.POINTER(C.c_void_p), pu32,
                              C.POINTER(C.c_int)]
            L.speexhip_resampler_get_info2.restype = i32
            L.speexhip_resampler_get_info2.argtypes = [p, C.c_void_p, u32]
            L.speexhip_warmup.restype = i32
            L.speexhip_warmup.argtypes = [i32]
            L.speexhip_debug_placement.restype = i32
            L.speexhip_debug_placement.argtypes = [i32, C.c_char_p, C.c_char_p, C.c_uint64, i32]
        if hasattr(L, "speexhip_block_acquire") or "SPEEXHIP_LIB_PATH" not in os.environ:
            L.speexhip_block_acquire.restype = C.c_void_p
            L.speexhip_block_acquire.argtypes = [C.c_uint64]
            L.speexhip_debug_placement_live.restype = i32
            L.speexhip_debug_placement_live.argtypes = [i32, C.c_char_p, C.c_char_p, C.c_uint64, i32, C.POINTER(C.c_uint32)]
            L.speexhip_debug_live_states.restype = C.c_uint32
            L.speexhip_debug_live_states.argtypes = [i32]
            L.speexhip_debug_pcie_peak.restype = i32
            L.speexhip_debug_pcie_peak.argtypes = [C.c_uint64, i32, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def strerror(code):
    return lib().speexhip_resampler_strerror(code).decode()


def design_filter(in_rate, out_rate, quality, want_table=True):
    """Host-only filter design; returns (info dict, table float32 array or None)."""
    info = Info()
    rc = lib().speexhip_design_filter(in_rate, out_rate, quality, C.byref(info), None, 0)
    if rc != 0:
        raise ValueError(strerror(rc))
    table = None
    if want_table:
        table = np.zeros(info.sinc_table_length, np.float32)
        rc = lib().speexhip_design_filter(in_rate, out_rate, quality, C.byref(info),
                                          table.ctypes.data_as(C.POINTER(C.c_float)), table.size)
        assert rc == 0
    return info.as_dict(), table


def plan_call(num, den, in_len, out_cap, last, frac):
    """Host-only stream bookkeeping; returns (consumed, produced, last', frac')."""
    l, f, c, p = C.c_int32(last), C.c_uint32(frac), C.c_uint32(), C.c_uint32()
    rc = lib().speexhip_plan_call(num, den, in_len, out_cap, C.byref(l), C.byref(f), C.byref(c),
                                  C.byref(p))
    if rc != 0:
        raise ValueError(strerror(rc))
    return c.value, p.value, l.value, f.value


def design_filter_frac(ratio_num, ratio_den, in_rate, out_rate, quality):
    """Host-only filter geometry for a ratio given separately from the rates; info dict."""
    info = Info()
    rc = lib().speexhip_design_filter_frac(ratio_num, ratio_den, in_rate, out_rate, quality,
                                           C.byref(info), None, 0)
    if rc != 0:
        raise ValueError(strerror(rc))
    return info.as_dict()


def plan_call_ex(num, den, in_len, out_cap, float_entry, block_in, last, frac, magic):
    """Host-only bookkeeping of one call for any entry point / state;
    returns (consumed, produced, last', frac', magic')."""
    l, f, m, c, p = C.c_int32(last), C.c_uint32(frac), C.c_uint32(magic), C.c_uint32(), C.c_uint32()
    rc = lib().speexhip_plan_call_ex(num, den, in_len, out_cap, int(float_entry), block_in, C.byref(l),
                                     C.byref(f), C.byref(m), C.byref(c), C.byref(p))
    if rc != 0:
        raise ValueError(strerror(rc))
    return c.value, p.value, l.value, f.value, m.value


def debug_plan64(ratio_num, ratio_den, quality, channels):
    """the round-4 plans: fp64-accumulate kernels (fast_path 5 / 4) or phase pairs for mono (6); host-only"""
    out = (C.c_uint32 * 8)()
    rc = lib().speexhip_debug_plan64(ratio_num, ratio_den, quality, channels, out)
    if rc:
        raise ValueError(strerror(rc))
    v = list(out)
    return {"fast_path": v[0], "r_or_p": v[1], "lane_periods": v[2], "row_len": v[3], "lds_bytes": v[4],
            "pad_or_stride": v[5], "trips": v[6], "last": v[7]}


def debug_launch_shape(ratio_num, ratio_den, quality, channels, streams, frames, float_io=False):
    """host-only: the period kernel's launch for a first call of `frames` frames on each of `streams` streams"""
    out = (C.c_uint32 * 10)()
    rc = lib().speexhip_debug_launch_shape(ratio_num, ratio_den, quality, channels, streams, frames, int(float_io), out)
    if rc:
        raise ValueError(strerror(rc))
    v = list(out)
    return {"phase_pairs": bool(v[0]), "r": v[1], "int16_window": bool(v[2]), "lane_periods": v[3], "tiles": v[4],
            "splits": v[5], "wave_groups": v[6], "shares": v[7], "threads": v[8], "touch": bool(v[9])}


def device_count():
    """logical devices the library can place states on (speexhip_device_count)"""
    return lib().speexhip_device_count()


def placement(device_count_, env_device, env_devices, k, current=0):
    """host-only: the placement rule (speexhip_debug_placement); None = unset environment variable"""
    enc = lambda v: None if v is None else str(v).encode()
    return lib().speexhip_debug_placement(device_count_, enc(env_device), enc(env_devices), k, current)


def placement_live(device_count_, env_device, env_devices, k, current, live):
    """host-only: the placement rule with live state counts per device (speexhip_debug_placement_live)"""
    enc = lambda v: None if v is None else str(v).encode()
    arr = (C.c_uint32 * max(len(live), 1))(*live)
    return lib().speexhip_debug_placement_live(device_count_, enc(env_device), enc(env_devices), k, current, arr)


def process_many(states, chunks, capacities, dtype=np.int16):
    """speexhip_resampler_process_many_int / _float: chunks[i] (frames x channels, or None with capacities[i] =
    (null_frames, capacity)) through states[i], all in one call.  Returns (outputs, consumed, codes)."""
    n = len(states)
    hs, ins, outs = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    il, ol, codes = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_int * n)()
    keep, bufs = [], []
    for i, (st, ch_) in enumerate(zip(states, chunks)):
        hs[i] = st._h
        if ch_ is None:
            ins[i], il[i], ol[i] = None, capacities[i][0], capacities[i][1]
        else:
            a = np.ascontiguousarray(ch_, dtype=dtype).reshape(-1, st.channels)
            keep.append(a)
            ins[i], il[i], ol[i] = a.ctypes.data, a.shape[0], capacities[i]
        b = np.zeros((max(int(ol[i]), 1), st.channels), dtype)
        bufs.append(b)
        outs[i] = b.ctypes.data
    fn = lib().speexhip_resampler_process_many_int if dtype == np.int16 else lib().speexhip_resampler_process_many_float
    rc = fn(n, hs, ins, il, outs, ol, codes)
    if rc not in (0, ERR_ALLOC_FAILED):
        raise RuntimeError(strerror(rc))
    return [bufs[i][: ol[i]].copy() for i in range(n)], list(il), list(codes)


class PinnedBlock:
    """A pinned block of the library the caller fills (speexhip_block_acquire, round 6): `.array(dtype, shape)` is a
    numpy view of it; the host-buffer calls use such memory in place -- the kernel reads it through PCIe.  Raises
    MemoryError when no block is free.  close() (or the context manager) gives it back."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().speexhip_block_acquire(self.nbytes)
        if not self.ptr:
            raise MemoryError("speexhip_block_acquire(%d): no pinned block free" % self.nbytes)

    def array(self, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert n <= self.nbytes
        buf = (C.c_char * n).from_address(self.ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def close(self):
        if self.ptr:
            lib().speexhip_block_release(C.c_void_p(self.ptr))
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def pcie_peak(nbytes, reps=6):
    """(h2d, d2h, each way with both at once) GB/s of plain pinned copies of nbytes: speexhip_debug_pcie_peak"""
    out = (C.c_double * 3)()
    rc = lib().speexhip_debug_pcie_peak(int(nbytes), int(reps), out)
    if rc:
        raise RuntimeError(strerror(rc))
    return tuple(out)


def device_clock():
    """(median GHz, slowest workgroup's GHz) the chip holds under an FIR-like load: the kind of box this is"""
    a, b = C.c_double(), C.c_double()
    rc = lib().speexhip_debug_device_clock(C.byref(a), C.byref(b))
    if rc:
        raise RuntimeError(strerror(rc))
    return a.value, b.value


def debug_plan(ratio_num, ratio_den, quality, channels):
    """host-only: which fast kernel this configuration gets and its geometry (speexhip_debug_plan)"""
    out = (C.c_uint32 * 8)()
    rc = lib().speexhip_debug_plan(ratio_num, ratio_den, quality, channels, out)
    if rc:
        raise ValueError(strerror(rc))
    v = list(out)
    return {"fast_path": v[0], "r_or_p": v[1], "lane_periods": v[2], "row_len": v[3], "lds_bytes": v[4],
            "pad": v[5], "fine_plan": bool(v[6]),
            # slide kernel: tap steps per loop iteration (out[7] means something else for period plans)
            "steps_per_iteration": v[7] if v[0] != 2 else 0,
            # period kernel: periods per tile of the int16-window plan that int16 calls take (0 = none)
            "w16_lane_periods": v[7] if v[0] == 2 else 0}


def plan_filter_change(old_taps, new_taps, magic, phase=None, old_den=1, new_den=1):
    """Host-only: (rc, shift, new_magic, last_delta, phase')."""
    sh, nm, ld = C.c_int64(), C.c_uint32(), C.c_int32()
    ph = C.c_uint32(phase or 0)
    rc = lib().speexhip_plan_filter_change(old_taps, new_taps, magic, C.byref(sh), C.byref(nm), C.byref(ld),
                                           C.byref(ph) if phase is not None else None, old_den, new_den)
    return rc, sh.value, nm.value, ld.value, ph.value


class Resampler:
    """Thin object over the C ABI state: host-buffer ``process`` and device-pointer
    ``process_device`` (same signature as oracle.Oracle.process for the shared test driver)."""

    def __init__(self, channels, in_rate, out_rate, quality=7, mode=None, ratio=None, device=None):
        err = C.c_int(0)
        if device is not None:
            self._h = lib().speexhip_resampler_init_on(device, channels, in_rate, out_rate, quality, C.byref(err))
        elif ratio is None:
            self._h = lib().speexhip_resampler_init(channels, in_rate, out_rate, quality, C.byref(err))
        else:
            self._h = lib().speexhip_resampler_init_frac(channels, ratio[0], ratio[1], in_rate, out_rate,
                                                         quality, C.byref(err))
        if not self._h:
            raise (RuntimeError if err.value == ERR_DEVICE else ValueError)(strerror(err.value))
        self.channels = channels
        if mode is not None:
            self.set_mode(mode)
        self.refresh()

    def refresh(self):
        i = self.info()
        self.num, self.den, self.taps = i["num_rate"], i["den_rate"], i["filt_len"]
        self.oversample, self.kind = i["oversample"], KERNEL_NAMES[i["kernel"]]
        self.table_len = i["sinc_table_length"]

    # ---- mid-stream control: same method names as oracle.Oracle / oracle.Reference ----
    def set_rate(self, in_rate, out_rate):
        rc = lib().speexhip_resampler_set_rate(self._h, in_rate, out_rate)
        self.refresh()
        return rc

    def set_rate_frac(self, num, den, in_rate, out_rate):
        rc = lib().speexhip_resampler_set_rate_frac(self._h, num, den, in_rate, out_rate)
        self.refresh()
        return rc

    def set_quality(self, quality):
        rc = lib().speexhip_resampler_set_quality(self._h, quality)
        self.refresh()
        return rc

    def rate(self):
        a, b = C.c_uint32(), C.c_uint32()
        lib().speexhip_resampler_get_rate(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def ratio(self):
        a, b = C.c_uint32(), C.c_uint32()
        lib().speexhip_resampler_get_ratio(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def quality(self):
        q = C.c_int()
        lib().speexhip_resampler_get_quality(self._h, C.byref(q))
        return q.value

    def input_latency(self):
        return lib().speexhip_resampler_get_input_latency(self._h)

    def output_latency(self):
        return lib().speexhip_resampler_get_output_latency(self._h)

    def skip_zeros(self):
        return lib().speexhip_resampler_skip_zeros(self._h)

    def reset_mem(self):
        return lib().speexhip_resampler_reset_mem(self._h)

    def pending(self, c=0):
        """channel c of the pending ("magic") frames"""
        return self._lines()[self.taps - 1:, c].copy()

    def _lines(self):
        n = self.taps - 1 + self.info()["magic_samples"]
        buf = np.zeros((max(n, 1), self.channels), np.float32)
        rc = lib().speexhip_resampler_get_history(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)))
        if rc:
            raise RuntimeError(strerror(rc))
        return buf[:n]

    def set_mode(self, mode):
        rc = lib().speexhip_resampler_set_mode(self._h, mode)
        if rc:
            raise ValueError(strerror(rc))

    def process_take(self, frames, out_capacity, float_io=False, keep=False):
        """The host-buffer call whose result stays in a pinned block of the library (what the N-API addon wraps in
        an external Buffer).  Returns (samples, consumed): a copy with the block released, or -- keep=True -- a
        view over the block plus the block's address as a third item (release with release_block)."""
        frames = np.ascontiguousarray(frames, dtype=np.float32 if float_io else np.int16)
        n = frames.shape[0]
        il, ol = C.c_uint32(n), C.c_uint32(out_capacity)
        ctype = C.c_float if float_io else C.c_int16
        blk = C.POINTER(ctype)()
        fn = (lib().speexhip_resampler_process_interleaved_float_take if float_io
              else lib().speexhip_resampler_process_interleaved_int_take)
        rc = fn(self._h, frames.ctypes.data_as(C.c_void_p) if n else None, C.byref(il), C.byref(ol), C.byref(blk))
        if rc == ERR_NO_BLOCK:
            raise MemoryError(strerror(rc))
        if rc:
            raise RuntimeError(strerror(rc))
        if not blk:
            empty = np.zeros((0, self.channels), dtype=frames.dtype)
            return (empty, il.value, 0) if keep else (empty, il.value)
        view = np.ctypeslib.as_array(blk, shape=(ol.value * self.channels,)).reshape(ol.value, self.channels)
        if keep:
            return view, il.value, C.cast(blk, C.c_void_p).value
        out = view.copy()
        lib().speexhip_block_release(C.cast(blk, C.c_void_p))
        return out, il.value

    @staticmethod
    def release_block(addr):
        lib().speexhip_block_release(C.c_void_p(addr))

    def release_stream(self):
        """before the caller destroys the stream of this state's last device-pointer call"""
        rc = lib().speexhip_resampler_release_stream(self._h)
        if rc:
            raise RuntimeError(strerror(rc))

    def info(self):
        i = Info()
        lib().speexhip_resampler_get_info(self._h, C.byref(i))
        return i.as_dict()

    def position(self):
        i = self.info()
        return i["last_sample"], i["samp_frac_num"]

    def history(self):
        """(taps-1, channels) float32: the reference's `mem` after the last call"""
        return self._lines()[: self.taps - 1].copy()

    # ---- raw calls: same names and results as oracle._RawMixin (return code + whole buffer) ----
    SENTINEL_I16, SENTINEL_F32 = 0x5A5A, 1234.5

    def raw_call(self, kind, x, cap, null_frames=0):
        dt, cdt, fill = ((np.int16, C.c_int16, self.SENTINEL_I16) if kind == "int" else
                         (np.float32, C.c_float, self.SENTINEL_F32))
        if x is None:
            ptr, n = None, int(null_frames)
        else:
            x = np.ascontiguousarray(x, dtype=dt).reshape(-1, self.channels)
            ptr, n = x.ctypes.data_as(C.POINTER(cdt)), x.shape[0]
        out = np.full((max(int(cap), 1), self.channels), fill, dt)
        il, ol = C.c_uint32(n), C.c_uint32(int(cap))
        fn = (lib().speexhip_resampler_process_interleaved_int if kind == "int"
              else lib().speexhip_resampler_process_interleaved_float)
        rc = fn(self._h, ptr, C.byref(il), out.ctypes.data_as(C.POINTER(cdt)), C.byref(ol))
        return rc, il.value, ol.value, out

    def channel_call(self, kind, c, x, cap, in_stride=1, out_stride=1, null_frames=0):
        dt, cdt, fill = ((np.int16, C.c_int16, self.SENTINEL_I16) if kind == "int" else
                         (np.float32, C.c_float, self.SENTINEL_F32))
        lib().speexhip_resampler_set_input_stride(self._h, in_stride)
        lib().speexhip_resampler_set_output_stride(self._h, out_stride)
        if x is None:
            ptr, n = None, int(null_frames)
        else:
            x = np.asarray(x, dtype=dt).reshape(-1)
            n = x.shape[0]
            buf = np.full(max((n - 1) * in_stride + 1, 1), fill, dt)
            buf[: (n - 1) * in_stride + 1: in_stride] = x
            ptr = buf.ctypes.data_as(C.POINTER(cdt))
        out = np.full(max((int(cap) - 1) * out_stride + 1, 1), fill, dt)
        il, ol = C.c_uint32(n), C.c_uint32(int(cap))
        fn = lib().speexhip_resampler_process_int if kind == "int" else lib().speexhip_resampler_process_float
        rc = fn(self._h, c, ptr, C.byref(il), out.ctypes.data_as(C.POINTER(cdt)), C.byref(ol))
        return rc, il.value, ol.value, out

    def positions(self):
        res = []
        for c in range(self.channels):
            a, b, m = C.c_int32(), C.c_uint32(), C.c_uint32()
            lib().speexhip_resampler_get_channel_position(self._h, c, C.byref(a), C.byref(b), C.byref(m))
            res.append((a.value, b.value, m.value))
        return res

    def process(self, frames, out_capacity, null_frames=0):
        """frames=None: the reference's in == NULL case (null_frames frames of silence)."""
        if frames is None:
            ptr, n = None, int(null_frames)
        else:
            frames = np.ascontiguousarray(frames, dtype=np.int16)
            if frames.ndim == 1:
                frames = frames.reshape(-1, self.channels)
            ptr, n = frames.ctypes.data_as(C.POINTER(C.c_int16)), frames.shape[0]
        out = np.zeros((max(int(out_capacity), 1), self.channels), np.int16)
        il, ol = C.c_uint32(n), C.c_uint32(int(out_capacity))
        rc = lib().speexhip_resampler_process_interleaved_int(
            self._h, ptr, C.byref(il),
            out.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(ol))
        if rc:
            raise RuntimeError(strerror(rc))
        return out[: ol.value].copy(), il.value

    def process_into(self, x, out, float_io=False):
        """The C call itself on the caller's own buffers, no copy on either side: x (frames x channels) in, `out`
        (capacity x channels) written in place -- either may be a view of a PinnedBlock, which the library then uses
        where it lies (round 6).  Returns (consumed, produced)."""
        assert x.flags["C_CONTIGUOUS"] and out.flags["C_CONTIGUOUS"] and x.dtype == out.dtype
        il, ol = C.c_uint32(x.shape[0]), C.c_uint32(out.shape[0])
        if float_io:
            rc = lib().speexhip_resampler_process_interleaved_float(
                self._h, C.cast(x.ctypes.data, C.POINTER(C.c_float)), C.byref(il),
                C.cast(out.ctypes.data, C.POINTER(C.c_float)), C.byref(ol))
        else:
            rc = lib().speexhip_resampler_process_interleaved_int(
                self._h, C.cast(x.ctypes.data, C.POINTER(C.c_int16)), C.byref(il),
                C.cast(out.ctypes.data, C.POINTER(C.c_int16)), C.byref(ol))
        if rc:
            raise RuntimeError(strerror(rc))
        return il.value, ol.value

    def process_float(self, frames, out_capacity, null_frames=0):
        """speexhip_resampler_process_interleaved_float with host buffers (float32 in / out)."""
        if frames is None:
            ptr, n = None, int(null_frames)
        else:
            frames = np.ascontiguousarray(frames, dtype=np.float32)
            if frames.ndim == 1:
                frames = frames.reshape(-1, self.channels)
            ptr, n = frames.ctypes.data_as(C.POINTER(C.c_float)), frames.shape[0]
        out = np.zeros((max(int(out_capacity), 1), self.channels), np.float32)
        il, ol = C.c_uint32(n), C.c_uint32(int(out_capacity))
        rc = lib().speexhip_resampler_process_interleaved_float(
            self._h, ptr, C.byref(il),
            out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ol))
        if rc:
            raise RuntimeError(strerror(rc))
        return out[: ol.value].copy(), il.value

    def peek(self, in_frames, out_capacity, float_entry=False):
        """(consumed, produced) of the next call, state untouched"""
        c, p_ = C.c_uint32(), C.c_uint32()
        rc = lib().speexhip_resampler_peek(self._h, in_frames, out_capacity, int(float_entry), C.byref(c), C.byref(p_))
        if rc:
            raise RuntimeError(strerror(rc))
        return c.value, p_.value

    def process_chunks(self, chunks, capacities, dtype=np.int16):
        """n consecutive calls as one launch (speexhip_resampler_process_chunks_int / _float).
        chunks: arrays of frames (or None with capacities[i] = (null_frames, capacity)).
        Returns (list of per-call outputs, list of frames consumed)."""
        n = len(chunks)
        keep, ptrs, lens, caps = [], (C.c_void_p * n)(), (C.c_uint32 * n)(), (C.c_uint32 * n)()
        for i, ch_ in enumerate(chunks):
            if ch_ is None:
                ptrs[i], lens[i], caps[i] = None, capacities[i][0], capacities[i][1]
            else:
                a = np.ascontiguousarray(ch_, dtype=dtype).reshape(-1, self.channels)
                keep.append(a)
                ptrs[i], lens[i], caps[i] = a.ctypes.data, a.shape[0], capacities[i]
        out = np.zeros((max(sum(caps), 1), self.channels), dtype)
        fn = (lib().speexhip_resampler_process_chunks_int if dtype == np.int16
              else lib().speexhip_resampler_process_chunks_float)
        rc = fn(self._h, n, ptrs, lens, C.c_void_p(out.ctypes.data), caps)
        if rc:
            raise RuntimeError(strerror(rc))
        outs, off = [], 0
        for i in range(n):
            outs.append(out[off: off + caps[i]].copy())
            off += caps[i]
        return outs, list(lens)

    def process_device(self, d_in_ptr, in_frames, d_out_ptr, out_capacity, stream_ptr=0, float_io=False):
        il, ol = C.c_uint32(in_frames), C.c_uint32(out_capacity)
        fn = (lib().speexhip_resampler_process_interleaved_float_device if float_io
              else lib().speexhip_resampler_process_interleaved_int_device)
        rc = fn(
            self._h, C.c_void_p(d_in_ptr), C.byref(il), C.c_void_p(d_out_ptr), C.byref(ol),
            C.c_void_p(stream_ptr))
        if rc:
            raise RuntimeError(strerror(rc))
        return il.value, ol.value

    def close(self):
        if getattr(self, "_h", None):
            lib().speexhip_resampler_destroy(self._h)
            self._h = None

    __del__ = close


class Batch:
    """n_streams independent streams with one shared filter; device pointers, one launch/call."""

    def __init__(self, n_streams, channels, in_rate, out_rate, quality=7, mode=None):
        err = C.c_int(0)
        self._h = lib().speexhip_batch_init(n_streams, channels, in_rate, out_rate, quality,
                                            C.byref(err))
        if not self._h:
            raise (RuntimeError if err.value == ERR_DEVICE else ValueError)(strerror(err.value))
        self.n_streams, self.channels = n_streams, channels
        if mode is not None:
            rc = lib().speexhip_batch_set_mode(self._h, mode)
            if rc:
                raise ValueError(strerror(rc))

    def info(self, stream=0):
        i = Info()
        lib().speexhip_batch_get_info(self._h, stream, C.byref(i))
        return i.as_dict()

    def set_rate_frac(self, num, den, in_rate, out_rate):
        return lib().speexhip_batch_set_rate_frac(self._h, num, den, in_rate, out_rate)

    def set_quality(self, quality):
        return lib().speexhip_batch_set_quality(self._h, quality)

    def skip_zeros(self):
        return lib().speexhip_batch_skip_zeros(self._h)

    def reset_mem(self):
        return lib().speexhip_batch_reset_mem(self._h)

    def lines(self, stream):
        """(taps-1+pending, channels) float32 of one stream: history then pending frames"""
        i = self.info(stream)
        n = i["filt_len"] - 1 + i["magic_samples"]
        buf = np.zeros((max(n, 1), self.channels), np.float32)
        rc = lib().speexhip_batch_get_history(self._h, stream, buf.ctypes.data_as(C.POINTER(C.c_float)))
        if rc:
            raise RuntimeError(strerror(rc))
        return buf[:n]

    def process_device(self, d_in_ptr, in_stride, in_frames, d_out_ptr, out_stride, out_capacity,
                       stream_ptr=0, float_io=False):
        """in_frames / out_capacity: int (same for all streams) or sequences of n_streams.
        float_io: the buffers hold float32 samples (strides in samples either way)."""
        n = self.n_streams
        il = (C.c_uint32 * n)(*([in_frames] * n if np.isscalar(in_frames) else in_frames))
        ol = (C.c_uint32 * n)(*([out_capacity] * n if np.isscalar(out_capacity) else out_capacity))
        fn = (lib().speexhip_batch_process_interleaved_float_device if float_io
              else lib().speexhip_batch_process_interleaved_int_device)
        rc = fn(
            self._h, C.c_void_p(d_in_ptr), in_stride, il, C.c_void_p(d_out_ptr), out_stride, ol,
            C.c_void_p(stream_ptr))
        if rc:
            raise RuntimeError(strerror(rc))
        return list(il), list(ol)

    def close(self):
        if getattr(self, "_h", None):
            lib().speexhip_batch_destroy(self._h)
            self._h = None

    __del__ = close


class SpeexResampler:
    """Python mirror of the reference's TypeScript class (src/index.ts:21-117), over the HIP
    library: same constructor arguments, lazy init, messages, length check and -- crucially --
    the grow-only output-capacity rule (src/index.ts:80-87,95) that decides how many frames
    each processChunk call may emit (and silently drops the rest)."""

    def __init__(self, channels, inRate, outRate, quality=7):
        self.channels, self.inRate, self.outRate, self.quality = channels, inRate, outRate, quality
        self._res = None
        self._outBufferSize = -1

    def processChunk(self, chunk):
        chunk = bytes(chunk) if not isinstance(chunk, (bytes, bytearray, memoryview)) else chunk
        n = len(chunk)
        if self.channels == 0 or n % (self.channels * 2) != 0:  # JS: x % 0 is NaN !== 0
            raise ValueError("Chunk length should be a multiple of channels * 2 bytes")
        if self._res is None:
            self._res = Resampler(self.channels, self.inRate, self.outRate, self.quality)
        target = math.ceil(n * self.outRate / self.inRate)
        if self._outBufferSize < target:
            self._outBufferSize = target
        capacity = int(self._outBufferSize / self.channels / 2)
        frames = np.frombuffer(chunk, dtype=np.int16).reshape(-1, self.channels)
        out, _ = self._res.process(frames, capacity)
        return out.tobytes()
