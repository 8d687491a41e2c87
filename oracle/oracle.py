"""ctypes front-ends for the CHECKERS (test infrastructure only -- see speex_oracle.c).

* ``Oracle``     -> oracle/liboracle.so        (our CPU restatement, always available after build)
* ``Reference``  -> oracle/_ref/libspeexref.so (the reference's own C, compiled from
                    /root/reference in the dev container; prebuilt file travels to the GPU box)

Both expose the same ``process(frames_int16[F, ch], out_capacity) -> (out[n, ch], in_used)`` call,
mirroring ``speex_resampler_process_interleaved_int`` (reference deps/speex/resample.c:1061).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libspeexref.so")

KIND_NAMES = ("direct_single", "direct_double", "interpolate_single", "interpolate_double")


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libspeexref.so."""
    subprocess.run(["make", "-C", HERE] + (["-s"] if quiet else []), check=True)


def have_reference():
    return os.path.exists(REF_SO)


def lcg_pcm(n_samples, seed=12345):
    """SURVEY section 4 generator: s = s*1664525 + 1013904223 (mod 2^32); sample = int16(s >> 16),
    one draw per interleaved sample in memory order.  Vectorised with wrap-around uint32 math."""
    if n_samples == 0:
        return np.zeros(0, np.int16)
    with np.errstate(over="ignore"):
        a = np.cumprod(np.full(n_samples, 1664525, np.uint32), dtype=np.uint32)  # a^(i+1)
        geo = np.cumsum(np.concatenate(([np.uint32(1)], a[:-1])), dtype=np.uint32)  # sum a^j, j<=i
        s = a * np.uint32(seed) + np.uint32(1013904223) * geo
    return (s >> np.uint32(16)).astype(np.uint16).view(np.int16)


def tone_pcm(n_frames, channels, seed=7, amp=9000.0):
    """Low-amplitude 'music-like' deterministic signal: a few sines + small LCG dither."""
    t = np.arange(n_frames, dtype=np.float64)
    out = np.zeros((n_frames, channels), np.float64)
    rng = np.random.RandomState(seed)
    for c in range(channels):
        for _ in range(5):
            f = rng.uniform(0.001, 0.45)
            out[:, c] += rng.uniform(0.1, 1.0) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    out *= amp / np.abs(out).max()
    dither = lcg_pcm(n_frames * channels, seed).reshape(n_frames, channels).astype(np.float64) / 4096.0
    return np.clip(np.round(out + dither), -32768, 32767).astype(np.int16)


class _Base:
    def process(self, frames, out_capacity, null_frames=0):
        """frames=None: the reference's in == NULL case (null_frames frames of silence)."""
        if frames is None:
            ptr, n = None, int(null_frames)
        else:
            frames = np.ascontiguousarray(frames, dtype=np.int16)
            if frames.ndim == 1:
                frames = frames.reshape(-1, self.channels)
            assert frames.shape[1] == self.channels
            ptr, n = frames.ctypes.data_as(C.POINTER(C.c_int16)), frames.shape[0]
        out = np.zeros((max(int(out_capacity), 1), self.channels), np.int16)
        il = C.c_uint32(n)
        ol = C.c_uint32(int(out_capacity))
        rc = self._process(ptr, C.byref(il),
                           out.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(ol))
        if rc != 0:
            raise RuntimeError("process failed: %d" % rc)
        return out[: ol.value].copy(), il.value


class _FloatMixin:
    def process_float(self, frames, out_capacity, null_frames=0):
        """speex_resampler_process_interleaved_float: float32 frames in, float32 frames out."""
        if frames is None:
            ptr, n = None, int(null_frames)
        else:
            frames = np.ascontiguousarray(frames, dtype=np.float32)
            if frames.ndim == 1:
                frames = frames.reshape(-1, self.channels)
            ptr, n = frames.ctypes.data_as(C.POINTER(C.c_float)), frames.shape[0]
        out = np.zeros((max(int(out_capacity), 1), self.channels), np.float32)
        il = C.c_uint32(n)
        ol = C.c_uint32(int(out_capacity))
        rc = self._process_float(ptr, C.byref(il),
                                 out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ol))
        if rc != 0:
            raise RuntimeError("process failed: %d" % rc)
        return out[: ol.value].copy(), il.value


SENTINEL_I16 = 0x5A5A   # what the raw calls below pre-fill their output buffers with:
SENTINEL_F32 = 1234.5   # every sample a call does NOT write must still hold it afterwards


class _RawMixin:
    """Calls that return the return code and the WHOLE output buffer (pre-filled with a sentinel),
    for states whose channels stand at different positions and for the per-channel entry points
    (speex_resampler_process_int / _process_float with strides, resample.c:927-1036, 1170-1188).
    Same method names on Oracle, Reference and the HIP mirror (speexhip.Resampler)."""

    def raw_call(self, kind, x, cap, null_frames=0):
        """kind 'int' | 'float'; x: (frames, ch) array or None.  -> (rc, used, produced, out[cap, ch])"""
        dt, cdt, fill = ((np.int16, C.c_int16, SENTINEL_I16) if kind == "int" else
                         (np.float32, C.c_float, SENTINEL_F32))
        if x is None:
            ptr, n = None, int(null_frames)
        else:
            x = np.ascontiguousarray(x, dtype=dt).reshape(-1, self.channels)
            ptr, n = x.ctypes.data_as(C.POINTER(cdt)), x.shape[0]
        out = np.full((max(int(cap), 1), self.channels), fill, dt)
        il, ol = C.c_uint32(n), C.c_uint32(int(cap))
        rc = self._raw_interleaved(kind, ptr, C.byref(il), out.ctypes.data_as(C.POINTER(cdt)), C.byref(ol))
        return rc, il.value, ol.value, out

    def channel_call(self, kind, c, x, cap, in_stride=1, out_stride=1, null_frames=0):
        """One channel: x = that channel's samples (1-D) or None.  The samples are laid out
        `in_stride` apart in a sentinel-filled buffer, the output buffer holds cap samples
        `out_stride` apart.  -> (rc, used, produced, whole output buffer)"""
        dt, cdt, fill = ((np.int16, C.c_int16, SENTINEL_I16) if kind == "int" else
                         (np.float32, C.c_float, SENTINEL_F32))
        self._set_strides(in_stride, out_stride)
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
        rc = self._raw_channel(kind, c, ptr, C.byref(il), out.ctypes.data_as(C.POINTER(cdt)), C.byref(ol))
        return rc, il.value, ol.value, out

    def positions(self):
        """[(last_sample, samp_frac_num, magic_samples)] per channel"""
        return [self._chan_pos(c) for c in range(self.channels)]


class Oracle(_Base, _FloatMixin, _RawMixin):
    """Our CPU restatement."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                build()
            L = C.CDLL(ORACLE_SO)
            L.orc_new.restype = C.c_void_p
            L.orc_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
            L.orc_free.argtypes = [C.c_void_p]
            L.orc_process_interleaved_int.restype = C.c_int
            L.orc_process_interleaved_int.argtypes = [C.c_void_p, C.POINTER(C.c_int16),
                                                      C.POINTER(C.c_uint32), C.POINTER(C.c_int16),
                                                      C.POINTER(C.c_uint32)]
            L.orc_process_interleaved_float.restype = C.c_int
            L.orc_process_interleaved_float.argtypes = [C.c_void_p, C.POINTER(C.c_float),
                                                        C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                                        C.POINTER(C.c_uint32)]
            L.orc_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
            L.orc_table.restype = C.POINTER(C.c_float)
            L.orc_table.argtypes = [C.c_void_p]
            L.orc_position.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
            L.orc_history.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
            L.orc_strerror.restype = C.c_char_p
            L.orc_strerror.argtypes = [C.c_int]
            L.orc_new_frac.restype = C.c_void_p
            L.orc_new_frac.argtypes = [C.c_uint32] * 5 + [C.c_int, C.POINTER(C.c_int)]
            L.orc_set_rate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
            L.orc_set_rate_frac.argtypes = [C.c_void_p] + [C.c_uint32] * 4
            L.orc_set_quality.argtypes = [C.c_void_p, C.c_int]
            L.orc_get_rate.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            L.orc_get_ratio.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            for f in (L.orc_get_quality, L.orc_input_latency, L.orc_output_latency, L.orc_skip_zeros,
                      L.orc_reset_mem, L.orc_block_in):
                f.argtypes = [C.c_void_p]
            L.orc_pending.restype = C.c_uint32
            L.orc_pending.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
            L.orc_process_int.restype = C.c_int
            L.orc_process_int.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int16), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]
            L.orc_process_float.restype = C.c_int
            L.orc_process_float.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
            L.orc_set_input_stride.argtypes = [C.c_void_p, C.c_uint32]
            L.orc_set_output_stride.argtypes = [C.c_void_p, C.c_uint32]
            L.orc_channel_position.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32),
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            cls._lib = L
        return cls._lib

    def __init__(self, channels, in_rate, out_rate, quality=7, ratio=None):
        L = self.lib()
        err = C.c_int(0)
        if ratio is None:
            self._h = L.orc_new(channels, in_rate, out_rate, quality, C.byref(err))
        else:
            self._h = L.orc_new_frac(channels, ratio[0], ratio[1], in_rate, out_rate, quality, C.byref(err))
        self.err = err.value
        if not self._h:
            raise ValueError(L.orc_strerror(err.value).decode())
        self.channels = channels
        self.refresh()

    def refresh(self):
        info = (C.c_uint32 * 8)()
        self.lib().orc_info(self._h, info)
        (self.num, self.den, self.taps, self.oversample, kind, self.table_len,
         self.step_int, self.step_frac) = list(info)
        self.kind = KIND_NAMES[kind]

    # ---- mid-stream control (reference resample.c:1084-1220) ----
    def set_rate(self, in_rate, out_rate):
        rc = self.lib().orc_set_rate(self._h, in_rate, out_rate)
        self.refresh()
        return rc

    def set_rate_frac(self, num, den, in_rate, out_rate):
        rc = self.lib().orc_set_rate_frac(self._h, num, den, in_rate, out_rate)
        self.refresh()
        return rc

    def set_quality(self, quality):
        rc = self.lib().orc_set_quality(self._h, quality)
        self.refresh()
        return rc

    def rate(self):
        a, b = C.c_uint32(), C.c_uint32()
        self.lib().orc_get_rate(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def ratio(self):
        a, b = C.c_uint32(), C.c_uint32()
        self.lib().orc_get_ratio(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def quality(self):
        return self.lib().orc_get_quality(self._h)

    def input_latency(self):
        return self.lib().orc_input_latency(self._h)

    def output_latency(self):
        return self.lib().orc_output_latency(self._h)

    def skip_zeros(self):
        return self.lib().orc_skip_zeros(self._h)

    def reset_mem(self):
        return self.lib().orc_reset_mem(self._h)

    def pending(self, c=0):
        n = self.lib().orc_pending(self._h, c, None)
        buf = np.zeros(max(n, 1), np.float32)
        self.lib().orc_pending(self._h, c, buf.ctypes.data_as(C.POINTER(C.c_float)))
        return buf[:n]

    def block_in(self):
        return self.lib().orc_block_in(self._h)

    def _process(self, i, il, o, ol):
        return self.lib().orc_process_interleaved_int(self._h, i, il, o, ol)

    def _process_float(self, i, il, o, ol):
        return self.lib().orc_process_interleaved_float(self._h, i, il, o, ol)

    def _raw_interleaved(self, kind, i, il, o, ol):
        return (self._process if kind == "int" else self._process_float)(i, il, o, ol)

    def _raw_channel(self, kind, c, i, il, o, ol):
        fn = self.lib().orc_process_int if kind == "int" else self.lib().orc_process_float
        return fn(self._h, c, i, il, o, ol)

    def _set_strides(self, a, b):
        self.lib().orc_set_input_stride(self._h, a)
        self.lib().orc_set_output_stride(self._h, b)

    def _chan_pos(self, c):
        a, b, m = C.c_int32(), C.c_uint32(), C.c_uint32()
        self.lib().orc_channel_position(self._h, c, C.byref(a), C.byref(b), C.byref(m))
        return a.value, b.value, m.value

    def table(self):
        p = self.lib().orc_table(self._h)
        return np.ctypeslib.as_array(p, shape=(self.table_len,)).copy()

    def position(self):
        a, b = C.c_int32(), C.c_uint32()
        self.lib().orc_position(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def history(self, c=0):
        buf = np.zeros(self.taps - 1, np.float32)
        self.lib().orc_history(self._h, c, buf.ctypes.data_as(C.POINTER(C.c_float)))
        return buf

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib().orc_free(self._h)
            self._h = None


class _RefState(C.Structure):
    """Field-for-field mirror of the reference's private state struct
    (deps/speex/resample.c:116-146) so tests can read filt_len / sinc_table / counters."""
    _fields_ = [
        ("in_rate", C.c_uint32), ("out_rate", C.c_uint32), ("num_rate", C.c_uint32),
        ("den_rate", C.c_uint32), ("quality", C.c_int), ("nb_channels", C.c_uint32),
        ("filt_len", C.c_uint32), ("mem_alloc_size", C.c_uint32), ("buffer_size", C.c_uint32),
        ("int_advance", C.c_int), ("frac_advance", C.c_int), ("cutoff", C.c_float),
        ("oversample", C.c_uint32), ("initialised", C.c_int), ("started", C.c_int),
        ("last_sample", C.POINTER(C.c_int32)), ("samp_frac_num", C.POINTER(C.c_uint32)),
        ("magic_samples", C.POINTER(C.c_uint32)), ("mem", C.POINTER(C.c_float)),
        ("sinc_table", C.POINTER(C.c_float)), ("sinc_table_length", C.c_uint32),
        ("resampler_ptr", C.c_void_p), ("in_stride", C.c_int), ("out_stride", C.c_int),
    ]


class Reference(_Base, _FloatMixin, _RawMixin):
    """The reference's own C implementation (oracle/_ref/libspeexref.so)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(REF_SO)
            L.speex_resampler_init.restype = C.POINTER(_RefState)
            L.speex_resampler_init.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                               C.POINTER(C.c_int)]
            L.speex_resampler_destroy.argtypes = [C.POINTER(_RefState)]
            L.speex_resampler_process_interleaved_int.restype = C.c_int
            L.speex_resampler_process_interleaved_int.argtypes = [
                C.POINTER(_RefState), C.POINTER(C.c_int16), C.POINTER(C.c_uint32),
                C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]
            L.speex_resampler_process_interleaved_float.restype = C.c_int
            L.speex_resampler_process_interleaved_float.argtypes = [
                C.POINTER(_RefState), C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
            L.speex_resampler_strerror.restype = C.c_char_p
            L.speex_resampler_strerror.argtypes = [C.c_int]
            P = C.POINTER(_RefState)
            L.speex_resampler_init_frac.restype = P
            L.speex_resampler_init_frac.argtypes = [C.c_uint32] * 5 + [C.c_int, C.POINTER(C.c_int)]
            L.speex_resampler_set_rate.argtypes = [P, C.c_uint32, C.c_uint32]
            L.speex_resampler_set_rate_frac.argtypes = [P] + [C.c_uint32] * 4
            L.speex_resampler_set_quality.argtypes = [P, C.c_int]
            L.speex_resampler_get_rate.argtypes = [P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            L.speex_resampler_get_ratio.argtypes = [P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            L.speex_resampler_get_quality.argtypes = [P, C.POINTER(C.c_int)]
            for f in (L.speex_resampler_get_input_latency, L.speex_resampler_get_output_latency,
                      L.speex_resampler_skip_zeros, L.speex_resampler_reset_mem):
                f.argtypes = [P]
            L.speex_resampler_process_int.restype = C.c_int
            L.speex_resampler_process_int.argtypes = [P, C.c_uint32, C.POINTER(C.c_int16), C.POINTER(C.c_uint32),
                                                      C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]
            L.speex_resampler_process_float.restype = C.c_int
            L.speex_resampler_process_float.argtypes = [P, C.c_uint32, C.POINTER(C.c_float),
                                                        C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                                        C.POINTER(C.c_uint32)]
            L.speex_resampler_set_input_stride.argtypes = [P, C.c_uint32]
            L.speex_resampler_set_output_stride.argtypes = [P, C.c_uint32]
            cls._lib = L
        return cls._lib

    def __init__(self, channels, in_rate, out_rate, quality=7, ratio=None):
        L = self.lib()
        err = C.c_int(0)
        if ratio is None:
            self._h = L.speex_resampler_init(channels, in_rate, out_rate, quality, C.byref(err))
        else:
            self._h = L.speex_resampler_init_frac(channels, ratio[0], ratio[1], in_rate, out_rate, quality,
                                                  C.byref(err))
        self.err = err.value
        if not self._h:
            raise ValueError(L.speex_resampler_strerror(err.value).decode())
        self.channels = channels
        self.refresh()

    def refresh(self):
        st = self._h.contents
        self.num, self.den = st.num_rate, st.den_rate
        self.taps, self.oversample = st.filt_len, st.oversample
        self.step_int, self.step_frac = st.int_advance, st.frac_advance
        direct = st.filt_len * st.den_rate <= st.filt_len * st.oversample + 8
        self.kind = KIND_NAMES[(0 if direct else 2) + (1 if st.quality > 8 else 0)]
        self.table_len = st.filt_len * st.den_rate if direct else st.filt_len * st.oversample + 8

    # ---- mid-stream control (deps/speex/resample.c:1084-1220) ----
    def set_rate(self, in_rate, out_rate):
        rc = self.lib().speex_resampler_set_rate(self._h, in_rate, out_rate)
        self.refresh()
        return rc

    def set_rate_frac(self, num, den, in_rate, out_rate):
        rc = self.lib().speex_resampler_set_rate_frac(self._h, num, den, in_rate, out_rate)
        self.refresh()
        return rc

    def set_quality(self, quality):
        rc = self.lib().speex_resampler_set_quality(self._h, quality)
        self.refresh()
        return rc

    def rate(self):
        a, b = C.c_uint32(), C.c_uint32()
        self.lib().speex_resampler_get_rate(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def ratio(self):
        a, b = C.c_uint32(), C.c_uint32()
        self.lib().speex_resampler_get_ratio(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def quality(self):
        q = C.c_int()
        self.lib().speex_resampler_get_quality(self._h, C.byref(q))
        return q.value

    def input_latency(self):
        return self.lib().speex_resampler_get_input_latency(self._h)

    def output_latency(self):
        return self.lib().speex_resampler_get_output_latency(self._h)

    def skip_zeros(self):
        return self.lib().speex_resampler_skip_zeros(self._h)

    def reset_mem(self):
        return self.lib().speex_resampler_reset_mem(self._h)

    def pending(self, c=0):
        st = self._h.contents
        base = c * st.mem_alloc_size + st.filt_len - 1
        return np.array([st.mem[base + j] for j in range(st.magic_samples[c])], np.float32)

    def block_in(self):
        st = self._h.contents
        return st.mem_alloc_size - (st.filt_len - 1)

    def _process(self, i, il, o, ol):
        return self.lib().speex_resampler_process_interleaved_int(self._h, i, il, o, ol)

    def _process_float(self, i, il, o, ol):
        return self.lib().speex_resampler_process_interleaved_float(self._h, i, il, o, ol)

    def _raw_interleaved(self, kind, i, il, o, ol):
        return (self._process if kind == "int" else self._process_float)(i, il, o, ol)

    def _raw_channel(self, kind, c, i, il, o, ol):
        fn = (self.lib().speex_resampler_process_int if kind == "int"
              else self.lib().speex_resampler_process_float)
        return fn(self._h, c, i, il, o, ol)

    def _set_strides(self, a, b):
        self.lib().speex_resampler_set_input_stride(self._h, a)
        self.lib().speex_resampler_set_output_stride(self._h, b)

    def _chan_pos(self, c):
        st = self._h.contents
        return int(st.last_sample[c]), int(st.samp_frac_num[c]), int(st.magic_samples[c])

    def table(self):
        return np.ctypeslib.as_array(self._h.contents.sinc_table, shape=(self.table_len,)).copy()

    def position(self):
        st = self._h.contents
        return st.last_sample[0], st.samp_frac_num[0]

    def history(self, c=0):
        st = self._h.contents
        base = c * st.mem_alloc_size
        return np.array([st.mem[base + j] for j in range(st.filt_len - 1)], np.float32)

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib().speex_resampler_destroy(self._h)
            self._h = None


def wrapper_capacity(chunk_bytes, in_rate, out_rate, channels, prev_out_buffer_size=-1):
    """The JS wrapper's output-capacity rule (reference src/index.ts:80-87,95): grow-only
    byte size ceil(len*out/in), then frames = trunc(size / channels / 2).  Returns
    (capacity_frames, new_out_buffer_size)."""
    import math
    target = math.ceil(chunk_bytes * out_rate / in_rate)
    size = max(prev_out_buffer_size, target)
    return int(size / channels / 2), size
