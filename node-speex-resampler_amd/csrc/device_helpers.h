// device_helpers.h -- device-side pieces shared by the kernels: typed access to the stream
// buffers in HBM, the history roll, staging of an input window into LDS as float (one-shot or
// split into fetch / commit for software pipelining) and rounding to PCM.  Templated on the
// sample type T of the call: int16_t (speex_resampler_process_interleaved_int, reference
// deps/speex/resample.c:1061) or float (speex_resampler_process_interleaved_float, :1038).
// The history is always float, like the reference's `mem` (resample.c:139), so int and float
// calls can be mixed on one stream.  Included only from .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace speexhip {
namespace {

// Stream buffers are HBM: name the global address space so that loads/stores compile to
// global_* (flat_* also counts against lgkmcnt and would stall the FIR loop's scalar-load waits).
template <typename T>
using G = __attribute__((address_space(1))) T;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native vector (HIP's uint4 is a struct)
typedef G<const u32x4> g_cuint4;
typedef G<uint32_t> g_u32;
typedef G<int16_t> g_i16;
typedef G<float> g_f32;
template <typename T>
__device__ __forceinline__ G<const T> *in_ptr(const StreamDesc &d) {
  return (G<const T> *)d.in;
}
template <typename T>
__device__ __forceinline__ G<T> *out_ptr(const StreamDesc &d) {
  return (G<T> *)d.out;
}
__device__ __forceinline__ G<const float> *hist_ptr(const StreamDesc &d) { return (G<const float> *)d.hist; }

// samples per 16-byte load
template <typename T>
struct PerLoad {
  static constexpr int value = 16 / sizeof(T);
};

// The next call's history: hist_keep frames of (history ++ input) starting `consumed` frames in,
// i.e. reference resample.c:898-899 (and :914-919 for pending frames) applied once over the
// whole call.
template <typename T>
__device__ __forceinline__ void roll_history(uint32_t channels, const StreamDesc &d, uint32_t nthr,
                                             uint32_t in_stride = 0, uint32_t hist_stride = 0) {
  if (in_stride == 0) in_stride = channels;      // interleaved frames
  if (hist_stride == 0) hist_stride = channels;
  const uint32_t hist_frames = d.hist_frames;
  const uint32_t total = d.hist_keep * channels;
  for (uint32_t i = threadIdx.x; i < total; i += nthr) {
    const uint32_t h = i / channels, c = i - h * channels;
    const int64_t v = static_cast<int64_t>(d.consumed) + h;
    float s;
    if (v < static_cast<int64_t>(hist_frames)) {
      s = hist_ptr(d)[v * hist_stride + c];
    } else {
      const int64_t f = v - hist_frames;
      s = (d.in != nullptr && f < static_cast<int64_t>(d.in_frames))
              ? static_cast<float>(in_ptr<T>(d)[f * in_stride + c])
              : 0.f;
    }
    ((g_f32 *)d.hist_next)[static_cast<size_t>(h) * hist_stride + c] = s;
  }
}

// One element of the virtual sequence in "input-relative" element units: q < 0 reaches back
// into the history (the hist_elems floats before the input), beyond either end is silence.
template <typename T>
__device__ __forceinline__ float rel_sample(const StreamDesc &d, int64_t q, int64_t hist_elems,
                                            int64_t in_elems) {
  if (q < 0) return q >= -hist_elems ? hist_ptr(d)[q + hist_elems] : 0.f;
  return (d.in != nullptr && q < in_elems) ? static_cast<float>(in_ptr<T>(d)[q]) : 0.f;
}

// ---- staging of an input window, split in two for software pipelining --------------------------
// window_fetch() issues the 16-byte loads of a window into registers (nothing waits on them);
// window_commit() converts them and writes the LDS image later -- typically after the FIR of
// the previous tile, whose FMAs hide the HBM latency.  A "group" is one 16-byte load: 8 int16
// or 4 float samples.
struct WindowGeom {
  int64_t q_base;              // input-relative element index of LDS float 0 (multiple of a group)
  int64_t hist_elems, in_elems;
  const void *src;             // first wholly-inside group
  uint32_t total;              // floats in the LDS image (multiple of a group)
  uint32_t head_end, tail_begin;  // scalar ranges [0, head_end) U [tail_begin, total)
  uint32_t u_begin, n_wide;    // wide groups: LDS floats [G*u_begin, G*(u_begin+n_wide))
  uint32_t xshift;             // float index of the window's first frame inside the image
  uint32_t pad, period_elems;  // bank padding: `pad` floats inserted after every period_elems
                               // floats (a period of the period kernel, a row of the slide
                               // kernel) counted from the window's first frame
  uint32_t period_magic;       // ceil(2^32 / period_elems): n / period_elems == umulhi(n, magic) for
                               // the n < 2^17 that index an LDS image (host: period_magic_of)
  uint32_t nthr;               // lanes of the workgroup (kernel argument, not g.nthr: see device_types.h)
};

// host + device: the multiplier above (exact for n * (magic*d - 2^32) < 2^32, i.e. n < 2^32 / d)
__host__ __device__ inline uint32_t period_magic_of(uint32_t d) {
  return d <= 1 ? 0u : static_cast<uint32_t>(((1ull << 32) + d - 1) / d);
}

template <typename T>
__device__ __forceinline__ WindowGeom window_geom(const StreamDesc &d, uint32_t channels,
                                                  uint32_t num, uint32_t tail_frames, uint32_t m_lo,
                                                  uint32_t m_cnt, uint32_t nthr, uint32_t pad = 0,
                                                  uint32_t period_magic = 0, uint32_t pad_every = 0) {
  constexpr int GS = PerLoad<T>::value;
  WindowGeom w;
  w.nthr = nthr;
  w.pad = pad;
  w.period_elems = pad_every ? pad_every : num * channels;  // floats between two paddings
  w.period_magic = period_magic;
  w.hist_elems = static_cast<int64_t>(d.hist_frames) * channels;
  w.in_elems = static_cast<int64_t>(d.in_frames) * channels;
  const int64_t q_lo =
      (static_cast<int64_t>(d.base_shift) + static_cast<int64_t>(m_lo) * num) * channels - w.hist_elems;
  w.q_base = (q_lo >= 0 ? q_lo / GS : -((-q_lo + GS - 1) / GS)) * GS;
  w.xshift = static_cast<uint32_t>(q_lo - w.q_base);
  const uint32_t span = (m_cnt - 1) * num + tail_frames;
  w.total = (w.xshift + span * channels + GS - 1) / GS * GS;
  w.head_end = w.total;
  w.tail_begin = w.total;
  w.u_begin = 0;
  w.n_wide = 0;
  w.src = d.in;
  const bool wide_ok = d.in != nullptr && w.in_elems >= GS && (reinterpret_cast<uintptr_t>(d.in) & 15u) == 0;
  if (wide_ok) {
    const int64_t q_max = w.in_elems - GS;  // last group start that is wholly inside
    const int64_t first = w.q_base < 0 ? -w.q_base : 0;
    const int64_t beyond = (q_max / GS) * GS + GS - w.q_base;
    w.head_end = static_cast<uint32_t>(min(first, static_cast<int64_t>(w.total)));
    w.tail_begin = static_cast<uint32_t>(
        min(max(beyond, static_cast<int64_t>(w.head_end)), static_cast<int64_t>(w.total)));
    w.u_begin = w.head_end / GS;
    w.n_wide = w.tail_begin / GS - w.u_begin;
    w.src = static_cast<const T *>(d.in) + (w.q_base + GS * static_cast<int64_t>(w.u_begin));
  }
  return w;
}

// The same for a window that lies wholly inside the call's input, 16-byte aligned -- every tile of a call but
// its first and its last few: false (and *w untouched) when the window is not of that kind.
template <typename T>
__device__ __forceinline__ bool window_geom_plain(const StreamDesc &d, uint32_t channels, uint32_t num,
                                                  uint32_t tail_frames, uint32_t m_lo, uint32_t m_cnt, uint32_t nthr,
                                                  uint32_t pad, uint32_t period_magic, WindowGeom *w,
                                                  uint32_t pad_every = 0) {
  constexpr int GS = PerLoad<T>::value;
  const int64_t hist_elems = static_cast<int64_t>(d.hist_frames) * channels;
  const int64_t q_lo = (static_cast<int64_t>(d.base_shift) + static_cast<int64_t>(m_lo) * num) * channels - hist_elems;
  const uint32_t xshift = static_cast<uint32_t>(q_lo) & (GS - 1);
  const uint32_t span = (m_cnt - 1) * num + tail_frames;
  const uint32_t total = (xshift + span * channels + GS - 1) & ~static_cast<uint32_t>(GS - 1);
  const int64_t in_elems = static_cast<int64_t>(d.in_frames) * channels;
  const int64_t q_base = q_lo - xshift;
  if (d.in == nullptr || (reinterpret_cast<uintptr_t>(d.in) & 15u) != 0 || q_lo < 0 ||
      q_base + total > (in_elems & ~static_cast<int64_t>(GS - 1)))
    return false;
  w->nthr = nthr;
  w->pad = pad;
  w->period_elems = pad_every ? pad_every : num * channels;
  w->period_magic = period_magic;
  w->hist_elems = hist_elems;
  w->in_elems = in_elems;
  w->q_base = q_base;
  w->xshift = xshift;
  w->total = total;
  w->head_end = 0;
  w->tail_begin = total;
  w->u_begin = 0;
  w->n_wide = total / GS;
  w->src = static_cast<const T *>(d.in) + q_base;
  return true;
}

template <typename T>
__device__ __forceinline__ u32x4 load_group(const WindowGeom &g, uint32_t unit) {
  return *(g_cuint4 *)(static_cast<const T *>(g.src) + PerLoad<T>::value * static_cast<size_t>(unit));
}

template <int UNR, typename T>
__device__ __forceinline__ void window_fetch(const WindowGeom &g, u32x4 (&w)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t unit = u * g.nthr + threadIdx.x;
    // clamped, never branched around: all UNR loads stay in flight
    w[u] = g.n_wide ? load_group<T>(g, min(unit, g.n_wide - 1)) : u32x4{0u, 0u, 0u, 0u};
  }
}

// one 16-byte group -> floats
__device__ __forceinline__ void unpack_group(const u32x4 &w, float (&f)[8], int16_t) {
  f[0] = static_cast<float>(static_cast<int>(w.x << 16) >> 16);
  f[1] = static_cast<float>(static_cast<int>(w.x) >> 16);
  f[2] = static_cast<float>(static_cast<int>(w.y << 16) >> 16);
  f[3] = static_cast<float>(static_cast<int>(w.y) >> 16);
  f[4] = static_cast<float>(static_cast<int>(w.z << 16) >> 16);
  f[5] = static_cast<float>(static_cast<int>(w.z) >> 16);
  f[6] = static_cast<float>(static_cast<int>(w.w << 16) >> 16);
  f[7] = static_cast<float>(static_cast<int>(w.w) >> 16);
}
__device__ __forceinline__ void unpack_group(const u32x4 &w, float (&f)[4], float) {
  // (by value: __builtin_bit_cast on a vector ELEMENT reads the vector's first lane)
  const uint32_t a = w.x, b = w.y, c = w.z, e = w.w;
  f[0] = __uint_as_float(a);
  f[1] = __uint_as_float(b);
  f[2] = __uint_as_float(c);
  f[3] = __uint_as_float(e);
}

// LDS float position of image float j when the layout is padded (pad floats after every
// period_elems floats counted from the window's first frame, i.e. from image float xshift)
__device__ __forceinline__ uint32_t padded_pos(const WindowGeom &g, uint32_t j) {
  const uint32_t n = j >= g.xshift ? j - g.xshift : 0u;
  const uint32_t period = g.period_magic ? __umulhi(n, g.period_magic) : n;  // n / period_elems
  return j + period * g.pad;
}

// the floats of one group, image floats j .. j+GS-1 (j a multiple of GS): 16-byte LDS writes
// (8-byte ones where the padding leaves the group only 8-byte aligned) unless the group
// straddles a padding boundary
// (E: the element type of the LDS image -- float, or double for the fp64-accumulate slide kernel, which then reads its
//  samples already widened; positions count elements either way)
template <typename T, typename E = float>
__device__ __forceinline__ void commit_group(E *xs, const WindowGeom &g, uint32_t j, const u32x4 &w) {
  constexpr int GS = PerLoad<T>::value;
  float f[GS];
  unpack_group(w, f, T());
  uint32_t a = j;
  bool contiguous = true;
  if (g.pad != 0) {
    a = padded_pos(g, j);
    contiguous = padded_pos(g, j + GS - 1) - a == GS - 1;
  }
  if constexpr (sizeof(E) == 8) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    if (contiguous && (a & 1u) == 0) {
#pragma unroll
      for (int k = 0; k < GS; k += 2)
        *reinterpret_cast<f64x2 *>(xs + a + k) = f64x2{static_cast<double>(f[k]), static_cast<double>(f[k + 1])};
    } else {
#pragma unroll
      for (int k = 0; k < GS; k++) xs[padded_pos(g, j + k)] = static_cast<E>(f[k]);
    }
    return;
  } else if (contiguous && (a & 3u) == 0) {
#pragma unroll
    for (int k = 0; k < GS; k += 4)
      *reinterpret_cast<float4 *>(xs + a + k) = make_float4(f[k], f[k + 1], f[k + 2], f[k + 3]);
  } else if (contiguous && (a & 1u) == 0) {
#pragma unroll
    for (int k = 0; k < GS; k += 2) *reinterpret_cast<float2 *>(xs + a + k) = make_float2(f[k], f[k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < GS; k++) xs[padded_pos(g, j + k)] = f[k];
  }
}

template <int UNR, typename T, typename E = float>
__device__ __forceinline__ void window_commit(E *xs, const StreamDesc &d, const WindowGeom &g,
                                              const u32x4 (&w)[UNR]) {
  constexpr int GS = PerLoad<T>::value;
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t unit = u * g.nthr + threadIdx.x;
    if (unit < g.n_wide) commit_group<T, E>(xs, g, GS * (g.u_begin + unit), w[u]);
  }
  // groups beyond the prefetched UNR per lane (small workgroups, very wide windows): further
  // rounds of UNR loads in flight at a time
  for (uint32_t base = UNR * g.nthr; base < g.n_wide; base += UNR * g.nthr) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t unit = base + u * g.nthr + threadIdx.x;
      v[u] = load_group<T>(g, min(unit, g.n_wide - 1));
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t unit = base + u * g.nthr + threadIdx.x;
      if (unit < g.n_wide) commit_group<T, E>(xs, g, GS * (g.u_begin + unit), v[u]);
    }
  }
  for (uint32_t j = threadIdx.x; j < g.head_end; j += g.nthr)
    xs[g.pad ? padded_pos(g, j) : j] = static_cast<E>(rel_sample<T>(d, g.q_base + j, g.hist_elems, g.in_elems));
  for (uint32_t j = g.tail_begin + threadIdx.x; j < g.total; j += g.nthr)
    xs[g.pad ? padded_pos(g, j) : j] = static_cast<E>(rel_sample<T>(d, g.q_base + j, g.hist_elems, g.in_elems));
}

// ---- the same for a "plain" window: no bank padding, every 16-byte group wholly inside the call's input
// (no history in front, no silence behind: all tiles of a call but its first and last few), at most UNR
// groups per lane.  Round 3: the general path above spends ~20 vector instructions per group on 64-bit
// addresses, clamps and the three-way split of the image; here the loads are buffer loads (one 32-bit lane
// offset for all of them, the group index of each load a scalar offset; a group past the end reads as zeros
// by the descriptor's bounds check, so nothing is clamped) and the LDS address is one add per group:
// 10 vector instructions per group, 8 of them the conversions.  A vector instruction costs a wave the issue
// slot of a v_pk_fma_f32 whatever it does, and staging was 100 of the ~260 that a wave spends outside its
// FIR loop (1313 inside) on BASELINE configs[1].
template <int UNR, typename T>
__device__ __forceinline__ bool window_is_plain(const WindowGeom &g) {
  constexpr uint32_t GS = PerLoad<T>::value;
  return g.pad == 0 && g.head_end == 0 && g.tail_begin == g.total && g.n_wide * GS == g.total &&
         g.n_wide <= static_cast<uint32_t>(UNR) * g.nthr;
}

template <int UNR, typename T>
__device__ __forceinline__ void window_fetch_plain(const WindowGeom &g, u32x4 (&w)[UNR]) {
  // raw buffer descriptor over the window's groups: base, stride 0, bytes, gfx9 data format word
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.src), 0, static_cast<int>(g.n_wide * 16u), 0x00020000);
  const int lane_off = static_cast<int>(threadIdx.x * 16u);
#pragma unroll
  for (int u = 0; u < UNR; u++)
    w[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, static_cast<int>(u * g.nthr * 16u), 0);
}

template <int UNR, typename T>
__device__ __forceinline__ void window_commit_plain(float *xs, const WindowGeom &g, const u32x4 (&w)[UNR]) {
  constexpr int GS = PerLoad<T>::value;
  float *lane_xs = xs + threadIdx.x * GS;
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t first = u * g.nthr;  // scalar: group index of lane 0's load
    if (first >= g.n_wide || threadIdx.x >= g.n_wide - first) continue;
    float f[GS];
    unpack_group(w[u], f, T());
    float *dst = lane_xs + first * GS;
#pragma unroll
    for (int k = 0; k < GS; k += 4) *reinterpret_cast<float4 *>(dst + k) = make_float4(f[k], f[k + 1], f[k + 2], f[k + 3]);
  }
}

// ---- ... and for a PADDED window whose 16-byte quarters never straddle a padding boundary: frames of 4, 8, 12 ...
// channels (the window starts on a multiple of the channel count, so on a multiple of 4 elements, and a
// period is a multiple of 4 elements too).  The general path tests every group for a boundary inside it
// and falls back to eight scalar writes with a division each; here a quarter's position is its index plus
// (period it lies in) x pad, the period by one multiply-high: BASELINE configs[3] (8 channels 48k->44.1k)
// staged alone took 140 us of a 584 us launch, more than its stores.
template <int UNR, typename T>
__device__ __forceinline__ bool window_is_plain_padded(const WindowGeom &g) {
  constexpr uint32_t GS = PerLoad<T>::value;
  return g.pad != 0 && g.period_magic != 0 && g.head_end == 0 && g.tail_begin == g.total && g.n_wide * GS == g.total &&
         g.n_wide <= static_cast<uint32_t>(UNR) * g.nthr && (g.xshift & 3u) == 0 && (g.period_elems & 3u) == 0 &&
         (g.pad & 3u) == 0;
}

template <int UNR, typename T>
__device__ __forceinline__ void window_commit_plain_padded(float *xs, const WindowGeom &g, const u32x4 (&w)[UNR]) {
  constexpr int GS = PerLoad<T>::value;
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t first = u * g.nthr;
    if (first >= g.n_wide || threadIdx.x >= g.n_wide - first) continue;
    float f[GS];
    unpack_group(w[u], f, T());
    const uint32_t j = (first + threadIdx.x) * GS;
#pragma unroll
    for (int k = 0; k < GS; k += 4) {
      // (j + k >= xshift or the quarter lies before the first frame: period 0 either way)
      const uint32_t n = j + k >= g.xshift ? j + k - g.xshift : 0u;
      const uint32_t a = j + k + __umulhi(n, g.period_magic) * g.pad;
      *reinterpret_cast<float4 *>(xs + a) = make_float4(f[k], f[k + 1], f[k + 2], f[k + 3]);
    }
  }
}

// ---- an int16 LDS window (round 3, "W16"): the image holds the samples as they come from HBM, two bytes
// each, and the FIR loop converts behind its LDS reads (csrc/gen_fir_loop.py).  Half the bytes per period:
// the wide windows of down-sampling ratios (num = 320, 441, 640 input frames per period) fit twice the
// periods per tile, i.e. twice the lanes of every wave.  Only for int16 calls on a stream whose history
// holds integer-valued samples (the engine knows: no float call so far); same geometry (WindowGeom counts
// ELEMENTS, whatever their size), same three-way split, same padding rule as the float image above.
__device__ __forceinline__ int16_t pcm_of(float v) { return static_cast<int16_t>(static_cast<int>(v)); }

__device__ __forceinline__ void commit_group16(int16_t *xs, const WindowGeom &g, uint32_t j, const u32x4 &w) {
  uint32_t a = j;
  bool contiguous = true;
  if (g.pad != 0) {
    a = padded_pos(g, j);
    contiguous = padded_pos(g, j + 7) - a == 7;
  }
  if (contiguous && (a & 7u) == 0) {
    *reinterpret_cast<u32x4 *>(xs + a) = w;
  } else if (contiguous && (a & 3u) == 0) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2_t *>(xs + a) = u32x2_t{w.x, w.y};
    *reinterpret_cast<u32x2_t *>(xs + a + 4) = u32x2_t{w.z, w.w};
  } else if (contiguous && (a & 1u) == 0) {
    uint32_t *o = reinterpret_cast<uint32_t *>(xs + a);
    o[0] = w.x;
    o[1] = w.y;
    o[2] = w.z;
    o[3] = w.w;
  } else {
    const uint32_t v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 8; k++)
      xs[padded_pos(g, j + k)] = static_cast<int16_t>((k & 1) ? v[k >> 1] >> 16 : v[k >> 1] & 0xffffu);
  }
}

template <int UNR>
__device__ __forceinline__ void window_commit16(int16_t *xs, const StreamDesc &d, const WindowGeom &g,
                                                const u32x4 (&w)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t unit = u * g.nthr + threadIdx.x;
    if (unit < g.n_wide) commit_group16(xs, g, 8 * (g.u_begin + unit), w[u]);
  }
  for (uint32_t base = UNR * g.nthr; base < g.n_wide; base += UNR * g.nthr) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t unit = base + u * g.nthr + threadIdx.x;
      v[u] = load_group<int16_t>(g, min(unit, g.n_wide - 1));
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t unit = base + u * g.nthr + threadIdx.x;
      if (unit < g.n_wide) commit_group16(xs, g, 8 * (g.u_begin + unit), v[u]);
    }
  }
  for (uint32_t j = threadIdx.x; j < g.head_end; j += g.nthr)
    xs[g.pad ? padded_pos(g, j) : j] = pcm_of(rel_sample<int16_t>(d, g.q_base + j, g.hist_elems, g.in_elems));
  for (uint32_t j = g.tail_begin + threadIdx.x; j < g.total; j += g.nthr)
    xs[g.pad ? padded_pos(g, j) : j] = pcm_of(rel_sample<int16_t>(d, g.q_base + j, g.hist_elems, g.in_elems));
}

// plain window (window_is_plain<UNR, int16_t>): the 16 bytes go from the load straight into the image
template <int UNR>
__device__ __forceinline__ void window_commit_plain16(int16_t *xs, const WindowGeom &g, const u32x4 (&w)[UNR]) {
  int16_t *lane_xs = xs + threadIdx.x * 8;
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t first = u * g.nthr;
    if (first >= g.n_wide || threadIdx.x >= g.n_wide - first) continue;
    *reinterpret_cast<u32x4 *>(lane_xs + first * 8) = w[u];
  }
}

// (Round 3 also staged the windows that need no conversion -- float samples into the float window, int16 samples
//  into the int16 window -- by LDS-DMA, global_load_lds_dwordx4: no VGPRs, no ds_write, any number of rounds.
//  Same-box A/B at 32 streams: float stereo 44.1k->48k 262.0 -> 266.9 us, float mono, 48k->11.025k and
//  44.1k->16k through the int16 window within 0.5 %; in-kernel stamps: descriptor -> window complete 3.4 us with
//  registers, 4.3 us by DMA, against 1.4 us for int16 samples (half the bytes): a CU takes in ~11 bytes per
//  cycle whichever way the 76 KB arrive, so the staging instructions were never the cost; removed.)

// Two floats -> packed s16 pair {lo, hi} with the reference's rounding: floor(x + .5), then
// saturation to [-32768, 32767] (equivalent to arch.h:208-209: the < -32767.5 / > 32766.5
// branches are the clamp of floor(x + .5)).  v_cvt_rpi_i32_f32 IS floor(x + .5) ("round to plus
// infinity" of the halves): one instruction per sample instead of add + floor + convert.
// tools/check_rpi.hip compared it on gfx950 with floor(.5 + (double)x) for every multiple of
// 1/64 in [-40000, 40000] and the +-1 ulp neighbours of every half-integer: no difference.
__device__ __forceinline__ uint32_t round_pack_pcm(float lo, float hi) {
  int a, b;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(a) : "v"(lo));
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(b) : "v"(hi));
  typedef short short2_t __attribute__((ext_vector_type(2)));
  const short2_t pk = __builtin_amdgcn_cvt_pk_i16(a, b);  // saturating v_cvt_pk_i16_i32
  return __builtin_bit_cast(uint32_t, pk);
}

}  // namespace
}  // namespace speexhip
