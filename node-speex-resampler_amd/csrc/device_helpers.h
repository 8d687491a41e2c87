// device_helpers.h -- device-side pieces shared by the fast kernels (kernels_tiled.hip,
// kernels_period.hip): staging of the interleaved s16 input window into LDS as float, rounding
// to PCM, and the history roll.  Included only from .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace speexhip {
namespace {

// Stream buffers are HBM: name the global address space so that loads/stores compile to
// global_* (flat_* also counts against lgkmcnt and would stall the FIR loop's scalar-load waits).
typedef __attribute__((address_space(1))) int16_t g_i16;
typedef __attribute__((address_space(1))) const int16_t g_ci16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native vector (HIP's uint4 is a struct)
typedef __attribute__((address_space(1))) const u32x4 g_cuint4;
typedef __attribute__((address_space(1))) uint32_t g_u32;
__device__ __forceinline__ g_ci16 *as_global(const int16_t *p) { return (g_ci16 *)p; }
__device__ __forceinline__ g_i16 *as_global(int16_t *p) { return (g_i16 *)p; }

// round-half-up + saturate, identical in value to floor(.5 + (double)v) of arch.h:208-209:
// v - floorf(v) is exact in fp32, so no double arithmetic is needed.
__device__ __forceinline__ int16_t round_pcm(float v) {
  if (v < -32767.5f) return -32768;
  if (v > 32766.5f) return 32767;
  const float fl = floorf(v);
  return static_cast<int16_t>(static_cast<int>(fl) + ((v - fl) >= 0.5f ? 1 : 0));
}

// The next call's history: the last taps-1 frames of (history ++ input[0..consumed)), i.e.
// reference resample.c:898-899 applied once over the whole call.
__device__ __forceinline__ void roll_history(uint32_t taps, uint32_t channels, const StreamDesc &d) {
  const uint32_t hist_frames = taps - 1;
  const uint32_t total = hist_frames * channels;
  for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
    const uint32_t h = i / channels, c = i - h * channels;
    const int64_t v = static_cast<int64_t>(d.consumed) + h;
    int16_t s;
    if (v < static_cast<int64_t>(hist_frames)) {
      s = as_global(d.hist)[v * channels + c];
    } else {
      const int64_t f = v - hist_frames;
      s = (d.in != nullptr && f < static_cast<int64_t>(d.in_frames)) ? as_global(d.in)[f * channels + c]
                                                                      : static_cast<int16_t>(0);
    }
    as_global(d.hist_next)[i] = s;
  }
}

// One element of the virtual sequence in "input-relative" element units: q < 0 reaches back
// into the history (the hist_elems int16 before the input), beyond either end is silence.
__device__ __forceinline__ float rel_sample(const StreamDesc &d, int64_t q, int64_t hist_elems,
                                            int64_t in_elems) {
  if (q < 0) return q >= -hist_elems ? static_cast<float>(as_global(d.hist)[q + hist_elems]) : 0.f;
  return (d.in != nullptr && q < in_elems) ? static_cast<float>(as_global(d.in)[q]) : 0.f;
}

// Stage `units` groups of 8 interleaved s16 samples starting at input-relative element q_base
// (a multiple of 8) into LDS as float.  Groups that lie wholly inside a 16-byte-aligned input
// buffer are fetched with ONE 16-byte load per lane, UNR loads in flight and no branch around
// them; the few elements that touch the history or the ends of the input (or everything, for
// an unaligned / absent buffer) take the per-element path.
template <int UNR>
__device__ __forceinline__ void stage_window(float *xs, const StreamDesc &d, int64_t q_base,
                                             uint32_t units, int64_t hist_elems, int64_t in_elems) {
  const uint32_t total = units * 8;
  uint32_t head_end = total, tail_begin = total;  // scalar ranges [0,head_end) U [tail_begin,total)
  const bool wide_ok = d.in != nullptr && in_elems >= 8 &&
                       (reinterpret_cast<uintptr_t>(d.in) & 15u) == 0;
  if (wide_ok) {
    const int64_t q_max = in_elems - 8;  // last group start that is wholly inside
    const int64_t first = q_base < 0 ? -q_base : 0;
    const int64_t beyond = (q_max / 8) * 8 + 8 - q_base;
    head_end = static_cast<uint32_t>(min(first, static_cast<int64_t>(total)));
    tail_begin = static_cast<uint32_t>(min(max(beyond, static_cast<int64_t>(head_end)),
                                           static_cast<int64_t>(total)));
    const uint32_t u_begin = head_end / 8, u_end = tail_begin / 8;
    const int16_t *src = d.in + (q_base + 8 * static_cast<int64_t>(u_begin));
    const uint32_t n = u_end - u_begin;
    for (uint32_t base = 0; base < n; base += blockDim.x * UNR) {
      u32x4 w[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t unit = min(base + u * blockDim.x + threadIdx.x, n - 1);
        w[u] = *(g_cuint4 *)(src + 8 * static_cast<size_t>(unit));
      }
#pragma unroll
      for (int u = 0; u < UNR; u++)
        asm volatile("" : "+v"(w[u].x), "+v"(w[u].y), "+v"(w[u].z), "+v"(w[u].w));
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t unit = base + u * blockDim.x + threadIdx.x;
        float4 lo, hi;
        lo.x = static_cast<float>(static_cast<int>(w[u].x << 16) >> 16);
        lo.y = static_cast<float>(static_cast<int>(w[u].x) >> 16);
        lo.z = static_cast<float>(static_cast<int>(w[u].y << 16) >> 16);
        lo.w = static_cast<float>(static_cast<int>(w[u].y) >> 16);
        hi.x = static_cast<float>(static_cast<int>(w[u].z << 16) >> 16);
        hi.y = static_cast<float>(static_cast<int>(w[u].z) >> 16);
        hi.z = static_cast<float>(static_cast<int>(w[u].w << 16) >> 16);
        hi.w = static_cast<float>(static_cast<int>(w[u].w) >> 16);
        if (unit < n) {
          float4 *dst = reinterpret_cast<float4 *>(xs + 8 * static_cast<size_t>(u_begin + unit));
          dst[0] = lo;
          dst[1] = hi;
        }
      }
    }
  }
  for (uint32_t j = threadIdx.x; j < head_end; j += blockDim.x)
    xs[j] = rel_sample(d, q_base + j, hist_elems, in_elems);
  for (uint32_t j = tail_begin + threadIdx.x; j < total; j += blockDim.x)
    xs[j] = rel_sample(d, q_base + j, hist_elems, in_elems);
}

// ---- the same staging split in two, for software pipelining ----------------------------------
// window_fetch() issues the 16-byte loads of a window into registers (nothing waits on them);
// window_commit() converts them and writes the LDS image later -- typically after the FIR of
// the previous tile, whose FMAs hide the HBM latency.
struct WindowGeom {
  int64_t q_base;              // input-relative element index of LDS float 0 (multiple of 8)
  int64_t hist_elems, in_elems;
  const int16_t *src;          // first wholly-inside group of 8
  uint32_t total;              // floats in the LDS image (multiple of 8)
  uint32_t head_end, tail_begin;  // scalar ranges [0, head_end) U [tail_begin, total)
  uint32_t u_begin, n_wide;    // wide groups: LDS floats [8*u_begin, 8*(u_begin+n_wide))
  uint32_t xshift;             // float index of the window's first frame inside the image
  uint32_t pad, period_elems;  // bank padding: `pad` floats inserted after every period_elems
                               // (= num*channels) floats counted from the window's first frame
};

__device__ __forceinline__ WindowGeom window_geom(const StreamDesc &d, uint32_t taps, uint32_t channels,
                                                  uint32_t num, uint32_t tail_frames, uint32_t m_lo,
                                                  uint32_t m_cnt, uint32_t pad = 0) {
  WindowGeom w;
  w.pad = pad;
  w.period_elems = num * channels;
  w.hist_elems = static_cast<int64_t>(taps - 1) * channels;
  w.in_elems = static_cast<int64_t>(d.in_frames) * channels;
  const int64_t q_lo =
      (static_cast<int64_t>(d.base_shift) + static_cast<int64_t>(m_lo) * num) * channels - w.hist_elems;
  w.q_base = (q_lo >= 0 ? q_lo / 8 : -((-q_lo + 7) / 8)) * 8;
  w.xshift = static_cast<uint32_t>(q_lo - w.q_base);
  const uint32_t span = (m_cnt - 1) * num + tail_frames;
  w.total = (w.xshift + span * channels + 7) / 8 * 8;
  w.head_end = w.total;
  w.tail_begin = w.total;
  w.u_begin = 0;
  w.n_wide = 0;
  w.src = d.in;
  const bool wide_ok = d.in != nullptr && w.in_elems >= 8 && (reinterpret_cast<uintptr_t>(d.in) & 15u) == 0;
  if (wide_ok) {
    const int64_t q_max = w.in_elems - 8;
    const int64_t first = w.q_base < 0 ? -w.q_base : 0;
    const int64_t beyond = (q_max / 8) * 8 + 8 - w.q_base;
    w.head_end = static_cast<uint32_t>(min(first, static_cast<int64_t>(w.total)));
    w.tail_begin = static_cast<uint32_t>(
        min(max(beyond, static_cast<int64_t>(w.head_end)), static_cast<int64_t>(w.total)));
    w.u_begin = w.head_end / 8;
    w.n_wide = w.tail_begin / 8 - w.u_begin;
    w.src = d.in + (w.q_base + 8 * static_cast<int64_t>(w.u_begin));
  }
  return w;
}

template <int UNR>
__device__ __forceinline__ void window_fetch(const WindowGeom &g, u32x4 (&w)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t unit = u * blockDim.x + threadIdx.x;
    // clamped, never branched around: all UNR loads stay in flight
    w[u] = g.n_wide ? *(g_cuint4 *)(g.src + 8 * static_cast<size_t>(min(unit, g.n_wide - 1)))
                    : u32x4{0u, 0u, 0u, 0u};
  }
}

__device__ __forceinline__ void unpack8(const u32x4 &w, float4 *dst) {
  float4 lo, hi;
  lo.x = static_cast<float>(static_cast<int>(w.x << 16) >> 16);
  lo.y = static_cast<float>(static_cast<int>(w.x) >> 16);
  lo.z = static_cast<float>(static_cast<int>(w.y << 16) >> 16);
  lo.w = static_cast<float>(static_cast<int>(w.y) >> 16);
  hi.x = static_cast<float>(static_cast<int>(w.z << 16) >> 16);
  hi.y = static_cast<float>(static_cast<int>(w.z) >> 16);
  hi.z = static_cast<float>(static_cast<int>(w.w << 16) >> 16);
  hi.w = static_cast<float>(static_cast<int>(w.w) >> 16);
  dst[0] = lo;
  dst[1] = hi;
}

// LDS float position of image float j when the layout is padded (pad floats after every
// period_elems floats counted from the window's first frame, i.e. from image float xshift)
__device__ __forceinline__ uint32_t padded_pos(const WindowGeom &g, uint32_t j) {
  return j + (j >= g.xshift ? (j - g.xshift) / g.period_elems : 0u) * g.pad;
}

// 8 consecutive image floats starting at j (multiple of 8): two 16-byte writes unless the group
// straddles a padding boundary
__device__ __forceinline__ void commit8(float *xs, const WindowGeom &g, uint32_t j, const u32x4 &w) {
  if (g.pad == 0) {
    unpack8(w, reinterpret_cast<float4 *>(xs + j));
    return;
  }
  const uint32_t a = padded_pos(g, j), b = padded_pos(g, j + 7);
  if (b - a == 7 && (a & 3u) == 0) {
    unpack8(w, reinterpret_cast<float4 *>(xs + a));
  } else {
    float4 t[2];
    unpack8(w, t);
    const float *f = reinterpret_cast<const float *>(t);
#pragma unroll
    for (int k = 0; k < 8; k++) xs[padded_pos(g, j + k)] = f[k];
  }
}

template <int UNR>
__device__ __forceinline__ void window_commit(float *xs, const StreamDesc &d, const WindowGeom &g,
                                              const u32x4 (&w)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; u++) {
    const uint32_t unit = u * blockDim.x + threadIdx.x;
    if (unit < g.n_wide) commit8(xs, g, 8 * (g.u_begin + unit), w[u]);
  }
  // groups beyond the prefetched UNR per lane (small workgroups, very wide windows): further
  // rounds of UNR loads in flight at a time
  for (uint32_t base = UNR * blockDim.x; base < g.n_wide; base += UNR * blockDim.x) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t unit = base + u * blockDim.x + threadIdx.x;
      v[u] = *(g_cuint4 *)(g.src + 8 * static_cast<size_t>(min(unit, g.n_wide - 1)));
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t unit = base + u * blockDim.x + threadIdx.x;
      if (unit < g.n_wide) commit8(xs, g, 8 * (g.u_begin + unit), v[u]);
    }
  }
  for (uint32_t j = threadIdx.x; j < g.head_end; j += blockDim.x)
    xs[g.pad ? padded_pos(g, j) : j] = rel_sample(d, g.q_base + j, g.hist_elems, g.in_elems);
  for (uint32_t j = g.tail_begin + threadIdx.x; j < g.total; j += blockDim.x)
    xs[g.pad ? padded_pos(g, j) : j] = rel_sample(d, g.q_base + j, g.hist_elems, g.in_elems);
}

// Two floats -> packed s16 pair {lo, hi} with the reference's rounding: floor(x + .5), then
// saturation to [-32768, 32767] (equivalent to arch.h:208-209: the < -32767.5 / > 32766.5
// branches are the clamp of floor(x + .5)).  The fp32 add is exact for every x but
// 0.5 - 2^-25, so no double arithmetic is needed on the fast path.
__device__ __forceinline__ uint32_t round_pack_pcm(float lo, float hi) {
  const int a = static_cast<int>(floorf(lo + 0.5f));
  const int b = static_cast<int>(floorf(hi + 0.5f));
  typedef short short2_t __attribute__((ext_vector_type(2)));
  const short2_t pk = __builtin_amdgcn_cvt_pk_i16(a, b);  // saturating v_cvt_pk_i16_i32
  return __builtin_bit_cast(uint32_t, pk);
}

}  // namespace
}  // namespace speexhip
