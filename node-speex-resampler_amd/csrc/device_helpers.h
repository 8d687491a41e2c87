// device_helpers.h -- device-side pieces shared by the fast kernels (kernels_tiled.hip,
// kernels_period.hip): staging of the interleaved s16 input window into LDS as float, rounding
// to PCM, and the history roll.  Included only from .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace speexhip {
namespace {

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
      s = d.hist[v * channels + c];
    } else {
      const int64_t f = v - hist_frames;
      s = (d.in != nullptr && f < static_cast<int64_t>(d.in_frames)) ? d.in[f * channels + c]
                                                                      : static_cast<int16_t>(0);
    }
    d.hist_next[i] = s;
  }
}

// One element of the virtual sequence in "input-relative" element units: q < 0 reaches back
// into the history (the hist_elems int16 before the input), beyond either end is silence.
__device__ __forceinline__ float rel_sample(const StreamDesc &d, int64_t q, int64_t hist_elems,
                                            int64_t in_elems) {
  if (q < 0) return q >= -hist_elems ? static_cast<float>(d.hist[q + hist_elems]) : 0.f;
  return (d.in != nullptr && q < in_elems) ? static_cast<float>(d.in[q]) : 0.f;
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
      uint4 w[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t unit = min(base + u * blockDim.x + threadIdx.x, n - 1);
        w[u] = *reinterpret_cast<const uint4 *>(src + 8 * static_cast<size_t>(unit));
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
