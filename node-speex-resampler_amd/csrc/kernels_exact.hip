// kernels_exact.hip -- bit-exact gfx950 kernels for the four Speex FIR inner loops.
//
// One lane = one output frame (x CT channels).  Each lane evaluates its FIR in EXACTLY the
// reference's association order and widths, with separate multiply and add (this file is
// compiled with -ffp-contract=off), so the int16 output is bit-identical to
//   resampler_basic_direct_single       reference deps/speex/resample.c:331-384
//   resampler_basic_direct_double       reference deps/speex/resample.c:389-435
//   resampler_basic_interpolate_single  reference deps/speex/resample.c:438-496
//   resampler_basic_interpolate_double  reference deps/speex/resample.c:501-558
// followed by WORD2INT (reference deps/speex/arch.h:208-209).  The int16 -> float
// deinterleave of resample.c:1001-1005 and the round/saturate/interleave of :1018-1022 are
// fused in.  The 160/1024 blocking of resample.c:988-1030 is NOT replayed: every output of
// a call is independent given (last0, frac0) -- see stream_plan.h.
//
// Data movement: a workgroup owns `outs_per_block` consecutive output frames of one stream.
// It stages the input frames those outputs touch (history ++ input, interleaved s16 in HBM,
// read once) into LDS as float, and the sinc table too; lanes of a wave then read adjacent
// or identical LDS words (adjacent outputs start <= 1 frame apart when up-sampling).
#include <hip/hip_runtime.h>

#include <cstring>

#include "device_helpers.h"
#include "device_types.h"
#include "filter_design.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace speexhip {
namespace {

// V = history ++ input at V-frame v, channel c (history is float, the input has the call's type T)
template <typename T>
__device__ __forceinline__ float virtual_sample(const StreamDesc &d, uint32_t hist_frames, uint32_t hist_stride,
                                                uint32_t in_stride, int64_t v, uint32_t c) {
  if (v < static_cast<int64_t>(hist_frames)) return hist_ptr(d)[v * hist_stride + c];
  v -= hist_frames;
  if (d.in == nullptr || v >= static_cast<int64_t>(d.in_frames)) return 0.f;
  return static_cast<float>(in_ptr<T>(d)[v * in_stride + c]);
}

// reference arch.h:208-209 -- the add and the floor are double
__device__ __forceinline__ int16_t word2int(float v) {
  if (v < -32767.5f) return -32768;
  if (v > 32766.5f) return 32767;
  return static_cast<int16_t>(floor(.5 + static_cast<double>(v)));
}

// reference resample.c:318-328 (float build)
__device__ __forceinline__ void cubic_weights(float f, float w[4]) {
  w[0] = -0.16667f * f + 0.16667f * f * f * f;
  w[1] = f + 0.5f * f * f - 0.5f * f * f * f;
  w[3] = -0.33333f * f + 0.5f * f * f - 0.16667f * f * f * f;
  w[2] = static_cast<float>(1. - w[0] - w[1] - w[3]);
}

// T = sample type of the call: int16_t (round + saturate on the way out, resample.c:1018-1022)
// or float (the FIR value as is, resample.c:927-963).
template <int KIND, int CT, bool STAGED, typename T>
__global__ __launch_bounds__(256) void resample_exact(ExactParams p, DescPack pack) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const StreamDesc d = pack.d[blockIdx.y];  // (the launch's descriptors travel in the kernel arguments: device_types.h)
  const uint32_t C = p.channels;
  const uint32_t hist_frames = d.hist_frames;

  if (blockIdx.x == gridDim.x - 1) {  // one extra workgroup per stream rolls the history
    if (blockIdx.z == 0) roll_history<T>(p.channels, d, p.outs_per_block, p.in_stride, p.hist_stride);
    return;
  }
  const uint32_t k_first = blockIdx.x * p.outs_per_block;
  if (k_first >= d.n_out) return;
  const uint32_t k_count = min(p.outs_per_block, d.n_out - k_first);
  const uint32_t c_first = blockIdx.z * CT;

  // window start of an output: V-frame index (V = history ++ input)
  const uint64_t t_first = static_cast<uint64_t>(d.frac0) + static_cast<uint64_t>(k_first) * p.num;
  const int64_t base = static_cast<int64_t>(d.last0) + static_cast<int64_t>(t_first / p.den);

  const float *tab = p.table;
  const float *xs = nullptr;
  if (STAGED) {
    float *tab_lds = lds;
    float *xs_lds = lds + ((p.table_len + 3) & ~3u);
    for (uint32_t i = threadIdx.x; i < p.table_len; i += p.outs_per_block) tab_lds[i] = p.table[i];
    const uint64_t t_last = t_first + static_cast<uint64_t>(k_count - 1) * p.num;
    const uint32_t span =
        static_cast<uint32_t>(static_cast<int64_t>(d.last0) + static_cast<int64_t>(t_last / p.den) - base) +
        p.taps;
    for (uint32_t i = threadIdx.x; i < span * CT; i += p.outs_per_block) {
      const uint32_t f = i / CT, ct = i - f * CT;
      const uint32_t c = c_first + ct;
      xs_lds[i] = c < C ? virtual_sample<T>(d, hist_frames, p.hist_stride, p.in_stride, base + f, c) : 0.f;
    }
    __syncthreads();
    tab = tab_lds;
    xs = xs_lds;
  }

  if (threadIdx.x >= k_count) return;
  const uint32_t k = k_first + threadIdx.x;
  const uint64_t t = static_cast<uint64_t>(d.frac0) + static_cast<uint64_t>(k) * p.num;
  const int64_t pos = static_cast<int64_t>(d.last0) + static_cast<int64_t>(t / p.den);
  const uint32_t phase = static_cast<uint32_t>(t % p.den);
  const uint32_t rel = static_cast<uint32_t>(pos - base);
  const int n = static_cast<int>(p.taps);

  auto sample = [&](int j, int ct) -> float {
    if (STAGED) return xs[(rel + j) * CT + ct];
    const uint32_t c = c_first + ct;
    return c < C ? virtual_sample<T>(d, hist_frames, p.hist_stride, p.in_stride, pos + j, c) : 0.f;
  };

  float y[CT];
  if (p.zero) {  // resample.c:561-591: the filter could not be built -- lengths stay right, samples are zero
#pragma unroll
    for (int ct = 0; ct < CT; ct++) y[ct] = 0.f;
  } else if (KIND == kDirectSingle) {
    const float *h = tab + static_cast<size_t>(phase) * n;
    float s[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) s[ct] = 0.f;
    for (int j = 0; j < n; j++) {
      const float hj = h[j];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) s[ct] = s[ct] + hj * sample(j, ct);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) y[ct] = s[ct];
  } else if (KIND == kDirectDouble) {
    const float *h = tab + static_cast<size_t>(phase) * n;
    double a[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) a[ct][0] = a[ct][1] = a[ct][2] = a[ct][3] = 0.;
    for (int j = 0; j < n; j += 4) {
      const float h0 = h[j], h1 = h[j + 1], h2 = h[j + 2], h3 = h[j + 3];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) {
        a[ct][0] += static_cast<double>(h0 * sample(j, ct));
        a[ct][1] += static_cast<double>(h1 * sample(j + 1, ct));
        a[ct][2] += static_cast<double>(h2 * sample(j + 2, ct));
        a[ct][3] += static_cast<double>(h3 * sample(j + 3, ct));
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
      y[ct] = static_cast<float>(a[ct][0] + a[ct][1] + a[ct][2] + a[ct][3]);
  } else {
    // uint32 products as in the reference (resample.c:454,458)
    const uint32_t scaled = phase * p.oversample;
    const int offset = static_cast<int>(scaled / p.den);
    const float frac = (static_cast<float>(scaled % p.den)) / p.den;
    float w[4];
    cubic_weights(frac, w);
    const float *t0 = tab + 4 + p.oversample - offset - 2;  // j = 0
    if (KIND == kInterpolateSingle) {
      float a[CT][4];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) a[ct][0] = a[ct][1] = a[ct][2] = a[ct][3] = 0.f;
      for (int j = 0; j < n; j++) {
        const float *tj = t0 + static_cast<size_t>(j) * p.oversample;
        const float c0 = tj[0], c1 = tj[1], c2 = tj[2], c3 = tj[3];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
          const float x = sample(j, ct);
          a[ct][0] = a[ct][0] + x * c0;
          a[ct][1] = a[ct][1] + x * c1;
          a[ct][2] = a[ct][2] + x * c2;
          a[ct][3] = a[ct][3] + x * c3;
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ct++)
        y[ct] = w[0] * a[ct][0] + w[1] * a[ct][1] + w[2] * a[ct][2] + w[3] * a[ct][3];
    } else {
      double a[CT][4];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) a[ct][0] = a[ct][1] = a[ct][2] = a[ct][3] = 0.;
      for (int j = 0; j < n; j++) {
        const float *tj = t0 + static_cast<size_t>(j) * p.oversample;
        const float c0 = tj[0], c1 = tj[1], c2 = tj[2], c3 = tj[3];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
          const float x = sample(j, ct);
          a[ct][0] += static_cast<double>(x * c0);
          a[ct][1] += static_cast<double>(x * c1);
          a[ct][2] += static_cast<double>(x * c2);
          a[ct][3] += static_cast<double>(x * c3);
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ct++)
        y[ct] = static_cast<float>(w[0] * a[ct][0] + w[1] * a[ct][1] + w[2] * a[ct][2] +
                                   w[3] * a[ct][3]);
    }
  }

  if constexpr (sizeof(T) == 4) {  // float I/O: no rounding
    G<float> *o = out_ptr<float>(d) + static_cast<size_t>(k) * p.out_stride + c_first;
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
      if (c_first + ct < C) o[ct] = y[ct];
  } else {
    G<int16_t> *o = out_ptr<int16_t>(d) + static_cast<size_t>(k) * p.out_stride + c_first;
    if (CT == 2 && c_first + 1 < C && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
      // both channels of an even-channel frame: one aligned 32-bit store
      const uint32_t packed = static_cast<uint16_t>(word2int(y[0])) |
                              (static_cast<uint32_t>(static_cast<uint16_t>(word2int(y[CT - 1]))) << 16);
      *(g_u32 *)o = packed;
    } else {
#pragma unroll
      for (int ct = 0; ct < CT; ct++)
        if (c_first + ct < C) o[ct] = word2int(y[ct]);
    }
  }
}

template <int KIND, int CT, bool STAGED, typename T>
hipError_t launch_k(const ExactParams &p, const DescPack *pack, dim3 grid, size_t lds_bytes, hipStream_t stream) {
  auto kern = resample_exact<KIND, CT, STAGED, T>;
  static std::atomic<uint64_t> seen{0};  // allow the full 160 KiB of dynamic LDS, once per device
  opt_in_lds_on_this_device(kern, seen);
  hipLaunchKernelGGL(kern, grid, dim3(p.outs_per_block), lds_bytes, stream, p, *pack);
  return hipGetLastError();
}

template <int KIND, typename T>
hipError_t launch_kind(const ExactParams &p, const DescPack *pack, int ct,
                       bool staged, dim3 grid, size_t lds, hipStream_t s) {
  if (ct == 2)
    return staged ? launch_k<KIND, 2, true, T>(p, pack, grid, lds, s)
                  : launch_k<KIND, 2, false, T>(p, pack, grid, 0, s);
  return staged ? launch_k<KIND, 1, true, T>(p, pack, grid, lds, s)
                : launch_k<KIND, 1, false, T>(p, pack, grid, 0, s);
}

template <typename T>
hipError_t launch_typed(const FilterSpec &f, const ExactParams &p, const DescPack *pack,
                        const ExactGeometry &g, dim3 grid, hipStream_t s) {
  switch (f.kind) {
    case kDirectSingle: return launch_kind<kDirectSingle, T>(p, pack, g.ct, g.staged, grid, g.lds_bytes, s);
    case kDirectDouble: return launch_kind<kDirectDouble, T>(p, pack, g.ct, g.staged, grid, g.lds_bytes, s);
    case kInterpolateSingle:
      return launch_kind<kInterpolateSingle, T>(p, pack, g.ct, g.staged, grid, g.lds_bytes, s);
    default: return launch_kind<kInterpolateDouble, T>(p, pack, g.ct, g.staged, grid, g.lds_bytes, s);
  }
}

}  // namespace

ExactGeometry exact_geometry(const FilterSpec &f, uint32_t channels, size_t lds_budget) {
  ExactGeometry g;
  g.ct = (channels % 2 == 0) ? 2 : 1;
  g.channel_groups = channels / g.ct;
  const size_t table_bytes = ((static_cast<size_t>(f.table_len) + 3) & ~size_t(3)) * 4;
  for (uint32_t opb : {256u, 128u, 64u}) {
    const uint64_t span = (static_cast<uint64_t>(opb - 1) * f.num + f.den - 1) / f.den + f.taps + 1;
    const size_t need = table_bytes + span * g.ct * 4;
    if (need <= lds_budget) {
      g.outs_per_block = opb;
      g.span_cap = static_cast<uint32_t>(span);
      g.lds_bytes = need;
      g.staged = true;
      return g;
    }
  }
  g.outs_per_block = 256;  // too long a filter for LDS: stream straight from L2/HBM
  g.span_cap = 0;
  g.lds_bytes = 0;
  g.staged = false;
  return g;
}

hipError_t launch_exact(const FilterSpec &f, const ExactGeometry &g, const float *d_table,
                        uint32_t channels, const DescPack *pack,
                        uint32_t n_streams, uint32_t max_n_out, bool float_io, hipStream_t stream,
                        const ExactStrides *strides, bool zero) {
  ExactParams p;
  p.table = d_table;
  p.table_len = f.table_len;
  p.num = f.num;
  p.den = f.den;
  p.taps = f.taps;
  p.oversample = f.oversample;
  p.channels = channels;
  p.outs_per_block = g.outs_per_block;
  p.span_cap = g.span_cap;
  p.in_stride = strides ? strides->in : channels;
  p.out_stride = strides ? strides->out : channels;
  p.hist_stride = strides ? strides->hist : channels;
  p.zero = zero ? 1u : 0u;
  const uint32_t blocks = (max_n_out + g.outs_per_block - 1) / g.outs_per_block;
  dim3 grid(blocks + 1, n_streams, g.channel_groups);
  return float_io ? launch_typed<float>(f, p, pack, g, grid, stream)
                  : launch_typed<int16_t>(f, p, pack, g, grid, stream);
}

// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(exact)

}  // namespace speexhip
