// kernels_tiled.hip -- the fast gfx950 FIR kernel: per-phase tap rows in LDS, register-tiled
// FMA, wave-level reduction across a split tap range.
//
// Algebra.  With K = k_shift + k (k_shift = phase_index_of(frac0), stream_plan.h) every output
// of every stream has the canonical decomposition K = m*den + r:
//     phase = (r*num) mod den,   window start = base_shift + m*num + (r*num) div den.
// Outputs with equal r share their taps; outputs with nearby r share their input window; the
// reference's interpolated kernels (deps/speex/resample.c:438-558) collapse to ONE dot product
// per output against the effective taps  H_r[j] = sum_t w_r[t] * table[4+(j+1)*os-off_r-2+t]
// (filter_design.cpp: phase_taps), the direct kernels (resample.c:331-435) already are one.
// So the whole call is   Out[r, m, c] = sum_j H_r[j] * V[base + m*num + delta_r + j][c]
// -- a Toeplitz-structured contraction evaluated with plain fp32 FMA on the vector ALUs (no
// MFMA).  Results are within +-1 LSB of the reference (different association order, FMA, taps
// pre-blended in double); the bit-exact path is kernels_exact.hip.
//
// Mapping.  A workgroup owns `periods` consecutive values of m for ALL r of one stream.
//   LDS:  tap rows  T[s/4][i][g] (float4), rows pre-shifted by d = delta_r - delta_{g*R} so that
//         the R rows of a group read the SAME input sample at the same s (16-byte aligned
//         ds_read_b128, consecutive g on consecutive 16-byte slots: conflict-free);
//         input window as float, channel-interleaved (ds_read_b64 for a channel pair).
//   lane: (ks, g, mg, cg): tap-range slice ks of KS, phase group g (R phases), period group mg
//         (M periods), channel group cg (CT channels).  R*M*CT accumulators per lane; per
//         4 taps it issues R ds_read_b128 + 4*M sample reads for 4*R*M*CT FMAs.
//   The KS partial sums are combined with a butterfly over adjacent lanes (wave-level MAC
//   reduction), then rounded (arch.h:208-209 semantics), interleaved and stored.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_helpers.h"
#include "device_types.h"
#include "filter_design.h"
#include "kernels.h"

namespace speexhip {
namespace {

// Copy of the n float4 of tap rows global -> LDS with UNR independent loads in flight per lane
// (indices clamped instead of branched around, values pinned, so hipcc keeps all loads
// outstanding).  Every `slice` float4 the destination skips `pad` float4: the tap slices of the
// KS lanes of a quad then start on different 16-byte bank slots (conflict-free ds_read_b128).
template <int UNR>
__device__ __forceinline__ void stage_rows(float4 *dst, const float4 *src, uint32_t n, uint32_t slice,
                                           uint32_t pad) {
  for (uint32_t base = 0; base < n; base += blockDim.x * UNR) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t i = base + u * blockDim.x + threadIdx.x;
      v[u] = src[min(i, n - 1)];
    }
#pragma unroll
    for (int u = 0; u < UNR; u++)
      asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const uint32_t i = base + u * blockDim.x + threadIdx.x;
      if (i < n) dst[i + (i / slice) * pad] = v[u];
    }
  }
}

// lane <-> lane^1 and lane <-> lane^2 exchanges inside a quad (DPP quad_perm, no LDS)
__device__ __forceinline__ float quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

template <int R, int M, int CT, bool PACKED>
__global__ __launch_bounds__(768) void resample_tiled(TiledParams p, const StreamDesc *streams,
                                                      DescPack pack) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const StreamDesc d = PACKED ? pack.d[blockIdx.y] : streams[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {
    roll_history(p.taps, p.channels, d);
    return;
  }
  if (d.n_out == 0) return;
  const uint32_t K_end = d.k_shift + d.n_out;               // exclusive canonical index
  const uint32_t m_total = (K_end + p.den - 1) / p.den;     // periods touched by this call
  const uint32_t m_lo = blockIdx.x * p.periods;
  if (m_lo >= m_total) return;
  const uint32_t m_cnt = min(p.periods, m_total - m_lo);

  float4 *T = reinterpret_cast<float4 *>(lds);
  float *xs = lds + static_cast<size_t>(p.table_f4 + (p.ksplit - 1) * p.slice_pad_f4) * 4;

  // ---- stage the input window: interleaved s16 in HBM -> float in LDS ----------------------
  // Window origin in input-relative elements, rounded down to the input's 16-byte grid.
  const int64_t hist_elems = static_cast<int64_t>(p.taps - 1) * p.channels;
  const int64_t in_elems = static_cast<int64_t>(d.in_frames) * p.channels;
  const int64_t q_lo = (static_cast<int64_t>(d.base_shift) + static_cast<int64_t>(m_lo) * p.num) *
                           p.channels - hist_elems;
  const int64_t q_base = (q_lo >= 0 ? q_lo / 8 : -((-q_lo + 7) / 8)) * 8;
  const uint32_t xshift = static_cast<uint32_t>(q_lo - q_base);
  const uint32_t span = (m_cnt - 1) * p.num + p.tail_frames;
  if (!(p.skip & 2u))
    stage_window<4>(xs, d, q_base, (xshift + span * p.channels + 7) / 8, hist_elems, in_elems);
  // ---- stage the tap rows (L2-resident after the first workgroups) -------------------------
  if (!(p.skip & 1u))
    stage_rows<8>(T, reinterpret_cast<const float4 *>(p.rows), p.table_f4, p.slice_f4, p.slice_pad_f4);
  __syncthreads();

  // ---- lane coordinates -------------------------------------------------------------------
  uint32_t t = threadIdx.x;
  const uint32_t ks = t % p.ksplit;
  t /= p.ksplit;
  const uint32_t g = t % p.groups;
  t /= p.groups;
  const uint32_t mg = t % p.mgroups;
  const uint32_t cg = t / p.mgroups;
  const bool lane_live = cg < p.cgroups;
  const uint32_t cgc = lane_live ? cg : 0;

  const uint32_t delta_g = static_cast<uint32_t>((static_cast<uint64_t>(g) * R * p.num) / p.den);
  uint32_t xoff[M];
#pragma unroll
  for (int mi = 0; mi < M; mi++) {
    uint32_t m = mg * M + mi;
    if (m >= m_cnt) m = m_cnt - 1;  // idle periods recompute the last one, never stored
    xoff[mi] = xshift + (m * p.num + delta_g) * p.channels + cgc * CT;
  }

  float acc[R][M][CT];
#pragma unroll
  for (int i = 0; i < R; i++)
#pragma unroll
    for (int mi = 0; mi < M; mi++)
#pragma unroll
      for (int ct = 0; ct < CT; ct++) acc[i][mi][ct] = 0.f;

  const uint32_t s4_begin = ks * p.s4_per_slice;
  const uint32_t s4_end = (p.skip & 4u) ? s4_begin : min(s4_begin + p.s4_per_slice, p.l4);
  const uint32_t C = p.channels;
  const float4 *tp = T + ks * (p.slice_f4 + p.slice_pad_f4) + g;  // this lane's slice, its group
  for (uint32_t s4 = s4_begin; s4 < s4_end; s4++, tp += R * p.groups) {
    float4 tap[R];
#pragma unroll
    for (int i = 0; i < R; i++) tap[i] = tp[i * p.groups];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float x[M][CT];
#pragma unroll
      for (int mi = 0; mi < M; mi++) {
        const float *px = xs + xoff[mi] + (s4 * 4 + u) * C;
        if (CT == 2) {
          const float2 v = *reinterpret_cast<const float2 *>(px);
          x[mi][0] = v.x;
          x[mi][CT - 1] = v.y;
        } else {
          x[mi][0] = *px;
        }
      }
#pragma unroll
      for (int i = 0; i < R; i++) {
        const float h = u == 0 ? tap[i].x : u == 1 ? tap[i].y : u == 2 ? tap[i].z : tap[i].w;
#pragma unroll
        for (int mi = 0; mi < M; mi++)
#pragma unroll
          for (int ct = 0; ct < CT; ct++) acc[i][mi][ct] = fmaf(h, x[mi][ct], acc[i][mi][ct]);
      }
    }
  }

  // ---- wave-level reduction over the tap slices (adjacent lanes, DPP butterfly) ------------
  if (p.ksplit >= 2) {
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
      for (int mi = 0; mi < M; mi++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[i][mi][ct] += quad_xor1(acc[i][mi][ct]);
  }
  if (p.ksplit >= 4) {
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
      for (int mi = 0; mi < M; mi++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[i][mi][ct] += quad_xor2(acc[i][mi][ct]);
  }

  // ---- round + interleave into an LDS image of the tile's output, then coalesced stores ----
  // The tile's valid frames [Kb, Ke) are one contiguous run of the stream's output.  The LDS
  // image reuses the input window; its origin is chosen so that LDS element index == global
  // element index (mod 8), which makes every interior 16-byte chunk aligned on both sides.
  if (p.skip & 8u) return;
  __syncthreads();  // every lane is done reading the window
  int16_t *ytile = reinterpret_cast<int16_t *>(xs);
  const uint64_t K_lo = static_cast<uint64_t>(m_lo) * p.den;
  const uint64_t Kb = max(K_lo, static_cast<uint64_t>(d.k_shift));
  const uint64_t Ke = min(K_lo + static_cast<uint64_t>(m_cnt) * p.den, static_cast<uint64_t>(K_end));
  int16_t *gout = d.out + (Kb - d.k_shift) * C;  // first element this tile writes
  const uint32_t lead = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(gout) >> 1) & 7u);
  const uint32_t n_elems = static_cast<uint32_t>(Ke - Kb) * C;
  if (lane_live) {
#pragma unroll
    for (int i = 0; i < R; i++) {
      if ((static_cast<uint32_t>(i) & (p.ksplit - 1)) != ks) continue;  // slice ks owns rows i == ks (mod KS)
      const uint32_t r = g * R + i;
      if (r >= p.den) continue;
#pragma unroll
      for (int mi = 0; mi < M; mi++) {
        const uint32_t m = mg * M + mi;
        const uint64_t K = K_lo + static_cast<uint64_t>(m) * p.den + r;
        if (m >= m_cnt || K < Kb || K >= Ke) continue;
        const uint32_t li = lead + static_cast<uint32_t>(K - Kb) * C + cg * CT;
        if (CT == 2 && (li & 1u) == 0) {
          const uint32_t packed =
              static_cast<uint16_t>(round_pcm(acc[i][mi][0])) |
              (static_cast<uint32_t>(static_cast<uint16_t>(round_pcm(acc[i][mi][CT - 1]))) << 16);
          *reinterpret_cast<uint32_t *>(ytile + li) = packed;
        } else {
#pragma unroll
          for (int ct = 0; ct < CT; ct++) ytile[li + ct] = round_pcm(acc[i][mi][ct]);
        }
      }
    }
  }
  __syncthreads();
  const uint32_t chunks = (lead + n_elems + 7) / 8;
  for (uint32_t j = threadIdx.x; j < chunks; j += blockDim.x) {
    const uint32_t e0 = j * 8;  // LDS element index of this 16-byte chunk
    if (e0 >= lead && e0 + 8 <= lead + n_elems) {
      *reinterpret_cast<uint4 *>(gout + (e0 - lead)) = *reinterpret_cast<const uint4 *>(ytile + e0);
    } else {
      for (uint32_t e = max(e0, lead); e < min(e0 + 8, lead + n_elems); e++) gout[e - lead] = ytile[e];
    }
  }
}

template <int R, int M, int CT>
hipError_t launch_rmc(const TiledParams &p, const StreamDesc *d_descs, const DescPack *pack, dim3 grid,
                      uint32_t threads, size_t lds_bytes, hipStream_t stream) {
  if (pack != nullptr) {
    auto kern = resample_tiled<R, M, CT, true>;
    static bool lds_opt_in = false;  // once per kernel: allow the full 160 KiB of dynamic LDS
    if (!lds_opt_in) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      lds_opt_in = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, stream, p, nullptr, *pack);
  } else {
    DescPack empty;
    std::memset(&empty, 0, sizeof(empty));
    auto kern = resample_tiled<R, M, CT, false>;
    static bool lds_opt_in = false;  // once per kernel: allow the full 160 KiB of dynamic LDS
    if (!lds_opt_in) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      lds_opt_in = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, stream, p, d_descs, empty);
  }
  return hipGetLastError();
}

uint32_t round_up(uint32_t v, uint32_t q) { return (v + q - 1) / q * q; }

// floats of LDS beyond span*channels: the window starts on the input's 16-byte grid (<= 7
// elements early) and is staged in groups of 8
const size_t kWindowSlack = 16;

}  // namespace

TiledPlan plan_tiled(const FilterSpec &f, uint32_t channels, size_t lds_budget) {
  TiledPlan t;
  t.ct = (channels % 2 == 0) ? 2 : 1;
  t.cgroups = channels / t.ct;
  if (f.den >= 7) {
    t.r = 10;
    t.m = 4;
  } else if (f.den >= 2) {
    t.r = 2;
    t.m = 8;
  } else {
    t.r = 1;
    t.m = 8;
  }
  t.groups = (f.den + t.r - 1) / t.r;
  // largest intra-group shift d = delta_r - delta_{g*R}
  uint32_t dmax = 0;
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint64_t d0 = (static_cast<uint64_t>(g) * t.r * f.num) / f.den;
    const uint32_t r_last = std::min<uint32_t>(g * t.r + t.r - 1, f.den - 1);
    const uint64_t d1 = (static_cast<uint64_t>(r_last) * f.num) / f.den;
    dmax = std::max<uint32_t>(dmax, static_cast<uint32_t>(d1 - d0));
  }
  t.row_len = round_up(f.taps + dmax, 4);
  t.l4 = t.row_len / 4;
  t.table_f4 = t.l4 * t.r * t.groups;
  const uint64_t delta_last_group = (static_cast<uint64_t>(t.groups - 1) * t.r * f.num) / f.den;
  t.tail_frames = static_cast<uint32_t>(delta_last_group) + t.row_len;
  t.table_bytes = static_cast<size_t>(t.table_f4) * 16;
  t.lds_budget = lds_budget;
  // a workgroup must hold the rows plus at least M periods of input
  const uint64_t min_span = static_cast<uint64_t>(t.m - 1) * f.num + t.tail_frames;
  const size_t min_image = (static_cast<size_t>(t.m) * f.den * channels + 8) * 2;
  t.usable = t.table_bytes + 3 * 15 * 16 + kWindowSlack * 4 +
                     std::max<size_t>(min_span * channels * 4, min_image) <= lds_budget &&
             t.groups * t.cgroups <= 768;
  return t;
}

void build_phase_rows(const FilterSpec &f, const TiledPlan &t, std::vector<float> *rows) {
  rows->assign(static_cast<size_t>(t.table_f4) * 4, 0.f);
  std::vector<double> h(f.taps);
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint64_t d0 = (static_cast<uint64_t>(g) * t.r * f.num) / f.den;
    for (uint32_t i = 0; i < t.r; i++) {
      const uint32_t r = g * t.r + i;
      if (r >= f.den) continue;  // padding rows of the last group stay zero
      const uint32_t phase = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) % f.den);
      const uint32_t shift = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) / f.den - d0);
      phase_taps(f, phase, h.data());
      for (uint32_t j = 0; j < f.taps; j++) {
        const uint32_t s = j + shift;
        (*rows)[((static_cast<size_t>(s / 4) * t.r + i) * t.groups + g) * 4 + (s & 3)] =
            static_cast<float>(h[j]);
      }
    }
  }
}

TiledLaunch tiled_geometry(const FilterSpec &f, const TiledPlan &t, uint32_t channels,
                           uint32_t n_streams, uint32_t max_periods, uint32_t target_workgroups) {
  TiledLaunch L;
  const uint32_t kMaxLanes = 768;  // 12 waves: 3 per SIMD at <= 168 VGPRs
  // The LDS behind the tap rows holds first the input window (float), then the tile's output
  // image (int16): periods are bounded by both, by the lane budget and by the chip fill.
  const size_t pad_room = 3 * 15 * 16;  // worst-case slice padding
  const size_t room = t.lds_budget - t.table_bytes - pad_room - kWindowSlack * 4;
  auto window_bytes = [&](uint64_t periods) {
    return ((periods - 1) * f.num + t.tail_frames) * channels * size_t(4);
  };
  auto image_bytes = [&](uint64_t periods) { return (periods * f.den * channels + 8) * size_t(2); };
  uint32_t want = (max_periods * n_streams + target_workgroups - 1) / std::max(1u, target_workgroups);
  want = round_up(std::max(want, 1u), t.m);
  const uint32_t base_lanes = t.groups * t.cgroups;
  uint32_t periods = want;
  while (periods > static_cast<uint32_t>(t.m) &&
         (window_bytes(periods) > room || image_bytes(periods) > room ||
          base_lanes * (periods / t.m) > kMaxLanes))
    periods -= t.m;
  const uint32_t mgroups = periods / t.m;
  uint32_t ksplit = 1;
  while (ksplit < 4 && base_lanes * mgroups * ksplit * 2 <= kMaxLanes && t.l4 / (ksplit * 2) >= 4)
    ksplit *= 2;
  L.periods = periods;
  L.mgroups = mgroups;
  L.ksplit = ksplit;
  L.s4_per_slice = (t.l4 + ksplit - 1) / ksplit;
  L.slice_f4 = L.s4_per_slice * t.r * t.groups;
  // slice starts 16/KS bank slots apart (mod 16): the KS lanes of a quad never share a slot
  const uint32_t target = ksplit > 1 ? 16 / ksplit : 0;
  L.slice_pad_f4 = ksplit > 1 ? (target + 16 - L.slice_f4 % 16) % 16 : 0;
  L.threads = round_up(base_lanes * mgroups * ksplit, 64);
  L.lds_bytes = t.table_bytes + size_t(ksplit - 1) * L.slice_pad_f4 * 16 + kWindowSlack * 4 +
                std::max(window_bytes(periods), image_bytes(periods));
  L.blocks = (max_periods + periods - 1) / periods;
  return L;
}

hipError_t launch_tiled(const FilterSpec &f, const TiledPlan &t, const float *d_rows, uint32_t channels,
                        const StreamDesc *h_descs, const StreamDesc *d_descs, const DescPack *pack,
                        uint32_t n_streams, uint32_t max_n_out, hipStream_t stream) {
  // periods touched by the busiest stream
  uint32_t max_periods = 0;
  for (uint32_t s = 0; s < n_streams; s++) {
    if (h_descs[s].n_out == 0) continue;
    const uint64_t k_end = static_cast<uint64_t>(h_descs[s].k_shift) + h_descs[s].n_out;
    max_periods = std::max<uint32_t>(max_periods, static_cast<uint32_t>((k_end + f.den - 1) / f.den));
  }
  (void)max_n_out;
  const TiledLaunch L = tiled_geometry(f, t, channels, n_streams, std::max(max_periods, 1u), 256);
  TiledParams p;
  p.rows = d_rows;
  p.table_f4 = t.table_f4;
  p.l4 = t.l4;
  p.groups = t.groups;
  p.cgroups = t.cgroups;
  p.num = f.num;
  p.den = f.den;
  p.taps = f.taps;
  p.channels = channels;
  p.periods = L.periods;
  p.mgroups = L.mgroups;
  p.ksplit = L.ksplit;
  p.s4_per_slice = L.s4_per_slice;
  p.slice_f4 = L.slice_f4;
  p.slice_pad_f4 = L.slice_pad_f4;
  p.tail_frames = t.tail_frames;
  static const uint32_t skip_mask = std::getenv("SPEEXHIP_SKIP") ? std::atoi(std::getenv("SPEEXHIP_SKIP")) : 0;
  p.skip = skip_mask;
  dim3 grid((max_periods == 0 ? 0 : L.blocks) + 1, n_streams, 1);
#define SPEEXHIP_TILED_CASE(RR, MM)                                                             \
  if (t.r == RR && t.m == MM)                                                                   \
    return t.ct == 2 ? launch_rmc<RR, MM, 2>(p, d_descs, pack, grid, L.threads, L.lds_bytes, stream) \
                     : launch_rmc<RR, MM, 1>(p, d_descs, pack, grid, L.threads, L.lds_bytes, stream);
  SPEEXHIP_TILED_CASE(10, 4)
  SPEEXHIP_TILED_CASE(2, 8)
  SPEEXHIP_TILED_CASE(1, 8)
#undef SPEEXHIP_TILED_CASE
  return hipErrorInvalidValue;
}

}  // namespace speexhip
