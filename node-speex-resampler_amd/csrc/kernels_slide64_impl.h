#pragma once
// kernels_slide64_impl.h -- the small-ratio fast kernel with an fp64 accumulator (round 4): what FAST mode runs
// for the reference's "double" kernels (quality 9 and 10: resampler_basic_direct_double, deps/speex/resample.c:
// 389-435, and resampler_basic_interpolate_double, :501-558) on the ratios of the slide kernel -- BASELINE
// configs[2] (24k -> 48k mono q10) among them.  Included by kernels_slide64_i16.hip / _f32.hip (one translation
// unit per sample type).
//
// The reference sums fp32 PRODUCTS (sinct[j]*iptr[j] is a float x float in C) in four fp64 partial sums; the
// fp32 FMA chain of the other fast kernels is narrower than that, so until round 4 only EXACT mode ran these
// filters at the reference's precision (88 us for one 2^20-frame call of configs[2], one lane per output).  Here
// every product is exact and the sum fp64 throughout -- v_fma_f64 on a sample widened once per LDS read and a tap
// designed, and kept, in double -- which is WIDER than the reference: what is left of the +-1 LSB tolerance is the
// reference's own rounding of each product to fp32 (tools/seg64_sim.py: ~5e-4 of the samples differ, by 1).
// v_fma_f64 with an SGPR-pair source issues at the rate of v_pk_fma_f32 (tools/ubench_fma64.hip: 4.3 cycles per wave
// instruction at 8 waves per SIMD), i.e. half the multiply-adds per instruction: the roofline of this kernel is
// the 78.6 TFLOP/s fp64 vector peak.
//
// Mapping (the slide kernel's, without its packing games -- an fp64 FMA has no second half to fill):
//   lane  = a block of P consecutive output periods of ONE channel: P*DEN fp64 accumulators and a register
//           window of (2P-1)*NUM samples as doubles; consecutive iterations (U = P*NUM tap steps) overlap by
//           (P-1)*NUM samples, which stay in registers (ring[f mod 2U], two copies of the loop body), so an
//           iteration reads and widens only its U new samples: P*DEN FMAs per LDS read + conversion;
//   taps  = wave-uniform doubles [step][phase] (phase rows shifted by delta_r), scalar loads -> SGPR pairs;
//   LDS   = the tile's input as float, rows of P*NUM frames, the slide kernel's image and loader;
//   out   = fp64 sum -> fp32 (the reference's spx_word16_t store, round to nearest even) -> WORD2INT; a lane owns
//           P*DEN consecutive frames of its channel: mono runs leave as whole dwords, the two lanes of a stereo
//           frame swap halves through DPP so that each writes P*DEN/2 whole frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "device_helpers.h"
#include "device_types.h"
#include "filter_design.h"
#include "kernels.h"

namespace speexhip {
namespace {

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned wide store
typedef __attribute__((address_space(1))) u32x4_a4 g_u32x4_a4;
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(1))) u32x2_a4 g_u32x2_a4;
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for64(F &&f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>());
    static_for64<LO + 1, HI>(f);
  }
}

// acc += tap * x, the tap a wave-uniform double in an SGPR pair
__device__ __forceinline__ void fma64(double &acc, const double &tap, const double &x) {
  asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "s"(tap), "v"(x));
}

// the value of the lane beside this one in its pair (quad_perm [1,0,3,2])
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }

// P: periods per lane; NUM / DEN: the ratio; U = P*NUM tap steps per iteration.  CH: the channel count when it is 1 or 2
// (every LDS offset of the FIR loop is then an immediate: with the count in a register the loop spent one vector add
// per sample read on addresses -- 8 of the 152 vector instructions of an iteration of BASELINE configs[2]), else 0.
template <int P, int NUM, int DEN, int CH, typename T>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(96))) void resample_slide64(
    SlideParams p, const double *__restrict__ rows, DescPack pack) {
  // The LDS image holds the samples as DOUBLES where a read feeds at least four FMAs (P*DEN >= 4: BASELINE configs[2]
  // has 16): widened once while staging instead of once per read in the loop -- the conversions were 8 of the 136
  // vector instructions of its iteration.  Below that (n:1 decimation: one or two FMAs per read) 8-byte reads would
  // make the LDS the bound, so those shapes keep the float image and widen in the loop.
  constexpr bool LD64 = P * DEN >= 4;
  using E = std::conditional_t<LD64, double, float>;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  E *xs = reinterpret_cast<E *>(lds_raw);
  const StreamDesc d = pack.d[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {
    roll_history<T>(p.channels, d, p.threads);
    return;
  }
  if (d.n_out == 0) return;
  const uint32_t C = CH != 0 ? static_cast<uint32_t>(CH) : p.channels;  // (= p.cgroups: one lane per channel of a lane block)
  const uint32_t K_end = d.k_shift + d.n_out;
  const uint32_t m_total = d.m_total;
  const uint32_t tile_periods = p.blocks_per_tile * P;
  const uint32_t m_lo = blockIdx.x * tile_periods;
  if (m_lo >= m_total) return;
  const uint32_t m_cnt = min(tile_periods, m_total - m_lo);

  // ---- stage: the slide kernel's image (kernels_slide_impl.h): rows of P*NUM frames, row_stride floats apart ----
  WindowGeom wg;
  if (!window_geom_plain<T>(d, C, NUM, NUM + p.row_len + P * NUM, m_lo, m_cnt, p.threads, p.row_stride - P * NUM * C,
                            p.row_magic, &wg, P * NUM * C))
    wg = window_geom<T>(d, C, NUM, NUM + p.row_len + P * NUM, m_lo, m_cnt, p.threads, p.row_stride - P * NUM * C,
                        p.row_magic, P * NUM * C);
  if (!SPEEXHIP_DIAG_SKIP(p, 2u)) {
    u32x4 w[4];
    window_fetch<4, T>(wg, w);
    window_commit<4, T, E>(xs, d, wg, w);
  }
  __syncthreads();

  // tap-range parts (p.parts > 1, small launches of long filters): as in the slide kernel
  uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t part = 0;
  while (wave >= p.base_waves) {  // (wave-uniform)
    wave -= p.base_waves;
    part++;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t c = lane % C;                                // channel of this lane
  const uint32_t lb = wave * p.blocks_per_wave + lane / C;    // lane block inside the tile
  const bool lane_live = (lane / C) < p.blocks_per_wave && lb * P < m_cnt;
  const E *xrow = xs + wg.xshift + min(lb, p.blocks_per_tile - 1) * p.row_stride + c;

  double acc[P][DEN];
#pragma unroll
  for (int pp = 0; pp < P; pp++)
#pragma unroll
    for (int r = 0; r < DEN; r++) acc[pp][r] = 0.0;

  constexpr int U = P * NUM;            // tap steps per iteration
  constexpr int TAPS_IT = U * DEN;      // tap doubles per iteration
  constexpr int W = (2 * P - 1) * NUM, OLD = W - U;
  constexpr bool CARRY = OLD > 0;
  constexpr int RING = CARRY ? 2 * U : U;
  constexpr bool TAP2 = CARRY && TAPS_IT <= 16;  // both banks in SGPRs (2 x 32 of the 96)
  static_assert(TAPS_IT <= 30, "one bank of taps must fit the SGPRs");
  double ring[RING];
#pragma unroll
  for (int j = 0; j < RING; j++) ring[j] = 0.0;
  const double *__restrict__ trow = rows;  // wave-uniform, __restrict__ kernel argument -> s_load
  uint32_t n_it = SPEEXHIP_DIAG_SKIP(p, 4u) ? 0 : p.row_len / U;  // even (plan_slide64)
  if (p.parts > 1) {
    const uint32_t pairs = n_it / 2;
    const uint32_t it0 = pairs * part / p.parts * 2, it1 = pairs * (part + 1) / p.parts * 2;
    trow += static_cast<size_t>(it0) * TAPS_IT;
    xrow += static_cast<size_t>(it0) * p.row_stride;
    n_it = it1 - it0;
  }
  double tpa[TAPS_IT], tpb[TAP2 ? TAPS_IT : 1];
  auto load_taps = [&](double (&t)[TAPS_IT], const double *tr) {
#pragma unroll
    for (int j = 0; j < TAPS_IT; j++) t[j] = tr[j];
  };
  if (n_it != 0) {
    static_for64<0, OLD>([&](auto j) { ring[decltype(j)::value] = static_cast<double>(xrow[decltype(j)::value * C]); });
    if constexpr (TAP2) load_taps(tpa, trow);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the loop is entered with nothing in flight
  }
  // one iteration; BASE = (it % 2) * U: window sample k is ring[(BASE + k) % RING]
  auto iteration = [&](auto base_c, double (&tp)[TAPS_IT], auto &tp_next) {
    constexpr int BASE = decltype(base_c)::value;
    if constexpr (!TAP2) load_taps(tp, trow);
    E raw[U];
    static_for64<OLD, W>([&](auto k_c) {  // the U new samples
      constexpr int k = decltype(k_c)::value;
      raw[k - OLD] = k < U ? xrow[k * C] : xrow[p.row_stride + (k - U) * C];
    });
    if constexpr (TAP2) load_taps(tp_next, trow + TAPS_IT);
    auto fmas = [&](bool new_samples) {
#pragma unroll
      for (int s = 0; s < U; s++)
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          if ((pp * NUM + s >= OLD) != new_samples) continue;
#pragma unroll
          for (int r = 0; r < DEN; r++) fma64(acc[pp][r], tp[s * DEN + r], ring[(BASE + pp * NUM + s) % RING]);
        }
    };
    if constexpr (CARRY) __builtin_amdgcn_sched_barrier(0);
    fmas(false);  // old samples only: runs while the loads above are in flight
    if constexpr (CARRY) __builtin_amdgcn_sched_barrier(0);
    static_for64<OLD, W>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      ring[(BASE + k) % RING] = static_cast<double>(raw[k - OLD]);
    });
    fmas(true);
    if constexpr (CARRY) __builtin_amdgcn_sched_barrier(0);
    trow += TAPS_IT;
    xrow += p.row_stride;
  };
  if constexpr (!CARRY) {
    for (uint32_t it = 0; it < n_it; it++) iteration(std::integral_constant<int, 0>(), tpa, tpa);
  } else {
    for (uint32_t it = 0; it < n_it; it += 2) {
      if constexpr (TAP2) {
        iteration(std::integral_constant<int, 0>(), tpa, tpb);
        iteration(std::integral_constant<int, U>(), tpb, tpa);
      } else {
        iteration(std::integral_constant<int, 0>(), tpa, tpa);
        iteration(std::integral_constant<int, U>(), tpa, tpa);
      }
    }
  }
  if (p.parts > 1) {
    // sums of set j >= 1, wave w: block (j - 1) * base_waves + w of P x DEN x 64 doubles, lanes side by side
    __syncthreads();  // every wave is done with the window
    double *sums = reinterpret_cast<double *>(xs);
    if (part != 0) {
      double *mine = sums + (static_cast<size_t>(part - 1) * p.base_waves + wave) * (P * DEN * 64) + lane;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int r = 0; r < DEN; r++) mine[(pp * DEN + r) * 64] = acc[pp][r];
    }
    __syncthreads();
    if (part != 0) return;
    for (uint32_t j = 1; j < p.parts; j++) {
      const double *theirs = sums + (static_cast<size_t>(j - 1) * p.base_waves + wave) * (P * DEN * 64) + lane;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int r = 0; r < DEN; r++) acc[pp][r] += theirs[(pp * DEN + r) * 64];
    }
  }
  if (SPEEXHIP_DIAG_SKIP(p, 8u)) return;

  // ---- fp64 -> fp32 (the reference stores its double sum into a float, resample.c:417 / :544) -> round,
  //      interleave, store: N = P*DEN consecutive output frames of this lane's channel --------------------
  constexpr int N = P * DEN;
  float v[N];
#pragma unroll
  for (int pp = 0; pp < P; pp++)
#pragma unroll
    for (int r = 0; r < DEN; r++) v[pp * DEN + r] = static_cast<float>(acc[pp][r]);
  const uint64_t K0 = static_cast<uint64_t>(m_lo + min(lb, p.blocks_per_tile - 1) * P) * p.den;
  const bool inside = K0 >= d.k_shift && K0 + static_cast<uint64_t>(N) <= K_end;
  const int64_t f0 = static_cast<int64_t>(K0) - static_cast<int64_t>(d.k_shift);  // first frame of the run in the call
  if constexpr (sizeof(T) == 4) {
    if (!lane_live) return;
    G<float> *o0 = out_ptr<float>(d) + f0 * C + c;
    if (C == 1 && inside) {  // N consecutive floats
#pragma unroll
      for (int q = 0; q + 4 <= N; q += 4) *(G<f32x4_a4> *)(o0 + q) = f32x4_a4{v[q], v[q + 1], v[q + 2], v[q + 3]};
      if constexpr (N % 4 >= 2) *(G<f32x2_a4> *)(o0 + N / 4 * 4) = f32x2_a4{v[N / 4 * 4], v[N / 4 * 4 + 1]};
      if constexpr (N % 2 != 0) o0[N - 1] = v[N - 1];
      return;
    }
#pragma unroll
    for (int q = 0; q < N; q++) {
      const uint64_t K = K0 + q;
      if (K < d.k_shift || K >= K_end) continue;
      o0[static_cast<int64_t>(q) * C] = v[q];
    }
    return;
  } else {
    g_i16 *o0 = out_ptr<int16_t>(d) + f0 * C + c;
    if constexpr (N % 2 == 0) {
      if (C == 2) {
        // stereo: lanes 2b (left) and 2b + 1 (right) hold the two channels of the same N frames.  Each rounds its
        // own, the pair swaps halves (one DPP move per frame), and each stores N/2 whole frames: the left lane the
        // first half of the run, the right lane the second.  (Every lane of the wave takes part in the swap: the
        // exits of dead lanes come after it.)
        int mine[N];
#pragma unroll
        for (int q = 0; q < N; q++) asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(mine[q]) : "v"(v[q]));
        const bool right = (lane & 1u) != 0;
        uint32_t w[N / 2];
#pragma unroll
        for (int q = 0; q < N / 2; q++) {
          const int theirs = pair_swap(right ? mine[q] : mine[N / 2 + q]);
          typedef short short2_t __attribute__((ext_vector_type(2)));
          const short2_t pk = right ? __builtin_amdgcn_cvt_pk_i16(theirs, mine[N / 2 + q])
                                    : __builtin_amdgcn_cvt_pk_i16(mine[q], theirs);
          w[q] = __builtin_bit_cast(uint32_t, pk);
        }
        if (!lane_live) return;
        g_i16 *of = out_ptr<int16_t>(d) + (f0 + (right ? N / 2 : 0)) * 2;  // first frame this lane stores
        if (inside && (reinterpret_cast<uintptr_t>(of) & 3u) == 0) {
          g_u32 *od = (g_u32 *)of;
#pragma unroll
          for (int q = 0; q + 4 <= N / 2; q += 4) *(g_u32x4_a4 *)(od + q) = u32x4_a4{w[q], w[q + 1], w[q + 2], w[q + 3]};
          if constexpr ((N / 2) % 4 >= 2) *(g_u32x2_a4 *)(od + (N / 2) / 4 * 4) = u32x2_a4{w[(N / 2) / 4 * 4], w[(N / 2) / 4 * 4 + 1]};
          if constexpr ((N / 2) % 2 != 0) od[N / 2 - 1] = w[N / 2 - 1];
          return;
        }
#pragma unroll
        for (int q = 0; q < N / 2; q++) {
          const uint64_t K = K0 + (right ? N / 2 : 0) + q;
          if (K < d.k_shift || K >= K_end) continue;
          of[2 * q] = static_cast<int16_t>(w[q] & 0xffffu);
          of[2 * q + 1] = static_cast<int16_t>(w[q] >> 16);
        }
        return;
      }
    }
    if (!lane_live) return;
    if constexpr (N % 2 == 0) {
      if (C == 1 && inside) {  // mono: N consecutive samples as whole dwords, at either alignment
        uint32_t w[N / 2];
#pragma unroll
        for (int q = 0; q < N / 2; q++) w[q] = round_pack_pcm(v[2 * q], v[2 * q + 1]);
        if ((reinterpret_cast<uintptr_t>(o0) & 3u) == 0) {
          g_u32 *od = (g_u32 *)o0;
#pragma unroll
          for (int q = 0; q + 4 <= N / 2; q += 4) *(g_u32x4_a4 *)(od + q) = u32x4_a4{w[q], w[q + 1], w[q + 2], w[q + 3]};
          if constexpr ((N / 2) % 4 >= 2) *(g_u32x2_a4 *)(od + (N / 2) / 4 * 4) = u32x2_a4{w[(N / 2) / 4 * 4], w[(N / 2) / 4 * 4 + 1]};
          if constexpr ((N / 2) % 2 != 0) od[N / 2 - 1] = w[N / 2 - 1];
          return;
        }
        if ((reinterpret_cast<uintptr_t>(o0) & 3u) == 2) {
          // the run starts on the upper half of a dword (k_shift odd): one sample, the dwords that straddle the
          // pairs (v_alignbit), one sample
          constexpr int M = N / 2 - 1;
          o0[0] = static_cast<int16_t>(w[0] & 0xffffu);
          if constexpr (M > 0) {
            uint32_t s[M];
#pragma unroll
            for (int q = 0; q < M; q++) s[q] = __builtin_amdgcn_alignbit(w[q + 1], w[q], 16);
            g_u32 *od = (g_u32 *)(o0 + 1);
#pragma unroll
            for (int q = 0; q + 4 <= M; q += 4) *(g_u32x4_a4 *)(od + q) = u32x4_a4{s[q], s[q + 1], s[q + 2], s[q + 3]};
            if constexpr (M % 4 >= 2) *(g_u32x2_a4 *)(od + M / 4 * 4) = u32x2_a4{s[M / 4 * 4], s[M / 4 * 4 + 1]};
            if constexpr (M % 2 != 0) od[M - 1] = s[M - 1];
          }
          o0[N - 1] = static_cast<int16_t>(w[N / 2 - 1] >> 16);
          return;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < N; q++) {
      const uint64_t K = K0 + q;
      if (K < d.k_shift || K >= K_end) continue;
      o0[static_cast<int64_t>(q) * C] = static_cast<int16_t>(round_pack_pcm(v[q], 0.f) & 0xffffu);
    }
  }
}

template <int P, int NUM, int DEN, int CH, typename T>
hipError_t launch_s64(const SlideParams &p, const double *rows, const DescPack *pack, dim3 grid, uint32_t threads,
                      size_t lds_bytes, hipStream_t stream) {
  static std::atomic<uint64_t> seen{0};
  opt_in_lds_on_this_device(resample_slide64<P, NUM, DEN, CH, T>, seen);
  hipLaunchKernelGGL((resample_slide64<P, NUM, DEN, CH, T>), grid, dim3(threads), lds_bytes, stream, p, rows, *pack);
  return hipGetLastError();
}

}  // namespace

// the instantiation table: (periods per lane, num, den) as kShapes64 lists them (kernels_slide.hip)
template <typename T>
hipError_t launch_slide64_shape(const SlidePlan &t, const SlideParams &p, const double *rows, const DescPack *pack, dim3 grid, uint32_t threads, size_t lds, hipStream_t stream) {
#define SPEEXHIP_S64_CASE(PP, NUMV, DENV)                                                                             \
  if (t.p == PP && t.num == NUMV && t.np == DENV) {                                                                  \
    if (t.cgroups == 1) return launch_s64<PP, NUMV, DENV, 1, T>(p, rows, pack, grid, threads, lds, stream);          \
    if (t.cgroups == 2) return launch_s64<PP, NUMV, DENV, 2, T>(p, rows, pack, grid, threads, lds, stream);          \
    return launch_s64<PP, NUMV, DENV, 0, T>(p, rows, pack, grid, threads, lds, stream);                              \
  }
  SPEEXHIP_S64_CASE(8, 1, 1)
  SPEEXHIP_S64_CASE(8, 1, 2)
  SPEEXHIP_S64_CASE(8, 1, 3)
  SPEEXHIP_S64_CASE(4, 1, 4)
  SPEEXHIP_S64_CASE(4, 1, 5)
  SPEEXHIP_S64_CASE(4, 1, 6)
  SPEEXHIP_S64_CASE(8, 2, 1)
  SPEEXHIP_S64_CASE(4, 2, 3)
  SPEEXHIP_S64_CASE(2, 2, 5)
  SPEEXHIP_S64_CASE(4, 3, 1)
  SPEEXHIP_S64_CASE(4, 3, 2)
  SPEEXHIP_S64_CASE(2, 3, 5)
  SPEEXHIP_S64_CASE(4, 4, 1)
  SPEEXHIP_S64_CASE(1, 4, 5)
  SPEEXHIP_S64_CASE(4, 5, 1)
  SPEEXHIP_S64_CASE(2, 5, 2)
  SPEEXHIP_S64_CASE(2, 5, 3)
  SPEEXHIP_S64_CASE(1, 5, 4)
  SPEEXHIP_S64_CASE(1, 5, 6)
  SPEEXHIP_S64_CASE(2, 6, 1)
  SPEEXHIP_S64_CASE(1, 6, 5)
  SPEEXHIP_S64_CASE(2, 7, 1)
  SPEEXHIP_S64_CASE(2, 8, 1)
  SPEEXHIP_S64_CASE(1, 8, 3)
  SPEEXHIP_S64_CASE(2, 9, 1)
  SPEEXHIP_S64_CASE(2, 10, 1)
  SPEEXHIP_S64_CASE(1, 12, 1)
  SPEEXHIP_S64_CASE(1, 16, 1)
  SPEEXHIP_S64_CASE(1, 20, 1)
  SPEEXHIP_S64_CASE(1, 24, 1)
  SPEEXHIP_S64_CASE(2, 7, 2)
  SPEEXHIP_S64_CASE(1, 9, 2)
#undef SPEEXHIP_S64_CASE
  return hipErrorInvalidValue;
}

}  // namespace speexhip
