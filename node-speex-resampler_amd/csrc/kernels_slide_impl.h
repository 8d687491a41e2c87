#pragma once
// kernels_slide_impl.h -- (included by kernels_slide_i16.hip / kernels_slide_f32.hip: one translation unit per
// sample type, so that the two halves of the instantiations compile side by side)
// kernels_slide.hip -- fast gfx950 kernel for small rational ratios (den <= 6 with num <= 6, 8:3, and n:1 for
// n <= 10, 12, 16, 20, 24):
// integer up-sampling 24k->48k, 16k->48k, 8k->48k, same-rate, 2:1 / 3:1 / 4:1 decimation, 3:2,
// 2:3 ... (BASELINE configs[2]; the reference picks resampler_basic_direct_* for
// most of these, deps/speex/resample.c:331-435).  +-1 LSB.
//
// Output K = m*den + r reads V[base + m*num + delta_r + s] for s < taps.  Consecutive periods m
// slide over the input by only `num` frames, so a lane that owns P consecutive periods needs,
// for U tap steps, just (P-1)*num + U input frames for P*den*U multiply-adds:
//   lane  = block of P consecutive periods (x one channel pair): P*den accumulator pairs and a
//           register window of (2P-1)*num frames re-read from LDS once per iteration
//           (U = P*num steps, so the window advances exactly one LDS row per iteration);
//   taps  = wave-uniform (every lane is at the same step): scalar loads -> SGPR operands of
//           v_pk_fma_f32, U*den taps per iteration; the rows of phase r are pre-shifted by
//           delta_r = (r*num) div den so all phases of a period read the same sample per step;
//   packing: even channel count -> one packed FMA = both channels of a frame (tap broadcast);
//            odd  channel count -> one packed FMA = two phases of one sample (sample broadcast),
//            den padded to even with a zero phase.
//   LDS   = the tile's input as float in rows of P*num frames, one row per lane, row stride
//           padded so that the 64 lanes of a wave hit distinct banks.
//   out   = each lane owns P*den consecutive output frames: contiguous wide stores.
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned wide store
typedef __attribute__((address_space(1))) u32x4_a4 g_u32x4_a4;
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [LO, HI)
template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>());
    static_for<LO + 1, HI>(f);
  }
}

// acc.xy += tap[hi].xx * x.xy   (channel-pair packing: tap broadcast from an SGPR pair)
__device__ __forceinline__ void fma_bcast_tap(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x, bool hi) {
  if (hi)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}
// acc.xy += tap.xy * x.xx (or x.yy)   (phase-pair packing: sample broadcast from one half of a pair of
// consecutive frames)
__device__ __forceinline__ void fma_bcast_x(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x, bool hi) {
  if (hi)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}

// PAIR_CH: true = channel pairs (NP = den accumulators per period), false = phase pairs (NP = ceil(den/2)).
// P: periods per lane; NUM: input frames per period; U = P*NUM tap steps per iteration.
// DENSE: the frame is exactly one lane's samples (mono for phase pairs, stereo for channel pairs): the
// channel count is a compile-time constant and every LDS offset of the FIR loop an immediate.
template <int P, int NUM, int NP, bool PAIR_CH, bool DENSE, typename T>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(96))) void resample_slide(
    SlideParams p, const float *__restrict__ rows, DescPack pack) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const StreamDesc d = pack.d[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {
    roll_history<T>(p.channels, d, p.threads);
    return;
  }
  if (d.n_out == 0) return;
  constexpr int CW = PAIR_CH ? 2 : 1;          // floats a lane reads per frame
  const uint32_t C = DENSE ? static_cast<uint32_t>(CW) : p.channels;
  const uint32_t K_end = d.k_shift + d.n_out;
  const uint32_t m_total = d.m_total;  // ceil(K_end / den), from the host (round 3)
  const uint32_t tile_periods = p.blocks_per_tile * P;  // lane blocks x P
  const uint32_t m_lo = blockIdx.x * tile_periods;
  if (m_lo >= m_total) return;
  const uint32_t m_cnt = min(tile_periods, m_total - m_lo);

  // ---- stage: frames [f0, f0 + m_cnt*num + row_len + one row) of V as float, in rows of P*NUM
  //      frames `row_stride` floats apart: the shared loader's padded image with the row as its
  //      padding period (16-byte loads, all in flight at once; device_helpers.h) ----
  // (tiles inside the call's input -- all but the first and last few -- get their geometry from a dozen scalar
  //  instructions: device_helpers.h, window_geom_plain)
  WindowGeom wg;
  if (!window_geom_plain<T>(d, C, NUM, NUM + p.row_len + P * NUM, m_lo, m_cnt, p.threads, p.row_stride - P * NUM * C,
                            p.row_magic, &wg, P * NUM * C))
    wg = window_geom<T>(d, C, NUM, NUM + p.row_len + P * NUM, m_lo, m_cnt, p.threads, p.row_stride - P * NUM * C,
                        p.row_magic, P * NUM * C);
  if (!SPEEXHIP_DIAG_SKIP(p, 2u)) {
    u32x4 w[4];
    window_fetch<4, T>(wg, w);
    window_commit<4, T>(xs, d, wg, w);
  }
  __syncthreads();

  // Tap-range parts (p.parts > 1; small launches of long filters, launch_slide): the workgroup's waves come in
  // `parts` sets of p.base_waves; set j runs iterations [it0, it1) of every lane block and the sums meet in LDS.
  uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t part = 0;
  while (wave >= p.base_waves) {  // (wave-uniform)
    wave -= p.base_waves;
    part++;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t cg = lane % p.cgroups;        // channel pair (PAIR_CH) or channel (phase pairs)
  const uint32_t lb = wave * p.blocks_per_wave + lane / p.cgroups;  // lane block inside the tile
  const bool lane_live = (lane / p.cgroups) < p.blocks_per_wave && lb * P < m_cnt;
  const float *xrow = xs + wg.xshift + min(lb, p.blocks_per_tile - 1) * p.row_stride + cg * CW;

  f32x2 acc[P][NP];
#pragma unroll
  for (int pp = 0; pp < P; pp++)
#pragma unroll
    for (int r = 0; r < NP; r++) acc[pp][r] = f32x2{0.f, 0.f};

  constexpr int U = P * NUM;                               // tap steps per iteration
  constexpr int TAPS_IT = PAIR_CH ? U * NP : U * NP * 2;   // tap floats per iteration
  constexpr int TP = TAPS_IT / 2;                         // ... as SGPR pairs
  static_assert(TAPS_IT % 2 == 0, "tap floats per iteration must pair up");
  // The window of iteration `it` is frames [it*U, it*U + W) of the lane's block, W = (2P-1)*NUM: its first
  // OLD = W - U frames were the last OLD of the previous iteration's.  They stay in registers -- frame f
  // lives in ring[f mod 2U], two copies of the loop body (even / odd iterations) name the registers -- and
  // an iteration reads only its U new frames from LDS: the NUM last of its own row and the first OLD of
  // the next one (8 reads instead of 15 for P = 8, NUM = 1; the loop used to be LDS-bound at 4.3 FMAs per
  // read).  The FMAs that touch only old frames (28 of 64 for P = 8, NUM = 1) are issued before the single
  // wait of the iteration, behind the LDS reads and behind the scalar loads of the NEXT iteration's taps
  // (two tap banks when they fit the SGPRs): before, every iteration began by waiting for its own taps
  // and samples.  host: row_len is a multiple of 2U (plan_slide), rows carry one iteration of zero taps
  // past the end (build_slide_rows).
  constexpr int W = (2 * P - 1) * NUM, OLD = W - U;
  constexpr int RING = OLD > 0 ? 2 * U : U;
  // (one period per lane -- OLD = 0, the 16:1 ... 24:1 shapes -- has nothing to carry over and nothing to
  //  run ahead of its loads: there the compiler's own interleaving of reads, partial waits and FMAs is
  //  faster than one wait per iteration, 142 vs 175 us for 32 streams of 192k -> 8k)
  constexpr bool CARRY = OLD > 0;
  constexpr bool TAP2 = CARRY && TAPS_IT <= 32;  // both banks in SGPRs (96 in all)
  // channel pairs: ring[f] = the frame's two channels; phase pairs (one channel per lane): ring[f / 2]
  // holds frames f and f + 1 -- the FMA broadcasts either half -- so two frames arrive per LDS read
  constexpr int RING_REGS = PAIR_CH ? RING : (RING + 1) / 2;
  f32x2 ring[RING_REGS];
#pragma unroll
  for (int j = 0; j < RING_REGS; j++) ring[j] = f32x2{0.f, 0.f};
  auto load_frame = [&](f32x2 *win, auto idx_c, const float *px) {
    constexpr int IDX = decltype(idx_c)::value;
    if constexpr (PAIR_CH)
      win[IDX] = *reinterpret_cast<const f32x2 *>(px);
    else if constexpr (IDX % 2 == 0)
      win[IDX / 2].x = *px;
    else
      win[IDX / 2].y = *px;
  };
  const float *__restrict__ trow = rows;  // wave-uniform, __restrict__ kernel argument -> s_load
  uint32_t n_it = SPEEXHIP_DIAG_SKIP(p, 4u) ? 0 : p.row_len / U;  // even
  if (p.parts > 1) {  // this set's range of iterations, on even bounds (the loop runs them in pairs)
    const uint32_t pairs = n_it / 2;
    const uint32_t it0 = pairs * part / p.parts * 2, it1 = pairs * (part + 1) / p.parts * 2;
    trow += static_cast<size_t>(it0) * TAPS_IT;
    xrow += static_cast<size_t>(it0) * p.row_stride;
    n_it = it1 - it0;
  }
  f32x2 tpa[TP], tpb[TAP2 ? TP : 1];
  auto load_taps = [&](f32x2 (&t)[TP], const float *tr) {
#pragma unroll
    for (int j = 0; j < TP; j++) t[j] = *reinterpret_cast<const f32x2 *>(tr + 2 * j);
  };
  if (n_it != 0) {
    static_for<0, OLD>([&](auto j) { load_frame(ring, j, xrow + decltype(j)::value * C); });  // frames [0, OLD) of row 0
    if constexpr (TAP2) load_taps(tpa, trow);
    // (the loop is entered with nothing in flight: otherwise hipcc waits at the top of every iteration)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  }
  // one iteration; BASE = (it % 2) * U: window frame k is ring[(BASE + k) % RING]
  auto iteration = [&](auto base_c, f32x2 (&tp)[TP], auto &tp_next) {
    constexpr int BASE = decltype(base_c)::value;
    if constexpr (!TAP2) load_taps(tp, trow);
    // (nothing carried over: a window of its own per iteration, so that the next one's reads need not wait
    //  for this one's FMAs to release the registers)
    f32x2 fresh[CARRY ? 1 : RING_REGS];
    f32x2 *win = CARRY ? ring : fresh;
    static_for<OLD, W>([&](auto k_c) {  // the U new frames
      constexpr int k = decltype(k_c)::value;
      load_frame(win, std::integral_constant<int, (BASE + k) % RING>(),
                 k < U ? xrow + k * C : xrow + p.row_stride + (k - U) * C);
    });
    if constexpr (TAP2) load_taps(tp_next, trow + TAPS_IT);
    // window frame k sits in a register this iteration loads: it is new itself, or (phase pairs) the
    // frame beside it in its register pair is
    auto loaded = [](int k) {
      if (k >= OLD) return true;
      if (PAIR_CH) return false;
      const int mate = (((BASE + k) % RING) ^ 1), km = (mate - BASE + RING) % RING;
      return km >= OLD && km < W;
    };
    auto fmas = [&](bool new_frames) {
#pragma unroll
      for (int s = 0; s < U; s++)
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          if (loaded(pp * NUM + s) != new_frames) continue;
#pragma unroll
          for (int r = 0; r < NP; r++) {
            const int f = (BASE + pp * NUM + s) % RING;
            if (PAIR_CH) {
              const int k = s * NP + r;
              fma_bcast_tap(acc[pp][r], tp[k >> 1], win[f], (k & 1) != 0);
            } else {
              fma_bcast_x(acc[pp][r], tp[s * NP + r], win[f / 2], (f & 1) != 0);
            }
          }
        }
    };
    if constexpr (CARRY) __builtin_amdgcn_sched_barrier(0);
    fmas(false);  // old frames only: runs while the loads above are in flight
    if constexpr (CARRY) __builtin_amdgcn_sched_barrier(0);
    fmas(true);
    if constexpr (CARRY) __builtin_amdgcn_sched_barrier(0);
    trow += TAPS_IT;
    xrow += p.row_stride;
  };
  if constexpr (!CARRY && PAIR_CH) {
    // one period per lane (16:1 ... 24:1, 6:5, 5:6) on channel pairs: nothing to carry over.  The plain loop -- taps, the whole
    // window, FMAs, register-held offsets even when the frame is dense -- measured faster here than the
    // carry loop's machinery with nothing to carry (32 streams of 192k -> 8k: 141 vs 163 us stereo, 218 vs
    // 342 us on 4 channels), so it stays as it was.  (Phase pairs keep the new loop, one copy per trip: two
    // frames per LDS read and no zeroing of unused halves -- 192k -> 8k mono 134 -> 92 us.)
    const uint32_t Cr = p.channels;
    for (uint32_t it = 0; it < n_it; it++, trow += TAPS_IT, xrow += p.row_stride) {
      f32x2 tp[TP];
#pragma unroll
      for (int j = 0; j < TP; j++) tp[j] = *reinterpret_cast<const f32x2 *>(trow + 2 * j);
      f32x2 xw[W];
#pragma unroll
      for (int j = 0; j < W; j++) {
        const float *px = xrow + (j < U ? j * Cr : p.row_stride + (j - U) * Cr);
        if (PAIR_CH) {
          xw[j] = *reinterpret_cast<const f32x2 *>(px);
        } else {
          xw[j].x = *px;
          xw[j].y = 0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < U; s++)
#pragma unroll
        for (int pp = 0; pp < P; pp++)
#pragma unroll
          for (int r = 0; r < NP; r++) {
            if (PAIR_CH) {
              const int k = s * NP + r;
              fma_bcast_tap(acc[pp][r], tp[k >> 1], xw[pp * NUM + s], (k & 1) != 0);
            } else {
              fma_bcast_x(acc[pp][r], tp[s * NP + r], xw[pp * NUM + s], false);
            }
          }
    }
  } else if constexpr (!CARRY) {
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
    // sums of set j >= 1, wave w: block (j - 1) * base_waves + w of P x NP x 64 pairs, lanes side by side
    __syncthreads();  // every wave is done with the window
    f32x2 *sums = reinterpret_cast<f32x2 *>(xs);
    if (part != 0) {
      f32x2 *mine = sums + (static_cast<size_t>(part - 1) * p.base_waves + wave) * (P * NP * 64) + lane;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int r = 0; r < NP; r++) mine[(pp * NP + r) * 64] = acc[pp][r];
    }
    __syncthreads();
    if (part != 0) return;
    for (uint32_t j = 1; j < p.parts; j++) {
      const f32x2 *theirs = sums + (static_cast<size_t>(j - 1) * p.base_waves + wave) * (P * NP * 64) + lane;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int r = 0; r < NP; r++) acc[pp][r] += theirs[(pp * NP + r) * 64];
    }
  }
  if (!lane_live || SPEEXHIP_DIAG_SKIP(p, 8u)) return;

  // ---- round, interleave, store: P*den consecutive output frames of this lane -----------------
  const uint64_t K0 = static_cast<uint64_t>(m_lo + lb * P) * p.den;
  const bool inside = K0 >= d.k_shift && K0 + static_cast<uint64_t>(P) * p.den <= K_end;
  if constexpr (sizeof(T) == 4) {
    // float I/O (resample.c:927-963): the FIR values as they are
    G<float> *o0 = out_ptr<float>(d) + (static_cast<int64_t>(K0) - static_cast<int64_t>(d.k_shift)) * C;
    const bool dense = PAIR_CH ? (C == 2) : (C == 1 && p.den == 2u * NP);
    if (dense && inside) {  // the lane's P*NP pairs are 2*P*NP consecutive floats
#pragma unroll
      for (int q = 0; q + 1 < P * NP; q += 2) {
        const f32x2 a = acc[q / NP][q % NP], b = acc[(q + 1) / NP][(q + 1) % NP];
        *(G<f32x4_a4> *)(o0 + 2 * q) = f32x4_a4{a.x, a.y, b.x, b.y};
      }
      if constexpr ((P * NP) % 2 != 0) {  // (one period per lane, one pair per period: the n:1 shapes)
        typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
        *(G<f32x2_a4> *)(o0 + 2 * (P * NP - 1)) = f32x2_a4{acc[P - 1][NP - 1].x, acc[P - 1][NP - 1].y};
      }
      return;
    }
#pragma unroll
    for (int pp = 0; pp < P; pp++)
#pragma unroll
      for (int r = 0; r < NP; r++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          // channel pairs: h = channel of the pair, phase r; phase pairs: phase 2r + h
          const uint32_t ph = PAIR_CH ? r : 2 * r + h;
          const uint64_t K = K0 + static_cast<uint64_t>(pp) * p.den + ph;
          if (ph >= p.den || K < d.k_shift || K >= K_end) continue;
          out_ptr<float>(d)[(K - d.k_shift) * C + (PAIR_CH ? cg * 2 + h : cg)] = h ? acc[pp][r].y : acc[pp][r].x;
        }
      }
    return;
  } else {
  uint32_t v[P * NP];  // packed s16 pairs in output order
#pragma unroll
  for (int pp = 0; pp < P; pp++)
#pragma unroll
    for (int r = 0; r < NP; r++) v[pp * NP + r] = round_pack_pcm(acc[pp][r].x, acc[pp][r].y);
  // The lane's pairs are consecutive dwords of the output when a frame is exactly one pair
  // (stereo) or phase pairs tile a mono period: wide dword-aligned stores, 16 bytes at a time.
  const bool dense = PAIR_CH ? (C == 2) : (C == 1 && p.den == 2u * NP);
  g_i16 *o0 = out_ptr<int16_t>(d) + (static_cast<int64_t>(K0) - static_cast<int64_t>(d.k_shift)) * C;
  if (dense && inside && (reinterpret_cast<uintptr_t>(o0) & 3u) == 0) {
    if constexpr ((P * NP) % 4 == 0) {
#pragma unroll
      for (int q = 0; q < P * NP; q += 4)
        *(g_u32x4_a4 *)(o0 + 2 * q) = u32x4_a4{v[q], v[q + 1], v[q + 2], v[q + 3]};
    } else {
      typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
      typedef __attribute__((address_space(1))) u32x2_a4 g_u32x2_a4;
#pragma unroll
      for (int q = 0; q + 1 < P * NP; q += 2) *(g_u32x2_a4 *)(o0 + 2 * q) = u32x2_a4{v[q], v[q + 1]};
      if constexpr ((P * NP) % 2 != 0) *(G<uint32_t> *)(o0 + 2 * (P * NP - 1)) = v[P * NP - 1];
    }
    return;
  }
  if constexpr (!PAIR_CH && P * NP >= 2) {
    // mono, the run starts on the upper half of a dword (k_shift odd): one sample, the dwords that straddle
    // the pairs (v_alignbit), one sample -- not 2*P*NP stores of 2 bytes
    if (dense && inside && (reinterpret_cast<uintptr_t>(o0) & 3u) == 2) {
      constexpr int N = P * NP - 1;
      uint32_t s[N];
#pragma unroll
      for (int q = 0; q < N; q++) s[q] = __builtin_amdgcn_alignbit(v[q + 1], v[q], 16);
      o0[0] = static_cast<int16_t>(v[0] & 0xffffu);
      G<uint32_t> *od = (G<uint32_t> *)(o0 + 1);
      typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
      typedef __attribute__((address_space(1))) u32x2_a4 g_u32x2_a4;
#pragma unroll
      for (int q = 0; q + 4 <= N; q += 4) *(g_u32x4_a4 *)(od + q) = u32x4_a4{s[q], s[q + 1], s[q + 2], s[q + 3]};
      if constexpr (N % 4 >= 2) *(g_u32x2_a4 *)(od + N / 4 * 4) = u32x2_a4{s[N / 4 * 4], s[N / 4 * 4 + 1]};
      if constexpr (N % 2 != 0) od[N - 1] = s[N - 1];
      o0[2 * P * NP - 1] = static_cast<int16_t>(v[P * NP - 1] >> 16);
      return;
    }
  }
#pragma unroll
  for (int pp = 0; pp < P; pp++) {
#pragma unroll
    for (int r = 0; r < NP; r++) {
      const uint32_t w = v[pp * NP + r];
      if (PAIR_CH) {
        const uint64_t K = K0 + static_cast<uint64_t>(pp) * p.den + r;
        if (K < d.k_shift || K >= K_end) continue;
        g_i16 *o = out_ptr<int16_t>(d) + (K - d.k_shift) * C + cg * 2;
        o[0] = static_cast<int16_t>(w & 0xffffu);
        o[1] = static_cast<int16_t>(w >> 16);
      } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint32_t ph = 2 * r + h;
          const uint64_t K = K0 + static_cast<uint64_t>(pp) * p.den + ph;
          if (ph >= p.den || K < d.k_shift || K >= K_end) continue;
          out_ptr<int16_t>(d)[(K - d.k_shift) * C + cg] = static_cast<int16_t>(h ? (w >> 16) : (w & 0xffffu));
        }
      }
    }
  }
  }
}

template <int P, int NUM, int NP, bool PAIR_CH, bool DENSE, typename T>
hipError_t launch_up(const SlideParams &p, const DescPack *pack, dim3 grid, uint32_t threads, size_t lds_bytes,
                     hipStream_t stream) {
  static std::atomic<uint64_t> seen{0};
  opt_in_lds_on_this_device(resample_slide<P, NUM, NP, PAIR_CH, DENSE, T>, seen);
  hipLaunchKernelGGL((resample_slide<P, NUM, NP, PAIR_CH, DENSE, T>), grid, dim3(threads), lds_bytes, stream, p, p.rows, *pack);
  return hipGetLastError();
}

}  // namespace

// the instantiation table: one launch per (periods per lane, num, accumulator pairs, packing) x dense / strided
template <typename T>
hipError_t launch_slide_shape(const SlidePlan &t, const SlideParams &p, const DescPack *pack,
                              dim3 grid, uint32_t threads, size_t lds, hipStream_t stream) {
#define SPEEXHIP_SLIDE_CASE(PP, NUMV, NPV, CHV)                                                                       \
  if (t.p == PP && t.num == NUMV && t.np == NPV && t.pair_ch == CHV) {                                                \
    if (t.cgroups == 1) return launch_up<PP, NUMV, NPV, CHV, true, T>(p, pack, grid, threads, lds, stream);   \
    return launch_up<PP, NUMV, NPV, CHV, false, T>(p, pack, grid, threads, lds, stream);                     \
  }
  SPEEXHIP_SLIDE_CASE(8, 1, 1, true)
  SPEEXHIP_SLIDE_CASE(8, 1, 2, true)
  SPEEXHIP_SLIDE_CASE(8, 1, 3, true)
  SPEEXHIP_SLIDE_CASE(4, 1, 4, true)
  SPEEXHIP_SLIDE_CASE(4, 1, 6, true)
  SPEEXHIP_SLIDE_CASE(8, 1, 1, false)
  SPEEXHIP_SLIDE_CASE(8, 1, 2, false)
  SPEEXHIP_SLIDE_CASE(4, 1, 3, false)
  SPEEXHIP_SLIDE_CASE(8, 2, 1, true)
  SPEEXHIP_SLIDE_CASE(4, 2, 3, true)
  SPEEXHIP_SLIDE_CASE(8, 2, 1, false)
  SPEEXHIP_SLIDE_CASE(4, 2, 2, false)
  SPEEXHIP_SLIDE_CASE(4, 3, 1, true)
  SPEEXHIP_SLIDE_CASE(4, 3, 2, true)
  SPEEXHIP_SLIDE_CASE(4, 3, 1, false)
  SPEEXHIP_SLIDE_CASE(4, 4, 1, true)
  SPEEXHIP_SLIDE_CASE(4, 4, 1, false)
  SPEEXHIP_SLIDE_CASE(4, 5, 1, true)
  SPEEXHIP_SLIDE_CASE(4, 5, 1, false)
  SPEEXHIP_SLIDE_CASE(4, 6, 1, true)
  SPEEXHIP_SLIDE_CASE(4, 6, 1, false)
  SPEEXHIP_SLIDE_CASE(2, 8, 1, true)
  SPEEXHIP_SLIDE_CASE(2, 8, 1, false)
  SPEEXHIP_SLIDE_CASE(2, 12, 1, true)
  SPEEXHIP_SLIDE_CASE(2, 12, 1, false)
  SPEEXHIP_SLIDE_CASE(4, 1, 5, true)
  SPEEXHIP_SLIDE_CASE(4, 2, 5, true)
  SPEEXHIP_SLIDE_CASE(2, 3, 5, true)
  SPEEXHIP_SLIDE_CASE(2, 4, 5, true)
  SPEEXHIP_SLIDE_CASE(4, 2, 3, false)
  SPEEXHIP_SLIDE_CASE(2, 3, 3, false)
  SPEEXHIP_SLIDE_CASE(2, 4, 3, false)
  SPEEXHIP_SLIDE_CASE(2, 7, 1, true)
  SPEEXHIP_SLIDE_CASE(2, 7, 1, false)
  SPEEXHIP_SLIDE_CASE(2, 9, 1, true)
  SPEEXHIP_SLIDE_CASE(2, 9, 1, false)
  SPEEXHIP_SLIDE_CASE(2, 10, 1, true)
  SPEEXHIP_SLIDE_CASE(2, 10, 1, false)
  SPEEXHIP_SLIDE_CASE(2, 5, 2, true)
  SPEEXHIP_SLIDE_CASE(2, 5, 3, true)
  SPEEXHIP_SLIDE_CASE(2, 5, 4, true)
  SPEEXHIP_SLIDE_CASE(2, 5, 2, false)
  SPEEXHIP_SLIDE_CASE(1, 16, 1, true)
  SPEEXHIP_SLIDE_CASE(1, 16, 1, false)
  SPEEXHIP_SLIDE_CASE(1, 20, 1, true)
  SPEEXHIP_SLIDE_CASE(1, 20, 1, false)
  SPEEXHIP_SLIDE_CASE(1, 24, 1, true)
  SPEEXHIP_SLIDE_CASE(1, 24, 1, false)
  SPEEXHIP_SLIDE_CASE(2, 8, 3, true)
  SPEEXHIP_SLIDE_CASE(2, 8, 2, false)
  SPEEXHIP_SLIDE_CASE(1, 6, 5, true)
  SPEEXHIP_SLIDE_CASE(1, 6, 3, false)
  SPEEXHIP_SLIDE_CASE(1, 5, 6, true)
  SPEEXHIP_SLIDE_CASE(1, 5, 3, false)
#undef SPEEXHIP_SLIDE_CASE
  return hipErrorInvalidValue;
}

}  // namespace speexhip
