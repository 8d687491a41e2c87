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

// acc.xy += tap[hi].xx * x.xy   (channel-pair packing: tap broadcast from an SGPR pair)
__device__ __forceinline__ void fma_bcast_tap(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x, bool hi) {
  if (hi)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}
// acc.xy += tap.xy * x.xx       (phase-pair packing: sample broadcast from the low half of a pair)
__device__ __forceinline__ void fma_bcast_x(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}

// PAIR_CH: true = channel pairs (NP = den accumulators per period), false = phase pairs (NP = ceil(den/2)).
// P: periods per lane; NUM: input frames per period; U = P*NUM tap steps per iteration.
template <int P, int NUM, int NP, bool PAIR_CH, bool PACKED, typename T>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(96))) void resample_slide(
    SlideParams p, const float *__restrict__ rows, const StreamDesc *__restrict__ streams, DescPack pack) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const StreamDesc d = PACKED ? pack.d[blockIdx.y] : streams[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {
    roll_history<T>(p.channels, d, p.threads);
    return;
  }
  if (d.n_out == 0) return;
  const uint32_t C = p.channels;
  const uint32_t K_end = d.k_shift + d.n_out;
  const uint32_t m_total = (K_end + p.den - 1) / p.den;
  const uint32_t tile_periods = p.blocks_per_tile * P;  // lane blocks x P
  const uint32_t m_lo = blockIdx.x * tile_periods;
  if (m_lo >= m_total) return;
  const uint32_t m_cnt = min(tile_periods, m_total - m_lo);

  // ---- stage: frames [f0, f0 + m_cnt*num + row_len + one row) of V as float, in rows of P*NUM
  //      frames `row_stride` floats apart: the shared loader's padded image with the row as its
  //      padding period (16-byte loads, all in flight at once; device_helpers.h) ----
  const WindowGeom wg = window_geom<T>(d, C, NUM, NUM + p.row_len + P * NUM, m_lo, m_cnt, p.threads,
                                       p.row_stride - P * NUM * C, p.row_magic, P * NUM * C);
  if (!(p.skip & 2u)) {
    u32x4 w[4];
    window_fetch<4, T>(wg, w);
    window_commit<4, T>(xs, d, wg, w);
  }
  __syncthreads();

  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t cg = lane % p.cgroups;        // channel pair (PAIR_CH) or channel (phase pairs)
  const uint32_t lb = wave * p.blocks_per_wave + lane / p.cgroups;  // lane block inside the tile
  const bool lane_live = (lane / p.cgroups) < p.blocks_per_wave && lb * P < m_cnt;
  constexpr int CW = PAIR_CH ? 2 : 1;          // floats this lane reads per frame
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
  const float *__restrict__ trow = rows;  // wave-uniform, __restrict__ kernel argument -> s_load
  const uint32_t n_it = (p.skip & 4u) ? 0 : p.row_len / U;
  for (uint32_t it = 0; it < n_it; it++, trow += TAPS_IT, xrow += p.row_stride) {
    f32x2 tp[TP];
#pragma unroll
    for (int j = 0; j < TP; j++) tp[j] = *reinterpret_cast<const f32x2 *>(trow + 2 * j);
    constexpr int W = (2 * P - 1) * NUM;  // frames it*U .. it*U + W-1 of this lane's block
    f32x2 xw[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
      const float *px = xrow + (j < U ? j * C : p.row_stride + (j - U) * C);
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
            fma_bcast_x(acc[pp][r], tp[s * NP + r], xw[pp * NUM + s]);
          }
        }
  }
  if (!lane_live || (p.skip & 8u)) return;

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
        o0[2 * (P * NP - 1)] = acc[P - 1][NP - 1].x;
        o0[2 * (P * NP - 1) + 1] = acc[P - 1][NP - 1].y;
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

template <int P, int NUM, int NP, bool PAIR_CH, typename T>
hipError_t launch_up(const SlideParams &p, const StreamDesc *d_descs, const DescPack *pack, dim3 grid,
                     uint32_t threads, size_t lds_bytes, hipStream_t stream) {
  DescPack empty;
  if (pack == nullptr) std::memset(&empty, 0, sizeof(empty));
  static std::atomic<uint64_t> seen_packed{0}, seen_ring{0};
  if (pack != nullptr)
    opt_in_lds_on_this_device(resample_slide<P, NUM, NP, PAIR_CH, true, T>, seen_packed);
  else
    opt_in_lds_on_this_device(resample_slide<P, NUM, NP, PAIR_CH, false, T>, seen_ring);
  if (pack != nullptr)
    hipLaunchKernelGGL((resample_slide<P, NUM, NP, PAIR_CH, true, T>), grid, dim3(threads), lds_bytes, stream, p,
                       p.rows, nullptr, *pack);
  else
    hipLaunchKernelGGL((resample_slide<P, NUM, NP, PAIR_CH, false, T>), grid, dim3(threads), lds_bytes, stream, p,
                       p.rows, d_descs, empty);
  return hipGetLastError();
}

}  // namespace

namespace {
struct SlideShape { uint32_t num, np; bool pair_ch; uint32_t p; };
// the instantiated (num, accumulator pairs per period, packing) -> periods per lane.  Bounds kept:
// tap floats per iteration <= 60, window (2P-1)*num <= 42 frames, P*np <= 24 accumulator pairs.
const SlideShape kShapes[] = {
    {1, 1, true, 8}, {1, 2, true, 8}, {1, 3, true, 8}, {1, 4, true, 4}, {1, 6, true, 4},
    {1, 1, false, 8}, {1, 2, false, 8}, {1, 3, false, 4},
    {2, 1, true, 8}, {2, 3, true, 4}, {2, 1, false, 8}, {2, 2, false, 4},
    {3, 1, true, 4}, {3, 2, true, 4}, {3, 1, false, 4},
    {4, 1, true, 4}, {4, 1, false, 4},
    // 5:1 and 6:1 decimation (48k -> 8k, 96k -> 16k): a 42-frame register window, 4 waves per SIMD
    {5, 1, true, 4}, {5, 1, false, 4}, {6, 1, true, 4}, {6, 1, false, 4},
    // 8:1 and 12:1 (192k -> 24k / 16k, 96k -> 12k / 8k): two periods per lane, 24- / 36-frame window
    {8, 1, true, 2}, {8, 1, false, 2}, {12, 1, true, 2}, {12, 1, false, 2},
    // den = 5 (8k -> 40k, 16k -> 40k, 24k -> 40k, 32k -> 40k): five accumulator pairs per period for channel
    // pairs; odd channel counts pad den to 6 phases (np = 3, the den = 6 shapes)
    {1, 5, true, 4}, {2, 5, true, 4}, {3, 5, true, 2}, {4, 5, true, 2},
    {2, 3, false, 4}, {3, 3, false, 2}, {4, 3, false, 2},
    // 7:1, 9:1, 10:1 (56k -> 8k, 72k -> 8k, 44.1k -> 4.41k, 80k -> 8k)
    {7, 1, true, 2}, {7, 1, false, 2}, {9, 1, true, 2}, {9, 1, false, 2}, {10, 1, true, 2}, {10, 1, false, 2},
    // 5:2, 5:3, 5:4 (40k -> 16k / 24k / 32k)
    {5, 2, true, 2}, {5, 3, true, 2}, {5, 4, true, 2}, {5, 2, false, 2},
    // 16:1, 20:1, 24:1 (128k -> 8k, 160k -> 8k, 192k -> 8k): ONE period per lane, its num frames the whole
    // register window -- one FMA per sample read, but still SGPR taps and no double arithmetic: 32 streams of
    // 192k -> 8k stereo q7 142 us against 1269 us on the exact kernel they used to fall back to, 16:1 156
    // against 1850.  (11:1 measured no gain -- 244 vs 230 us -- and stays on the exact kernel.)
    {16, 1, true, 1}, {16, 1, false, 1},
    {20, 1, true, 1}, {20, 1, false, 1}, {24, 1, true, 1}, {24, 1, false, 1},
    // 8:3 (32k -> 12k, 64k -> 24k, 128k -> 48k), 6:5 (48k -> 40k) and 5:6 (40k -> 48k): 68 / 64 / 67 us
    // against 408 / 535 / 550 on the exact kernel (stereo q7, 32 streams of 2^18 frames)
    {8, 3, true, 2}, {8, 2, false, 2}, {6, 5, true, 1}, {6, 3, false, 1}, {5, 6, true, 1}, {5, 3, false, 1},
};
}  // namespace

// LDS of a workgroup of `waves` waves: one row per lane block + the rows the last lane's window runs into
// (+ 16 floats: the image starts on the input's 16-byte grid and ends on a whole load)
static size_t slide_lds_bytes(const SlidePlan &t, uint32_t waves) {
  const uint32_t steps = t.p * t.num;
  const size_t rows_needed = static_cast<size_t>(64 / t.cgroups) * waves + t.row_len / steps + 2;
  return (rows_needed * t.row_stride + 16) * 4;
}
const size_t kSlideLdsLimit = 160 * 1024;  // what one workgroup can have on gfx950

SlidePlan plan_slide(const FilterSpec &f, uint32_t channels) {
  SlidePlan t;
  t.pair_ch = channels % 2 == 0;
  t.np = t.pair_ch ? f.den : (f.den + 1) / 2;
  t.cgroups = t.pair_ch ? channels / 2 : channels;
  t.num = f.num;
  t.usable = false;
  for (const SlideShape &sh : kShapes)
    if (sh.num == f.num && sh.np == t.np && sh.pair_ch == t.pair_ch) {
      t.p = sh.p;
      t.usable = f.den <= 6 && t.cgroups <= 64;
    }
  if (!t.usable) return t;
  const uint32_t steps = t.p * f.num;  // tap steps per iteration
  const uint32_t dmax = static_cast<uint32_t>((static_cast<uint64_t>(f.den - 1) * f.num) / f.den);
  t.row_len = (f.taps + dmax + steps - 1) / steps * steps;
  const uint32_t row_elems = steps * channels;
  // row stride: distinct banks for the lanes of a wave (ds_read_b64: stride/2 odd; b32: stride odd)
  t.row_stride = row_elems;
  if (t.pair_ch) {
    if ((t.row_stride / 2) % 2 == 0) t.row_stride += 2;
  } else {
    if (t.row_stride % 2 == 0) t.row_stride += 1;
  }
  // a long filter on many channels (12:1 q10 on 8 channels: 200 KB with 8 waves) runs smaller workgroups
  // (launch_slide); one that does not even fit two waves runs the exact kernel
  if (slide_lds_bytes(t, 2) > kSlideLdsLimit) t.usable = false;
  return t;
}

void build_slide_rows(const FilterSpec &f, const SlidePlan &t, std::vector<float> *rows) {
  // [step][phase] (phase count padded to 2*np for phase pairs), rows of phase r shifted by
  // delta_r, + one iteration of zero padding
  const uint32_t width = t.pair_ch ? f.den : 2 * t.np;
  rows->assign(static_cast<size_t>(t.row_len + t.p * f.num) * width, 0.f);
  std::vector<double> h(f.taps);
  for (uint32_t r = 0; r < f.den; r++) {
    const uint32_t phase = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) % f.den);
    const uint32_t shift = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) / f.den);
    phase_taps(f, phase, h.data());
    for (uint32_t j = 0; j < f.taps; j++)
      (*rows)[static_cast<size_t>(j + shift) * width + r] = static_cast<float>(h[j]);
  }
}

hipError_t launch_slide(const FilterSpec &f, const SlidePlan &t, const float *d_rows, uint32_t channels,
                        const StreamDesc *h_descs, const StreamDesc *d_descs, const DescPack *pack,
                        uint32_t n_streams, bool float_io, hipStream_t stream) {
  uint32_t max_periods = 0;
  for (uint32_t s = 0; s < n_streams; s++) {
    if (h_descs[s].n_out == 0) continue;
    const uint64_t k_end = static_cast<uint64_t>(h_descs[s].k_shift) + h_descs[s].n_out;
    max_periods = std::max<uint32_t>(max_periods, static_cast<uint32_t>((k_end + f.den - 1) / f.den));
  }
  // lane blocks per wave, waves per workgroup: fewer waves when the launch is small, so that it
  // still spreads over the chip
  const uint32_t blocks_per_wave = 64 / t.cgroups;
  // (8 waves per workgroup: the kernel's 96 SGPRs admit 7 waves per SIMD, i.e. three such
  //  workgroups per CU but only one of 16 waves; measured 16 -> 8: 2:1 decimation 241 -> 205 us,
  //  16k->48k mono 463 -> 425 us, the rest within 2 %)
  static const uint32_t max_waves = std::getenv("SPEEXHIP_SLIDE_WAVES") ? std::atoi(std::getenv("SPEEXHIP_SLIDE_WAVES")) : 8;
  uint32_t waves = max_waves;
  while (waves > 2 && static_cast<uint64_t>(max_periods) * n_streams < 512ull * waves * blocks_per_wave * t.p)
    waves /= 2;
  while (waves > 2 && slide_lds_bytes(t, waves) > kSlideLdsLimit) waves /= 2;  // (fits with 2: plan_slide)
  SlideParams p;
  p.rows = d_rows;
  p.den = f.den;
  p.taps = f.taps;
  p.row_len = t.row_len;
  p.channels = channels;
  p.cgroups = t.cgroups;
  p.blocks_per_wave = blocks_per_wave;
  p.blocks_per_tile = blocks_per_wave * waves;
  p.row_stride = t.row_stride;
  p.row_magic = period_magic_of(t.p * f.num * channels);
  static const uint32_t skip_mask = std::getenv("SPEEXHIP_SKIP") ? std::atoi(std::getenv("SPEEXHIP_SKIP")) : 0;
  p.skip = skip_mask;
  const uint32_t tile_periods = p.blocks_per_tile * t.p;
  const uint32_t tiles = (max_periods + tile_periods - 1) / tile_periods;
  // LDS: one row per lane block, + the rows the last lane's window runs into
  const size_t lds = slide_lds_bytes(t, waves);
  dim3 grid((max_periods == 0 ? 0 : tiles) + 1, n_streams, 1);
  const uint32_t threads = waves * 64;
  p.threads = threads;
#define SPEEXHIP_SLIDE_CASE(PP, NUMV, NPV, CHV)                           \
  if (t.p == PP && t.num == NUMV && t.np == NPV && t.pair_ch == CHV)      \
    return float_io ? launch_up<PP, NUMV, NPV, CHV, float>(p, d_descs, pack, grid, threads, lds, stream) \
                    : launch_up<PP, NUMV, NPV, CHV, int16_t>(p, d_descs, pack, grid, threads, lds, stream);
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
