#pragma once
// kernels_period_impl.h -- the device side of the period kernel (and launch_rc, the one place that names a
// kernel instance): included by kernels_period.hip (host side + the fp32 instances) and kernels_period64.hip (the
// fp64-accumulate instances, round 4), so that the two sets of instances compile side by side.
//
// kernels_period.hip -- the primary fast gfx950 FIR kernel ("period-lane" mapping).
//
// With K = k_shift + k = m*den + r (stream_plan.h) every output of every stream is
//     Out[r, m, c] = sum_s Tp[r][s] * V[base + m*num + delta_{g*R} + s][c]
// where Tp[r] are the effective taps of phase (r*num) mod den (the reference's four
// interpolation accumulators collapsed, deps/speex/resample.c:438-558; the direct kernels
// :331-435 as they are), pre-shifted so that the R phases of group g = r / R read the same
// input sample at the same step s.  fp32 FMA on the vector ALUs, no MFMA; +-1 LSB.
//
// Mapping (what makes it fast on CDNA4):
//   * lane  = one output PERIOD m (x one channel pair): the 64 lanes of a wave need the SAME
//     tap at every step, so taps never touch LDS or VGPRs -- they are wave-uniform, fetched by
//     scalar loads (s_load_dwordx16, L2 / scalar cache) and fed to v_pk_fma_f32 as SGPR
//     operands; one packed FMA updates both channels of a frame.
//   * wave  = one group of R consecutive phases: R accumulator pairs per lane, R FMAs per
//     LDS sample read; ~30 VGPRs -> 8 waves per SIMD hide the scalar-load and LDS latencies.
//   * LDS holds only the input window (float, channel-interleaved): lanes read it at a stride
//     of num*channels floats (conflict-free ds_read_b64 for the common ratios), and two
//     workgroups fit per CU, so one workgroup's staging / stores overlap the other's FMAs.
//   * outputs go straight from registers to HBM (4 bytes per lane per row); the partial lines
//     meet in L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "device_helpers.h"
#include "device_types.h"
#include "filter_design.h"
#include "kernels.h"

namespace speexhip {
// Diagnostics build only (-DSPEEXHIP_STAMPS, tools/stamps.py): every workgroup records when it
// reached a few points, on the 100 MHz s_memrealtime clock all CUs share.  Not in the product.
#ifdef SPEEXHIP_STAMPS
static __device__ unsigned long long g_stamps[8192 * 16];
#define STAMP(k)                                                                                         \
  do {                                                                                                   \
    const uint32_t lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                \
    if ((threadIdx.x & 63u) == 0 && lin_ < 8192)                                                         \
      atomicMax(&g_stamps[lin_ * 16 + (k)], (unsigned long long)__builtin_amdgcn_s_memrealtime());        \
  } while (0)
#define STAMP_FIRST(k)                                                                                   \
  do {                                                                                                   \
    const uint32_t lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                \
    if ((threadIdx.x & 63u) == 0 && lin_ < 8192)                                                         \
      atomicMax(&g_stamps[lin_ * 16 + (k)], ~(unsigned long long)__builtin_amdgcn_s_memrealtime());       \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_FIRST(k) do { } while (0)
#endif
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
// taps per bank of the FIR loop's two-bank pipeline, by phases per wave (host + device)
__host__ __device__ constexpr int bank_taps(int r) { return r == 10 ? 20 : 30; }
// dword-aligned wide global stores (global memory needs only dword alignment for x2/x4)
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(1))) u32x4_a4 g_u32x4_a4;
typedef __attribute__((address_space(1))) u32x2_a4 g_u32x2_a4;
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// acc += tap * x on both halves, the tap being element HI of a wave-uniform pair held in SGPRs.
// Written as one instruction so that the odd element is selected in place with op_sel: left to
// itself hipcc copies odd taps into even SGPRs first (one s_mov each), and the scalar ALU --
// ONE per CU, shared by all 32 resident waves -- becomes the bottleneck of the loop.
__device__ __forceinline__ void fma_tap(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x, bool hi) {
  if (hi)  // constant after unrolling: the branch folds away
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}

#include "fir_loop_asm.inc"

// Wave priority of the FIR loop (and of whatever follows it): 0.  Bit 2 of p.prio (diagnostics, SPEEXHIP_PRIO=7)
// gives the workgroup in the CU's second slot (HW_ID.TG_ID) priority 1 instead: an experiment of round 3.
// Two workgroups that share a CU are dispatched together and do the same work; in a kernel of nothing but
// equal phases they stay in lockstep generation after generation (tools/probe_slots.hip: the pairs start
// within 0.2 us of each other every time), and a model of that -- both staging, both computing, both
// storing -- predicted exactly the launch time measured here.  The stamps say otherwise for THIS kernel
// (tools/stamps.py, profiles/r03_stamps_cfg2_s32.txt): the pairs run 11.5 us apart (median) of a 26 us
// cycle, a CU has two workgroups in their FIR loops 60 % of the time, one 37.5 %, none 2.2 %; what the lone
// one loses is issue efficiency (4 waves per SIMD: 5.1 cycles per v_pk_fma_f32 against 4.5 at 8), not time
// behind a partner.  Unequal priorities made every launch SLOWER: cfg2 32 streams 200 -> 215 us, mono
// 128 -> 145, 8 channels 569 -> 602 (the low-priority workgroup starves and its slot turns over late:
// median turnover 1.4 -> 2.2 us, p90 2.2 -> 11.8).  Not the default.
__device__ __forceinline__ void set_fir_priority(const PeriodParams &p) {
  if (p.prio & 4u) {
    const uint32_t tg_id = (__builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11)));  // HW_REG_HW_ID bits [19:16]
    if (tg_id & 1u) {
      __builtin_amdgcn_s_setprio(1);
      return;
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// What a lane needs to know about its place in a tile.
struct LaneCtx {
  uint32_t C;       // channels per frame
  uint32_t cg;      // channel group of this lane
  bool live;        // the lane's period exists in this tile
  bool live_b;      // single-channel lanes: the lane's SECOND period (half a tile further) exists
  uint32_t xlane;   // float index of the lane's first sample of a group with delta_g = 0
  uint64_t K_lane;  // canonical output index of the lane's period, phase 0
};

// ROWS (round 5; padded 8-channel frames on the fp32 chain: BASELINE configs[3]): the channel pair of a lane is its ROW
// of 16 lanes (cg = lane / 16, period = lane % 16) instead of lane % 4.  Two things follow.  The four lanes of a frame
// sit in the four rows of one column, where v_permlane32_swap / v_permlane16_swap exchange them: a 4 x 4 transpose of
// (frame, channel pair) dwords is four swap instructions, and a lane then owns whole 16-byte frames -- 3 stores per
// group instead of 10 four-byte ones 16 bytes apart (store_group_rows).  And with a pad of 4 floats per period the 32
// lanes of a half-wave (2 channel pairs x 16 periods, 8 bytes each) cover all 64 banks once: the loop's reads are
// conflict-free as before (plan_period_r picks the pad for this mapping).
template <int CT, bool ONE_GROUP, bool PADDED, int CGF = 0, bool PP = false, bool ROWS = false>
__device__ __forceinline__ LaneCtx lane_ctx(const PeriodParams &p, uint32_t xshift, uint32_t m_lo,
                                            uint32_t m_cnt, uint32_t lane) {
  // ONE_GROUP: the frame is exactly one channel group (mono, stereo): the sample stride is a
  // compile-time constant and the LDS reads of an iteration share one address register.
  // CGF != 0: a frame of CGF channel groups (4, 6, 8 channels as 2, 3, 4 pairs; round 5: 5 and 7 single channels) known at compile time:
  // the same for the common multi-channel layouts -- the FIR loop of 8 channels spent 4 vector adds per
  // 40 FMAs on LDS addresses with the stride in a register.
  constexpr uint32_t kGroups = ONE_GROUP ? 1u : static_cast<uint32_t>(CGF);
  const uint32_t cgroups = kGroups != 0 ? kGroups : p.cgroups;
  LaneCtx c;
  c.C = kGroups != 0 ? kGroups * CT : p.channels;
  static_assert(!ROWS || (CGF == 4 && CT == 2), "the row mapping is for frames of four channel pairs");
  c.cg = ONE_GROUP ? 0 : ROWS ? lane >> 4 : lane % cgroups;
  const uint32_t pl = ONE_GROUP ? lane : ROWS ? (lane & 15u) : lane / cgroups;  // period of this lane inside the tile
  // CT == 1 (odd channel counts): a packed FMA has no second channel to work on, so the lane takes
  // a second PERIOD instead, half a tile further (p.half_periods): .x = period pl, .y = pl + half.
  // (PP -- phase pairs, round 4: a single-channel lane with ONE period and 2R phases of it; FirLoopAsmPP)
  const uint32_t lane_max = (CT == 1 && !PP) ? p.half_periods : p.lane_periods;
  c.live = pl < lane_max && pl < m_cnt;
  c.live_b = CT == 1 && !PP && pl < lane_max && pl + p.half_periods < m_cnt;
  // PADDED: the LDS image carries p.pad floats after every period (num frames) so that the
  // lanes of a wave -- num*C floats apart, a multiple of the bank count for e.g. 8 channels at
  // num = 160 -- hit distinct banks; the per-step offset is then wave-uniform scalar arithmetic.
  c.xlane = xshift + min(pl, lane_max - 1) * (p.num * c.C + (PADDED ? p.pad : 0u)) + c.cg * CT;
  c.K_lane = static_cast<uint64_t>(m_lo + pl) * p.den;
  return c;
}

// acc[i] += group g's taps times the lane's samples, over the iterations the host tabulated for g.
// CF: floats per frame when that is a compile-time constant (mono, stereo, 4 / 6 / 8 channels), else 0.
// W16: the LDS window holds int16 samples (device_helpers.h); ISA loop only.
template <int R, int CT, bool PADDED, int CF = 0, bool W16 = false>
__device__ __forceinline__ void fir_group(const PeriodParams &p, const float *__restrict__ rows, const float *xs,
                                          const LaneCtx &c, uint32_t g, bool skip_all, f32x2 (&acc)[R]) {
  const uint32_t C = c.C;
  const uint32_t delta_g = p.delta[g];  // (g*R*num) div den, tabulated on the host
  // (delta_g < num: no padding boundary before the group's first sample)
  // (E: the element type of the LDS window -- int16 samples as they came from HBM for W16, floats otherwise; positions count
  //  elements either way)
  using E = typename std::conditional<W16, int16_t, float>::type;
  const E *xp = reinterpret_cast<const E *>(xs) + c.xlane + delta_g * C;
#ifndef SPEEXHIP_CXX_FIR_LOOP
  constexpr bool kIsa16 = W16 && CF != 0 && FirLoopAsm<R, CT, CF, PADDED, W16>::available;
#else
  constexpr bool kIsa16 = false;
#endif
  // Round 6: an int16 window on the layouts WITHOUT an ISA loop too (frames of 9, 11, 13-15, 17 ... channels: the C++ loop
  // below converts each sample it reads) -- their wide-window decimators held half the periods per tile of their
  // neighbours with ISA loops and ran at 0.10-0.16 of the vector peak beside 0.33-0.39 (profiles/r06_sweep_frames.txt).
  if constexpr (kIsa16) {
    using Isa = FirLoopAsm<R, CT, CF, PADDED, true>;
    constexpr uint32_t kStepsPerTrip = 2 * Isa::steps_per_bank;
    const uint32_t trips = skip_all ? 0u : p.delta[2 * p.groups + g];
    const uint32_t head = R == 10 ? trips & 15u : 0u, tail = R == 10 ? (trips >> 4) & 15u : 0u;
    const float *rows_g = rows + static_cast<size_t>(g) * p.l4 * (2 * bank_taps(R));
    const uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xs)) + (c.xlane + delta_g * C) * 2u;
    auto sgpr = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
    const uint64_t rows_bits = reinterpret_cast<uint64_t>(rows_g);
    const float *rows_s = reinterpret_cast<const float *>(static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits))) |
                                                          static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits >> 32))) << 32);
    Isa::run(acc, rows_s, addr, CT == 1 ? addr + p.half_offset * 2u : 0u, sgpr(head), sgpr((trips >> 8) - head - tail),
             sgpr(tail), sgpr(PADDED ? p.delta[p.groups + g] : 0u), sgpr(p.wrap_step),
             sgpr((kStepsPerTrip * CF + p.pad) * 2u));
    return;
  }
#ifndef SPEEXHIP_CXX_FIR_LOOP
  // The loop in ISA (csrc/gen_fir_loop.py) for the layouts it is generated for; the C++ loop below is
  // its reference -- same taps, same samples, same order per accumulator -- and runs the other layouts
  // (odd channel counts >= 3, more than 8 channels).  -DSPEEXHIP_CXX_FIR_LOOP builds the A/B library.
  using Isa = FirLoopAsm<R, CT, CF, PADDED, false>;
  if constexpr (!W16 && CF != 0 && Isa::available) {
    constexpr uint32_t kStepsPerTrip = 2 * Isa::steps_per_bank;
    static_assert(kStepsPerTrip * R == 2 * bank_taps(R), "the ISA loop and the tap rows disagree on the trip");
    const uint32_t trips = skip_all ? 0u : p.delta[2 * p.groups + g];  // head | tail << 4 | total << 8
    const uint32_t head = R == 10 ? trips & 15u : 0u, tail = R == 10 ? (trips >> 4) & 15u : 0u;
    const float *rows_g = rows + static_cast<size_t>(g) * p.l4 * (2 * bank_taps(R));
    const uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xp));
    // (wave-uniform values all of them, but hipcc keeps some of them in VGPRs -- the group index is a loop
    //  counter it moved to the vector side -- and will not copy them back by itself)
    auto sgpr = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
    const uint64_t rows_bits = reinterpret_cast<uint64_t>(rows_g);
    const float *rows_s = reinterpret_cast<const float *>(static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits))) |
                                                          static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits >> 32))) << 32);
    // (Round 5, VERDICT r4 #3a, measured and not kept: a padded window whose group crosses ONE period boundary --
    //  BASELINE configs[3]: rows of 26 trips, boundaries 40 trips apart -- run as the UNPADDED loop on either side of it,
    //  the window address stepped over the padding between the two runs, instead of counting down to the boundary in
    //  every trip: four scalar instructions less per trip, one drained and refilled bank per group.  Same box, same
    //  library, three repetitions each: cfg4 x 32 573.8 / 573.8 / 574.0 us split against 574.7 / 573.9 / 575.8 counting,
    //  one stream 25.08 / 24.96 / 25.18 against 25.10 / 24.95 / 24.86, 8 channels 32k->44.1k 375.8 / 377.0 / 374.5 against
    //  381.1 / 375.4 / 376.2, 48k->11.025k 273.2 / 273.6 / 273.6 against 273.7 / 274.0 / 273.9: profiles/r05_ab_split.txt.
    //  The scalar unit is 39 % busy in this launch (90 M scalar instructions on 231 M CU cycles) and its instructions
    //  issue beside other waves' FMAs: they were never what the loop waits for.)
    Isa::run(acc, rows_s, addr, CT == 1 ? addr + p.half_offset * 4u : 0u, sgpr(head), sgpr((trips >> 8) - head - tail),
             sgpr(tail), sgpr(PADDED ? p.delta[p.groups + g] : 0u), sgpr(p.wrap_step),
             sgpr((kStepsPerTrip * CF + p.pad) * 4u));
    return;
  }
#endif
  // padded layout: the host shifted this group's start by <= 3 frames so that the one padding
  // boundaries its window crosses fall between iterations (the first before iteration wrap_it)
  // (counted DOWN to the next boundary, like the loop itself: with 80 SGPRs there is no register to
  //  spare for loop bounds, and a bound reloaded from the kernel arguments drags an lgkmcnt(0) wait
  //  -- i.e. the whole tap prefetch -- into every iteration)
  uint32_t to_wrap = PADDED ? p.delta[p.groups + g] : 0u;
  // Taps are wave-uniform: they travel HBM/L2 -> scalar cache -> SGPRs (s_load: the row pointer
  // is a __restrict__ kernel argument, so the loads are provably invariant) and feed
  // v_pk_fma_f32 directly.
  // (s_setprio around the loop -- FIR waves ahead of staging / storing ones -- measured 6 % slower.)
  // Bank A = steps 0-1 of an iteration, bank B = steps 2-3: each bank is its 2R taps (R SGPR
  // pairs) plus its two sample reads.  Order per iteration, pinned with sched_barrier:
  //   wait A | issue loads B | 2R FMAs A | wait B | issue loads A(next) | 2R FMAs B
  // A wait is lgkmcnt(0) (scalar loads return out of order and share the counter with LDS),
  // so a bank's loads must be issued right AFTER the other bank's wait; `touch_bank` is an
  // empty asm that reads the bank and thereby makes hipcc put the wait exactly there.
  // A bank is 20 taps = 10 SGPR pairs for R = 10 (2 steps).  The R = 5 kernel only runs launches of
  // one generation -- one workgroup per CU, 4 waves per SIMD, so the 80-SGPR cap that buys the second
  // resident workgroup is not needed there -- and takes banks of 30 taps (6 steps): 30 FMAs between
  // two waits instead of 20 give the scalar loads half as much time again to land.
  constexpr int BANK = bank_taps(R), NP = BANK / 2;  // taps / SGPR pairs per bank
  constexpr int STEPS = BANK / R;                     // steps per bank; an iteration is two banks
  static_assert(R == 10 || R == 5, "phases per wave");
  const float *__restrict__ trow = rows + static_cast<size_t>(g) * p.l4 * (2 * BANK);
  f32x2 ta[NP], tb[NP], xa[STEPS], xb[STEPS];
  auto load_bank = [&](f32x2 (&t)[NP], f32x2 (&x)[STEPS], const float *tp, const E *sp) {
#pragma unroll
    for (int j = 0; j < NP; j++) t[j] = *reinterpret_cast<const f32x2 *>(tp + 2 * j);
#pragma unroll
    for (int u = 0; u < STEPS; u++) {
      if constexpr (W16) {  // (one 4-byte read and two conversions for a channel pair; two 2-byte reads for a lane's two periods)
        x[u].x = static_cast<float>(sp[u * C]);
        x[u].y = static_cast<float>(CT == 2 ? sp[u * C + 1] : sp[u * C + p.half_offset]);
      } else if (CT == 2) {
        x[u] = *reinterpret_cast<const f32x2 *>(sp + u * C);
      } else {
        // (mono: two steps of each period arrive as (a0, a1), (b0, b1) and hipcc re-pairs them with ~7 v_mov
        //  per 4 steps; one v_pk_mov_b32 per pair instead -- 4 per 4 steps -- measured slower, 145.5 vs
        //  142.1 us for 32 mono streams of 44.1k -> 48k, 170 vs 159 for 48k -> 44.1k)
        x[u].x = sp[u * C];
        x[u].y = sp[u * C + p.half_offset];  // the same step of the lane's second period
      }
    }
  };
  auto touch_bank = [&](const f32x2 (&t)[NP], const f32x2 (&x)[STEPS]) {
    if constexpr (NP == 10) {
      asm volatile("" ::"s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]), "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]),
                   "s"(t[8]), "s"(t[9]), "v"(x[0]), "v"(x[1]));
    } else {
      static_assert(NP == 15 && STEPS == 6, "touch_bank lists 10 pairs + 2 samples or 15 pairs + 6 samples");
      asm volatile("" ::"s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]), "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]),
                   "s"(t[8]), "s"(t[9]), "s"(t[10]), "s"(t[11]), "s"(t[12]), "s"(t[13]), "s"(t[14]), "v"(x[0]),
                   "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]));
    }
  };
  // phases [LO, HI) of the group only: the head and tail iterations of a group, where the host
  // knows half of the rows to be all zero (below)
  auto fma_bank = [&](const f32x2 (&t)[NP], const f32x2 (&x)[STEPS], auto lo_c, auto hi_c) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
    for (int u = 0; u < STEPS; u++)
#pragma unroll
      for (int i = LO; i < HI; i++) fma_tap(acc[i], t[(u * R + i) >> 1], x[u], ((u * R + i) & 1) != 0);
  };
  load_bank(ta, xa, trow, xp);
  // (Per trip hipcc spends two vector instructions beside the 40 FMAs: the window address and the trip
  //  count, which it keeps in a VGPR -- v_add_co, branch on vcc -- because no SGPR is left under the cap.
  //  Forcing the count into an SGPR, or running two iterations per trip (one address step, scalar count:
  //  1 in 81), pushes SGPR spills past the 64 VGPRs into scratch in every instantiation: 32 streams
  //  208 -> 219 us, 8 channels 607 -> 667, float 273 -> 337.  Re-reading the parameters and the descriptor
  //  from memory after the loop, so that they need not live across it, takes the headline instance from
  //  18 to 3 spilled SGPRs -- and the scalar count still spills: it is the loop's own 40 taps + pointers +
  //  temporaries that fill the budget.)
  auto run = [&](uint32_t count, auto lo_c, auto hi_c) {
    for (uint32_t left = count; left != 0; left--) {
      touch_bank(ta, xa);
      __builtin_amdgcn_sched_barrier(0);
      load_bank(tb, xb, trow + BANK, xp + STEPS * C);
      __builtin_amdgcn_sched_barrier(0);
      fma_bank(ta, xa, lo_c, hi_c);
      __builtin_amdgcn_sched_barrier(0);
      touch_bank(tb, xb);
      __builtin_amdgcn_sched_barrier(0);
      trow += 2 * BANK;
      xp += 2 * STEPS * C;
      if (PADDED && --to_wrap == 0) {  // wave-uniform: the window pointer steps over the bank padding
        xp += p.pad;
        to_wrap = p.wrap_step;         // the next period boundary, num/4 iterations on (or never)
      }
      // next iteration's bank A: the rows carry one iteration of zero padding past the last
      // group and the window one step group of slack, so the final prefetch stays in bounds
      load_bank(ta, xa, trow, xp);
      __builtin_amdgcn_sched_barrier(0);
      fma_bank(tb, xb, lo_c, hi_c);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // The R rows of a group start up to R-1 steps apart (each phase's window begins ~num/den input
  // frames after the previous one's), so the loop covers taps + that spread steps and every row is
  // zero outside its own `taps` of them.  Whole iterations in which one HALF of the rows is zero are
  // run on the other half only: `head` leading iterations where rows R/2.. have not begun, `tail`
  // trailing ones where rows ..R/2-1 have ended (44.1k->48k q7: 1400 -> 1320 or 1340 FMAs per group).
  // The counts come from the host per group (build_period_rows); skipped products are exact zeros.
  if constexpr (R == 10) {
    const uint32_t trips = skip_all ? 0u : p.delta[2 * p.groups + g];  // head | tail << 4 | total << 8
    const uint32_t head = trips & 15u, tail = (trips >> 4) & 15u;
    // (the accumulators are made opaque between the loops: left alone the register allocator ties
    //  all of them into one 32-register tuple in some instantiations and spills it around each loop)
    auto pin = [&]() {
      asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                   "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]));
    };
    run(head, std::integral_constant<int, 0>(), std::integral_constant<int, R / 2>());
    pin();
    run((trips >> 8) - head - tail, std::integral_constant<int, 0>(), std::integral_constant<int, R>());
    pin();
    run(tail, std::integral_constant<int, R / 2>(), std::integral_constant<int, R>());
  } else {
    run(skip_all ? 0u : p.delta[2 * p.groups + g] >> 8, std::integral_constant<int, 0>(), std::integral_constant<int, R>());
  }
  touch_bank(ta, xa);  // retire the last prefetch
}

// The same with an fp64 accumulator (round 4; FirLoopAsm64, csrc/gen_fir_loop.py): the reference's double kernels,
// deps/speex/resample.c:389-435 and :501-558.  rows: the group's taps as doubles [trip][step][R]; a trip is
// 2 * steps_per_bank steps (2 for R = 10, 6 for R = 5), the host's tables count in those (build_period_rows64).
template <int R, int CT, bool PADDED, int CF, bool W16 = false>
__device__ __forceinline__ void fir_group64(const PeriodParams &p, const double *__restrict__ rows, const float *xs,
                                            const LaneCtx &c, uint32_t g, bool skip_all, uint32_t part, uint32_t parts,
                                            double (&acc)[R][2]) {
  using Isa = FirLoopAsm64<R, CT, CF, PADDED, W16>;
  constexpr uint32_t EB = W16 ? 2u : 4u;  // bytes per window element (round 5: an int16 window here too)
  static_assert(CF != 0 && Isa::available, "the fp64 accumulator runs the ISA loop of its layout");
  constexpr uint32_t kStepsPerTrip = 2 * Isa::steps_per_bank;
  // (padded windows: the host's boundary tables count 4-step iterations; a trip here is kStepsPerTrip steps)
  constexpr uint32_t kPerIt = 4 / kStepsPerTrip;
  auto sgpr = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
  const uint32_t delta_g = p.delta[g];
  const uint32_t trips = skip_all ? 0u : p.delta[2 * p.groups + g];  // head | tail << 4 | total << 8
  const uint32_t head = R == 10 ? trips & 15u : 0u, tail = R == 10 ? (trips >> 4) & 15u : 0u, total = trips >> 8;
  // tap-range shares (fir_tile_parts64): this wave runs trips [t0, t1) of the group
  const uint32_t t0 = sgpr(total * part / parts), t1 = sgpr(total * (part + 1) / parts);
  auto overlap = [&](uint32_t lo, uint32_t hi) {
    const uint32_t a = max(t0, lo), b = min(t1, hi);
    return b > a ? b - a : 0u;
  };
  const uint32_t main_end = total - tail;
  uint32_t wraps = 0, to_wrap = 0;
  const uint32_t wrap_step = p.wrap_step * kPerIt;
  if constexpr (PADDED) {
    const uint32_t to_wrap0 = p.delta[p.groups + g] * kPerIt;
    to_wrap = to_wrap0;
    if (to_wrap0 != 0 && t0 >= to_wrap0) {
      const uint32_t past = t0 - to_wrap0;
      wraps = 1 + past / wrap_step;
      to_wrap = wrap_step - past % wrap_step;
    } else if (to_wrap0 != 0) {
      to_wrap = to_wrap0 - t0;
    }
  }
  const double *rows_g = rows + (static_cast<size_t>(g) * p.l4 + t0) * (kStepsPerTrip * R);
  const uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xs)) +
                        ((c.xlane + delta_g * c.C) + t0 * kStepsPerTrip * CF + wraps * p.pad) * EB;
  const uint64_t rows_bits = reinterpret_cast<uint64_t>(rows_g);
  const double *rows_s = reinterpret_cast<const double *>(static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits))) |
                                                          static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits >> 32))) << 32);
  Isa::run(acc, rows_s, addr, CT == 1 ? addr + p.half_offset * EB : 0u, sgpr(overlap(0, head)), sgpr(overlap(head, main_end)),
           sgpr(overlap(main_end, total)), sgpr(to_wrap), sgpr(wrap_step), sgpr((kStepsPerTrip * CF + p.pad) * EB));
}

// Round / interleave / store the R phases of group g for this lane's period: R consecutive
// frames per lane, 4 bytes each per channel pair.  With two workgroups per CU the stores
// overlap the other workgroup's FMAs (an LDS transpose for fully coalesced stores measured
// slower).
template <int R, int CT, bool ONE_GROUP, typename T>
__device__ __forceinline__ void store_group(const PeriodParams &p, const StreamDesc &d, const LaneCtx &c,
                                            uint32_t g, const f32x2 (&acc)[R]) {
  const uint32_t C = c.C, cg = c.cg;
  if constexpr (CT == 1) {
    // single-channel lanes: .x belongs to the lane's period, .y to its second period half a tile on
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (half == 0 ? !c.live : !c.live_b) continue;
      const int64_t k0 = static_cast<int64_t>(c.K_lane) + static_cast<int64_t>(half * p.half_periods) * p.den +
                         static_cast<int64_t>(g) * R - d.k_shift;
      const int64_t lo64 = k0 < 0 ? -k0 : 0;
      const int64_t hi64 = min(static_cast<int64_t>(R), min(static_cast<int64_t>(p.den) - static_cast<int64_t>(g) * R,
                                                           static_cast<int64_t>(d.n_out) - k0));
      G<T> *o = out_ptr<T>(d) + k0 * static_cast<int64_t>(C) + cg;
      if constexpr (ONE_GROUP && sizeof(T) == 2) {
        // mono int16: the lane's R samples leave as whole dwords -- 16 + 4 bytes (R = 10) or 8 (R = 5) -- around at
        // most two odd samples: the first when the run starts on the upper half of a dword (k_shift odd, or every
        // second period of an odd den), the last when what remains is odd.  (Until round 3 a run that started on
        // an upper half went out as R stores of 2 bytes, and so did every run of R = 5: a store instruction costs
        // one line request per lane whatever its width.)
        if (lo64 == 0 && hi64 == R && (reinterpret_cast<uintptr_t>(d.out) & 1u) == 0) {
          float v[R];
#pragma unroll
          for (int i = 0; i < R; i++) v[i] = half == 0 ? acc[i].x : acc[i].y;
          auto dwords = [&](g_i16 *at, auto first_c, auto count_c) {  // samples [first, first + 2*count) as dwords
            constexpr int F = decltype(first_c)::value, N = decltype(count_c)::value;
            uint32_t w[N > 0 ? N : 1];
#pragma unroll
            for (int j = 0; j < N; j++) w[j] = round_pack_pcm(v[F + 2 * j], v[F + 2 * j + 1]);
            g_u32 *od = (g_u32 *)at;
#pragma unroll
            for (int j = 0; j + 4 <= N; j += 4) *(g_u32x4_a4 *)(od + j) = u32x4_a4{w[j], w[j + 1], w[j + 2], w[j + 3]};
            if constexpr (N % 4 >= 2) *(g_u32x2_a4 *)(od + N / 4 * 4) = u32x2_a4{w[N / 4 * 4], w[N / 4 * 4 + 1]};
            if constexpr (N % 2 != 0) od[N - 1] = w[N - 1];
          };
          auto single = [&](int i) { o[i] = static_cast<int16_t>(round_pack_pcm(v[i], 0.f) & 0xffffu); };
          if ((reinterpret_cast<uintptr_t>(o) & 2u) == 0) {
            dwords(o, std::integral_constant<int, 0>(), std::integral_constant<int, R / 2>());
            if constexpr (R % 2 != 0) single(R - 1);
          } else {
            single(0);
            dwords(o + 1, std::integral_constant<int, 1>(), std::integral_constant<int, (R - 1) / 2>());
            if constexpr (R % 2 == 0) single(R - 1);
          }
          continue;
        }
      }
      if constexpr (ONE_GROUP && sizeof(T) == 4) {
        // mono float: the lane's R samples are R consecutive floats -- 16 + 16 + 8 bytes instead of R stores of 4
        // (a store instruction costs one line request per lane whatever its width: 20 of them per lane made a
        //  tile's store phase 4.5 us and 32 float mono streams of 44.1k -> 48k 228 us against 123 for int16)
        if (lo64 == 0 && hi64 == R) {
          float v[R];
#pragma unroll
          for (int i = 0; i < R; i++) v[i] = half == 0 ? acc[i].x : acc[i].y;
          // (opaque copies, as for the stereo float stores below)
#pragma unroll
          for (int i = 0; i < R; i++) asm volatile("" : "+v"(v[i]));
          G<float> *of = (G<float> *)o;
#pragma unroll
          for (int i = 0; i + 4 <= R; i += 4) *(G<f32x4_a4> *)(of + i) = f32x4_a4{v[i], v[i + 1], v[i + 2], v[i + 3]};
          if constexpr (R % 4 >= 2) *(G<f32x2_a4> *)(of + R / 4 * 4) = f32x2_a4{v[R / 4 * 4], v[R / 4 * 4 + 1]};
          if constexpr (R % 2 != 0) of[R - 1] = v[R - 1];
          continue;
        }
      }
#pragma unroll
      for (int i = 0; i < R; i++, o += C) {
        if (i < lo64 || i >= hi64) continue;
        const float v = half == 0 ? acc[i].x : acc[i].y;
        if constexpr (sizeof(T) == 4)
          o[0] = v;
        else
          o[0] = static_cast<int16_t>(round_pack_pcm(v, 0.f) & 0xffffu);
      }
    }
    return;
  }
  // rows i in [i_lo, i_hi) of this group are real phases that fall inside this call
  const int64_t k0 = static_cast<int64_t>(c.K_lane) + static_cast<int64_t>(g) * R - d.k_shift;
  const int64_t lo64 = k0 < 0 ? -k0 : 0;
  const int64_t hi64 = min(static_cast<int64_t>(R), min(static_cast<int64_t>(p.den) - static_cast<int64_t>(g) * R,
                                                       static_cast<int64_t>(d.n_out) - k0));
  const int i_lo = static_cast<int>(min(lo64, static_cast<int64_t>(R)));
  const int i_hi = static_cast<int>(max(hi64, static_cast<int64_t>(0)));
  if constexpr (sizeof(T) == 4) {
    // float I/O (resample.c:927-963): the FIR value as is
    G<float> *o = out_ptr<float>(d) + k0 * static_cast<int64_t>(C) + cg * CT;
    if (ONE_GROUP && CT == 2 && i_lo == 0 && i_hi == R) {
#pragma unroll
      for (int i = 0; i + 1 < R; i += 2) {
        // (through opaque copies: a 16-byte store straight from two accumulators makes the register
        //  allocator tie all R of them into one tuple, which it then spills around the FIR loops)
        float a0 = acc[i].x, a1 = acc[i].y, a2 = acc[i + 1].x, a3 = acc[i + 1].y;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        *(G<f32x4_a4> *)(o + 2 * i) = f32x4_a4{a0, a1, a2, a3};
      }
      if constexpr (R % 2 != 0) *(G<f32x2_a4> *)(o + 2 * (R - 1)) = f32x2_a4{acc[R - 1].x, acc[R - 1].y};
      return;
    }
#pragma unroll
    for (int i = 0; i < R; i++, o += C) {
      if (i < i_lo || i >= i_hi) continue;
      if constexpr (CT == 2)
        *(G<f32x2_a4> *)o = f32x2_a4{acc[i].x, acc[i].y};  // one 8-byte store (the buffer is only known 4-byte aligned)
      else
        o[0] = acc[i].x;
    }
  } else {
    g_i16 *o = out_ptr<int16_t>(d) + k0 * static_cast<int64_t>(C) + cg * CT;
    const bool aligned = CT == 2 && ((reinterpret_cast<uintptr_t>(d.out) | (C * 2u)) & 3u) == 0;
    if (ONE_GROUP && CT == 2 && aligned && i_lo == 0 && i_hi == R) {
      // the lane's R frames are R consecutive dwords: 16 + 16 + 8 bytes instead of R narrow
      // stores (each store instruction costs one line request per lane whatever its width)
      uint32_t v[R];
#pragma unroll
      for (int i = 0; i < R; i++) v[i] = round_pack_pcm(acc[i].x, acc[i].y);
      g_u32 *od = (g_u32 *)o;
      // (Round 5, measured and not kept: write-through stores -- global_store_dwordx4 ... sc1.  The eight XCDs' L2s are not
      //  coherent with each other, so the end of a kernel writes its dirty lines back (B / 6 TB/s at the boundary,
      //  MI355X_MICROARCH.md: 0.8 us for this launch's 4.6 MB), and with write-through stores nothing is dirty by then.
      //  But a lane's 16-byte pieces lie 640 bytes apart, and written through each is a fabric write of its own where
      //  L2 would have merged eight of them into a line: one stream 11.9 -> 15.7 us, 4 streams 33.4 -> 42.3,
      //  32 streams 206 -> 250 (same box, profiles/r05_ab_sc1.txt).)
#pragma unroll
      for (int i = 0; i + 4 <= R; i += 4) *(g_u32x4_a4 *)(od + i) = u32x4_a4{v[i], v[i + 1], v[i + 2], v[i + 3]};
      if constexpr (R % 4 >= 2) *(g_u32x2_a4 *)(od + R / 4 * 4) = u32x2_a4{v[R / 4 * 4], v[R / 4 * 4 + 1]};
      if constexpr (R % 2 != 0) od[R - 1] = v[R - 1];
      return;
    }
    // (8 channels: a lane's 4-byte pieces, 16 bytes apart, could leave as whole 16-byte frames after a 4 x 4
    //  transpose inside each quad of lanes -- rows cg, 4 + cg, 8 + cg per lane, 3 stores instead of 10.  With the
    //  transpose left out (wrong data, right addresses) BASELINE configs[3] at 32 streams went 588 -> 569 us, and
    //  the 48 vector instructions of three butterfly transposes cost about that much again: not built.)
#pragma unroll
    for (int i = 0; i < R; i++, o += C) {
      if (i < i_lo || i >= i_hi) continue;
      if (CT == 2) {
        const uint32_t v = round_pack_pcm(acc[i].x, acc[i].y);
        if (aligned) {
          *(g_u32 *)o = v;
        } else {
          o[0] = static_cast<int16_t>(v & 0xffffu);
          o[1] = static_cast<int16_t>(v >> 16);
        }
      } else {
        o[0] = static_cast<int16_t>(round_pack_pcm(acc[i].x, 0.f) & 0xffffu);
      }
    }
  }
}

// The stores of the ROWS mapping (lane_ctx): int16 frames of 8 channels.  Every lane of the wave runs this (the swaps
// move data between lanes whatever their state); `live` = the lane's period exists.  A block of four frames that lies
// wholly inside the call for every live lane (wave-uniform test) is transposed across the four rows by four swaps and
// leaves as one 16-byte store per lane; the last two frames of a group of ten as 8 bytes per lane; any other block --
// the call's first and last periods, the padding phases of the filter's last group -- as 4-byte pieces like store_group.
template <int R>
__device__ __forceinline__ void store_group_rows(const PeriodParams &p, const StreamDesc &d, const LaneCtx &c, uint32_t g,
                                                 bool live, const f32x2 (&acc)[R]) {
  static_assert(R == 10, "written for groups of ten phases");
  const int64_t k0 = static_cast<int64_t>(c.K_lane) + static_cast<int64_t>(g) * R - d.k_shift;
  const int64_t lo64 = k0 < 0 ? -k0 : 0;
  const int64_t hi64 = min(static_cast<int64_t>(R), min(static_cast<int64_t>(p.den) - static_cast<int64_t>(g) * R,
                                                       static_cast<int64_t>(d.n_out) - k0));
  const int i_lo = static_cast<int>(min(lo64, static_cast<int64_t>(R)));
  const int i_hi = static_cast<int>(max(hi64, static_cast<int64_t>(0)));
  uint32_t v[R];
#pragma unroll
  for (int i = 0; i < R; i++) v[i] = round_pack_pcm(acc[i].x, acc[i].y);
  const bool dword_ok = (reinterpret_cast<uintptr_t>(d.out) & 3u) == 0;
  g_i16 *frame0 = out_ptr<int16_t>(d) + k0 * 8;  // frame k0 of the call, channel 0
  const uint32_t row = c.cg;                     // this lane's row = the channel pair it computed
  auto narrow = [&](int first, int count) {      // frames [first, first + count) as the lane's own 4-byte pieces
#pragma unroll
    for (int i = first; i < first + count; i++) {
      if (!live || i < i_lo || i >= i_hi) continue;
      g_i16 *o = frame0 + i * 8 + row * 2;
      if (dword_ok) {
        *(g_u32 *)o = v[i];
      } else {
        o[0] = static_cast<int16_t>(v[i] & 0xffffu);
        o[1] = static_cast<int16_t>(v[i] >> 16);
      }
    }
  };
  auto whole = [&](int first, int count) {       // wave-uniform: no live lane has a frame of the block outside the call
    const bool cut = live && (i_lo > first || i_hi < first + count || !dword_ok);
    return __builtin_amdgcn_ballot_w64(cut) == 0;
  };
#pragma unroll
  for (int b = 0; b < 2; b++) {
    if (!whole(4 * b, 4)) {
      narrow(4 * b, 4);
      continue;
    }
    // rows hold columns: after the swaps row r holds frame 4b + r whole (w0..w3 = its four channel pairs)
    auto s02 = __builtin_amdgcn_permlane32_swap(v[4 * b], v[4 * b + 2], false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(v[4 * b + 1], v[4 * b + 3], false, false);
    auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    if (live) *(g_u32x4_a4 *)(frame0 + (4 * b + static_cast<int>(row)) * 8) = u32x4_a4{t01[0], t01[1], t23[0], t23[1]};
  }
  if (!whole(8, 2)) {
    narrow(8, 2);
  } else {
    // two frames over four rows: row r gets channel pairs 2 (r & 1), 2 (r & 1) + 1 of frame 8 + (r >> 1)
    auto s = __builtin_amdgcn_permlane32_swap(v[8], v[9], false, false);
    auto t = __builtin_amdgcn_permlane16_swap(s[0], s[1], false, false);
    if (live) *(g_u32x2_a4 *)(frame0 + (8 + static_cast<int>(row >> 1)) * 8 + (row & 1u) * 4) = u32x2_a4{t[0], t[1]};
  }
}

// Phase pairs (round 4; FirLoopAsmPP): a lane is (period, channel) -- one period, ONE channel of a frame of CF = 1, 2
// or 3 -- with 2R phases: acc[i] = phases 2i, 2i + 1 of group g.  rows: [group][trip][step][2R] floats; trips of 2 * steps_per_bank steps (2 for R = 10, 6 for R = 5).
// part / parts: tap-range shares (fir_tile_parts): this wave runs trips [total*part/parts, total*(part+1)/parts).
template <int R, int CF, bool PADDED, bool W16>
__device__ __forceinline__ void fir_group_pp(const PeriodParams &p, const float *__restrict__ rows, const float *xs,
                                             const LaneCtx &c, uint32_t g, bool skip_all, uint32_t part, uint32_t parts,
                                             f32x2 (&acc)[R]) {
  using Isa = FirLoopAsmPP<R, CF, PADDED, W16>;
  static_assert(Isa::available, "phase pairs run their ISA loop");
  constexpr uint32_t kStepsPerTrip = 2 * Isa::steps_per_bank;
  constexpr uint32_t kPerIt = 4 / kStepsPerTrip;  // (the padded plan's boundary tables count 4-step iterations)
  constexpr uint32_t EB = W16 ? 2u : 4u;
  auto sgpr = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
  const uint32_t delta_g = p.delta[g];
  const uint32_t trips = skip_all ? 0u : p.delta[2 * p.groups + g];  // head | tail << 4 | total << 8
  const uint32_t head = R == 10 ? trips & 15u : 0u, tail = R == 10 ? (trips >> 4) & 15u : 0u, total = trips >> 8;
  const uint32_t t0 = sgpr(total * part / parts), t1 = sgpr(total * (part + 1) / parts);
  auto overlap = [&](uint32_t lo, uint32_t hi) {
    const uint32_t a = max(t0, lo), b = min(t1, hi);
    return b > a ? b - a : 0u;
  };
  const uint32_t main_end = total - tail;
  uint32_t wraps = 0, to_wrap = 0;
  const uint32_t wrap_step = p.wrap_step * kPerIt;
  if constexpr (PADDED) {
    const uint32_t to_wrap0 = p.delta[p.groups + g] * kPerIt;
    to_wrap = to_wrap0;
    if (to_wrap0 != 0 && t0 >= to_wrap0) {
      const uint32_t past = t0 - to_wrap0;
      wraps = 1 + past / wrap_step;
      to_wrap = wrap_step - past % wrap_step;
    } else if (to_wrap0 != 0) {
      to_wrap = to_wrap0 - t0;
    }
  }
  const float *rows_g = rows + (static_cast<size_t>(g) * p.l4 + t0) * (kStepsPerTrip * 2 * R);
  const uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xs)) +
                        ((c.xlane + delta_g * CF) + t0 * kStepsPerTrip * CF + wraps * p.pad) * EB;
  const uint64_t rows_bits = reinterpret_cast<uint64_t>(rows_g);
  const float *rows_s = reinterpret_cast<const float *>(static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits))) |
                                                        static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits >> 32))) << 32);
  Isa::run(acc, rows_s, addr, sgpr(overlap(0, head)), sgpr(overlap(head, main_end)), sgpr(overlap(main_end, total)), sgpr(to_wrap),
           sgpr(wrap_step), sgpr((kStepsPerTrip * CF + p.pad) * EB));
}

// ... and its stores.  The lane's 2R phases of group g are 2R consecutive FRAMES of its channel.  Mono: 2R consecutive
// samples -- whole dwords around at most two odd samples (the run may start on the upper half of a dword: k_shift
// odd, an odd den), float runs as 16-byte pieces.  Stereo: the two lanes of a frame (left, right: neighbours) swap
// halves by DPP (quad_perm [1,0,3,2]) so that the left lane holds the first R frames whole and the right lane the
// last R: R dwords (int16) or 2R floats each.  Three channels, and any run cut by the call's ends or the last group's
// padding phases: sample by sample.
__device__ __forceinline__ uint32_t pp_pair_swap(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xF, 0xF, true));
}
template <int R, int CF, typename T>
__device__ __forceinline__ void store_group_pp(const PeriodParams &p, const StreamDesc &d, const LaneCtx &c, uint32_t g,
                                               const f32x2 (&acc)[R]) {
  constexpr int N = 2 * R;
  const int64_t k0 = static_cast<int64_t>(c.K_lane) + static_cast<int64_t>(g) * N - d.k_shift;
  const int64_t lo64 = k0 < 0 ? -k0 : 0;
  const int64_t hi64 = min(static_cast<int64_t>(N), min(static_cast<int64_t>(p.den) - static_cast<int64_t>(g) * N,
                                                       static_cast<int64_t>(d.n_out) - k0));
  G<T> *o = out_ptr<T>(d) + k0 * CF + c.cg;
  if constexpr (CF == 2) {
    // (both lanes of a pair take this branch or neither: same period, same group, same call)
    if (lo64 == 0 && hi64 == N && (reinterpret_cast<uintptr_t>(d.out) & 3u) == 0) {
      const bool right = c.cg != 0;
      if constexpr (sizeof(T) == 2) {
        int mine[N];
#pragma unroll
        for (int i = 0; i < R; i++) {
          asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(mine[2 * i]) : "v"(acc[i].x));
          asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(mine[2 * i + 1]) : "v"(acc[i].y));
        }
        uint32_t w[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
          const int theirs = static_cast<int>(pp_pair_swap(static_cast<uint32_t>(right ? mine[q] : mine[R + q])));
          typedef short short2_t __attribute__((ext_vector_type(2)));
          const short2_t pk = right ? __builtin_amdgcn_cvt_pk_i16(theirs, mine[R + q]) : __builtin_amdgcn_cvt_pk_i16(mine[q], theirs);
          w[q] = __builtin_bit_cast(uint32_t, pk);
        }
        g_u32 *od = (g_u32 *)(out_ptr<int16_t>(d) + (k0 + (right ? R : 0)) * 2);
#pragma unroll
        for (int j = 0; j + 4 <= R; j += 4) *(g_u32x4_a4 *)(od + j) = u32x4_a4{w[j], w[j + 1], w[j + 2], w[j + 3]};
        if constexpr (R % 4 >= 2) *(g_u32x2_a4 *)(od + R / 4 * 4) = u32x2_a4{w[R / 4 * 4], w[R / 4 * 4 + 1]};
        if constexpr (R % 2 != 0) od[R - 1] = w[R - 1];
      } else {
        float mine[N];
#pragma unroll
        for (int i = 0; i < R; i++) {
          mine[2 * i] = acc[i].x;
          mine[2 * i + 1] = acc[i].y;
        }
        G<float> *of = out_ptr<float>(d) + (k0 + (right ? R : 0)) * 2;
#pragma unroll
        for (int q = 0; q + 1 < R; q += 2) {
          const float t0 = __uint_as_float(pp_pair_swap(__float_as_uint(right ? mine[q] : mine[R + q])));
          const float t1 = __uint_as_float(pp_pair_swap(__float_as_uint(right ? mine[q + 1] : mine[R + q + 1])));
          *(G<f32x4_a4> *)(of + 2 * q) = right ? f32x4_a4{t0, mine[R + q], t1, mine[R + q + 1]} : f32x4_a4{mine[q], t0, mine[q + 1], t1};
        }
        if constexpr (R % 2 != 0) {
          const float t0 = __uint_as_float(pp_pair_swap(__float_as_uint(right ? mine[R - 1] : mine[N - 1])));
          *(G<f32x2_a4> *)(of + 2 * (R - 1)) = right ? f32x2_a4{t0, mine[N - 1]} : f32x2_a4{mine[R - 1], t0};
        }
      }
      return;
    }
  }
  if constexpr (CF == 1 && sizeof(T) == 2) {
    if (lo64 == 0 && hi64 == N && (reinterpret_cast<uintptr_t>(d.out) & 1u) == 0) {
      g_u32 *od;
      if ((reinterpret_cast<uintptr_t>(o) & 2u) == 0) {
        uint32_t w[R];
#pragma unroll
        for (int i = 0; i < R; i++) w[i] = round_pack_pcm(acc[i].x, acc[i].y);
        od = (g_u32 *)o;
#pragma unroll
        for (int j = 0; j + 4 <= R; j += 4) *(g_u32x4_a4 *)(od + j) = u32x4_a4{w[j], w[j + 1], w[j + 2], w[j + 3]};
        if constexpr (R % 4 >= 2) *(g_u32x2_a4 *)(od + R / 4 * 4) = u32x2_a4{w[R / 4 * 4], w[R / 4 * 4 + 1]};
        if constexpr (R % 2 != 0) od[R - 1] = w[R - 1];
      } else {
        // the run starts on the upper half of a dword: sample 0, the R - 1 dwords that straddle the pairs, sample N - 1
        o[0] = static_cast<int16_t>(round_pack_pcm(acc[0].x, 0.f) & 0xffffu);
        uint32_t w[R - 1];
#pragma unroll
        for (int i = 0; i + 1 < R; i++) w[i] = round_pack_pcm(acc[i].y, acc[i + 1].x);
        od = (g_u32 *)(o + 1);
#pragma unroll
        for (int j = 0; j + 4 <= R - 1; j += 4) *(g_u32x4_a4 *)(od + j) = u32x4_a4{w[j], w[j + 1], w[j + 2], w[j + 3]};
        if constexpr ((R - 1) % 4 >= 2) *(g_u32x2_a4 *)(od + (R - 1) / 4 * 4) = u32x2_a4{w[(R - 1) / 4 * 4], w[(R - 1) / 4 * 4 + 1]};
        if constexpr ((R - 1) % 2 != 0) od[R - 2] = w[R - 2];
        o[N - 1] = static_cast<int16_t>(round_pack_pcm(acc[R - 1].y, 0.f) & 0xffffu);
      }
      return;
    }
  } else if constexpr (CF == 1) {
    if (lo64 == 0 && hi64 == N) {
      G<float> *of = (G<float> *)o;
#pragma unroll
      for (int i = 0; i + 1 < R; i += 2) {
        float a0 = acc[i].x, a1 = acc[i].y, a2 = acc[i + 1].x, a3 = acc[i + 1].y;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));  // (opaque copies, as for the stereo float stores)
        *(G<f32x4_a4> *)(of + 2 * i) = f32x4_a4{a0, a1, a2, a3};
      }
      if constexpr (R % 2 != 0) *(G<f32x2_a4> *)(of + 2 * (R - 1)) = f32x2_a4{acc[R - 1].x, acc[R - 1].y};
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (i < lo64 || i >= hi64) continue;
    const float v = (i & 1) ? acc[i / 2].y : acc[i / 2].x;
    if constexpr (sizeof(T) == 4)
      o[i * CF] = v;
    else
      o[i * CF] = static_cast<int16_t>(round_pack_pcm(v, 0.f) & 0xffffu);
  }
}

// The kernel's parameters and the workgroup's descriptor as they lie in memory (the kernel-argument segment; the
// descriptor ring for large batches), in the constant address space: what is read through these comes by scalar
// loads.  The ISA loop names 40 tap SGPRs; with the parameters and the descriptor held in registers across it
// hipcc spilled ~26 SGPRs into VGPR lanes around every group (v_writelane / v_readlane: ~60 vector instructions
// per wave, as many as the conversions of the staging).  fir_tile therefore reads both AGAIN on either side of
// the loop, through pointers made opaque so that the compiler cannot keep the first copies alive instead:
// a handful of scalar loads that hit the scalar cache.
typedef const __attribute__((address_space(4))) PeriodParams *KParams;
typedef const __attribute__((address_space(4))) StreamDesc *KDesc;
template <typename P>
__device__ __forceinline__ P opaque(P ptr) {
  asm volatile("" : "+s"(ptr));
  return ptr;
}
template <typename X>
__device__ __forceinline__ X load_k(const __attribute__((address_space(4))) X *ptr) {
  // (dword by dword: a struct cannot be copy-constructed from another address space; unused fields' loads vanish)
  static_assert(sizeof(X) % 4 == 0, "dword-sized structs");
  union {
    X v;
    uint32_t w[sizeof(X) / 4];
  } u;
  const __attribute__((address_space(4))) uint32_t *q = (const __attribute__((address_space(4))) uint32_t *)opaque(ptr);
#pragma unroll
  for (size_t i = 0; i < sizeof(X) / 4; i++) u.w[i] = q[i];
  return u.v;
}
struct KernArgs {  // layout of resample_period's kernel arguments
  PeriodParams p;
  const float *rows;
  DescPack pack;
};

// FIR of one tile (m_cnt periods starting at m_lo) for the phase groups owned by this wave,
// followed by round / interleave / store.  `zsplit` of `nsplit` workgroups share the tile's groups.
// AM: arithmetic of the FIR loop -- 0 = fp32 chain (FirLoopAsm or the C++ loop), 1 = fp64 accumulator (FirLoopAsm64),
// 2 = phase pairs for single-channel lanes (FirLoopAsmPP)
template <int R, int CT, bool ONE_GROUP, bool PADDED, typename T, int CGF = 0, bool W16 = false, int AM = 0>
__device__ __forceinline__ void fir_tile(const PeriodParams &p0, const StreamDesc &d0, KParams pp,
                                         const float *__restrict__ rows, KDesc dp, const float *xs, uint32_t xshift,
                                         uint32_t m_lo, uint32_t m_cnt, uint32_t wave, uint32_t lane, uint32_t zsplit,
                                         uint32_t nsplit) {
  // (layouts that run the C++ loop keep the copies the kernel already holds: re-reading them there only added
  //  register pressure -- scratch in every such instance)
#ifndef SPEEXHIP_CXX_FIR_LOOP
  constexpr bool kReload = AM != 0 || FirLoopAsm<R, CT, ONE_GROUP ? CT : CT * CGF, PADDED, W16>::available;
#else
  constexpr bool kReload = AM != 0;
#endif
  auto params = [&]() -> PeriodParams {
    if constexpr (kReload) return load_k(pp);
    return p0;
  };
  // (the first group runs on the copy the prologue holds -- it dies at the loop, nothing later reads it --, every
  //  further group on the copy read behind the previous one's loop: a read in FRONT of the first loop as well put
  //  a scalar-load round trip between the staging barrier and the FIR of every workgroup, ~0.2 us that a
  //  one-generation launch cannot hide)
  constexpr bool kRows = PADDED && CGF == 4 && !W16 && AM == 0;  // (lane_ctx: rows of 16 lanes = channel pairs)
  const LaneCtx c = lane_ctx<CT, ONE_GROUP, PADDED, CGF, AM == 2, kRows>(p0, xshift, m_lo, m_cnt, lane);
  const uint32_t g_step = p0.wave_groups * nsplit;
  uint32_t g = zsplit * p0.wave_groups + wave;
  if (g >= p0.groups) return;
  PeriodParams p = p0;
  for (;;) {
    f32x2 acc[R];  // .x = first channel of the pair, .y = second (unused when CT == 1)
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] = f32x2{0.f, 0.f};
#ifdef SPEEXHIP_STAMPS
    const unsigned long long fir_t0 = __builtin_amdgcn_s_memtime(), fir_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (AM == 2) {
      fir_group_pp<R, ONE_GROUP ? 1 : CGF, PADDED, W16>(p, rows, xs, c, g, SPEEXHIP_DIAG_SKIP(p, 4u), 0u, 1u, acc);
    } else if constexpr (AM == 1) {
      // fp64 sums, then the reference's store of its double sum into a float (resample.c:417, :544)
      double acc64[R][2];
#pragma unroll
      for (int i = 0; i < R; i++) acc64[i][0] = acc64[i][1] = 0.0;
      fir_group64<R, CT, PADDED, ONE_GROUP ? CT : CT * CGF, W16>(p, reinterpret_cast<const double *>(rows), xs, c, g,
                                                                SPEEXHIP_DIAG_SKIP(p, 4u), 0u, 1u, acc64);
#pragma unroll
      for (int i = 0; i < R; i++) acc[i] = f32x2{static_cast<float>(acc64[i][0]), static_cast<float>(acc64[i][1])};
    } else {
      fir_group<R, CT, PADDED, ONE_GROUP ? CT : CT * CGF, W16>(p, rows, xs, c, g, SPEEXHIP_DIAG_SKIP(p, 4u), acc);
    }
#ifdef SPEEXHIP_STAMPS
    {
      asm volatile("" ::"v"(acc[0]));
      const unsigned long long fir_t1 = __builtin_amdgcn_s_memtime(), fir_r1 = __builtin_amdgcn_s_memrealtime();
      const uint32_t lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      if ((threadIdx.x & 63u) == 0 && lin_ < 8192) atomicMax(&g_stamps[lin_ * 16 + 7], fir_t1 - fir_t0);
      if (threadIdx.x == 0 && lin_ < 8192) {  // wave 0: shader cycles and 100 MHz ticks of the same interval
        g_stamps[lin_ * 16 + 9] = fir_t1 - fir_t0;
        g_stamps[lin_ * 16 + 10] = fir_r1 - fir_r0;
      }
    }
#endif
    STAMP(5);
    const PeriodParams q = params();  // (... and the far side)
    if constexpr (kRows && sizeof(T) == 2 && R == 10) {
      if (!SPEEXHIP_DIAG_SKIP(q, 8u)) {  // (every lane: the transposes exchange data between the rows of the wave)
        StreamDesc d;
        if constexpr (kReload)
          d = load_k(dp);
        else
          d = d0;
        if (q.prio & 2u) __builtin_amdgcn_s_setprio(2);
        store_group_rows<R>(q, d, c, g, c.live, acc);
        if (q.prio & 2u) set_fir_priority(q);
        STAMP(6);
      }
    } else if (!SPEEXHIP_DIAG_SKIP(q, 8u) && c.live) {
      StreamDesc d;
      if constexpr (kReload)
        d = load_k(dp);
      else
        d = d0;
      if (q.prio & 2u) __builtin_amdgcn_s_setprio(2);
      if constexpr (AM == 2)
        store_group_pp<R, ONE_GROUP ? 1 : CGF, T>(q, d, c, g, acc);
      else
        store_group<R, CT, ONE_GROUP, T>(q, d, c, g, acc);
      if (q.prio & 2u) set_fir_priority(q);
      STAMP(6);
    }
    g += g_step;
    if (g >= q.groups) return;
    p = q;
  }
}

// ---- tap-range shares (round 3, second take) -------------------------------------------------------
// A launch that cannot fill the chip runs its tiles in shares of a few phase groups each, and a share's
// few FIR waves then sit one to a SIMD: a wave alone waits out every scalar-load round trip (14-19 cycles
// per packed FMA instead of 4.5-5).  For long filters -- the decimators: 48k -> 22.05k 304 steps,
// 48k -> 11.025k 604, 44.1k -> 8k 744 -- that wave's R x steps FMAs are the launch (48k -> 11.025k stereo,
// one stream: 33 us for 48 000 frames as for 2^20).  Here the `parts` waves of a group each take a range
// of its trips (rows pointer, window address and padding count-down advanced to the range's first trip),
// the partial sums meet in the dead window behind a barrier, and part 0 stores.  Only where a wave's chain is
// long: for BASELINE configs[1] the same scheme lost (two barriers and a pass through LDS against
// 640 FMAs per wave: 13.22 -> 13.94 us, profiles/r03_ab_ksplit.txt); see launch_period_plan for the rule.
template <int R, int CT, bool PADDED, int CF, bool W16>
__device__ __forceinline__ void fir_group_part(const PeriodParams &p, const float *__restrict__ rows, const float *xs,
                                               const LaneCtx &c, uint32_t g, uint32_t part, uint32_t parts,
                                               f32x2 (&acc)[R]) {
  using Isa = FirLoopAsm<R, CT, CF, PADDED, W16>;
  static_assert(CF != 0 && Isa::available, "tap-range shares run the ISA loop");
  constexpr uint32_t kStepsPerTrip = 2 * Isa::steps_per_bank;
  constexpr uint32_t EB = W16 ? 2u : 4u;  // bytes per window element
  auto sgpr = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
  const uint32_t delta_g = p.delta[g];
  const uint32_t trips = SPEEXHIP_DIAG_SKIP(p, 4u) ? 0u : p.delta[2 * p.groups + g];  // head | tail << 4 | total << 8
  const uint32_t head = R == 10 ? trips & 15u : 0u, tail = R == 10 ? (trips >> 4) & 15u : 0u, total = trips >> 8;
  const uint32_t t0 = sgpr(total * part / parts), t1 = sgpr(total * (part + 1) / parts);
  auto overlap = [&](uint32_t lo, uint32_t hi) {  // trips of [t0, t1) inside [lo, hi)
    const uint32_t a = max(t0, lo), b = min(t1, hi);
    return b > a ? b - a : 0u;
  };
  const uint32_t main_end = total - tail;
  // the padded walk: a boundary behind trip to_wrap0, then every wrap_step trips (0 = none)
  uint32_t wraps = 0, to_wrap = 0;
  if constexpr (PADDED) {
    const uint32_t to_wrap0 = p.delta[p.groups + g];
    to_wrap = to_wrap0;
    if (to_wrap0 != 0 && t0 >= to_wrap0) {
      const uint32_t past = t0 - to_wrap0;
      wraps = 1 + past / p.wrap_step;
      to_wrap = p.wrap_step - past % p.wrap_step;
    } else if (to_wrap0 != 0) {
      to_wrap = to_wrap0 - t0;
    }
  }
  const float *rows_g = rows + (static_cast<size_t>(g) * p.l4 + t0) * (2 * bank_taps(R));
  const uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xs)) +
                        ((c.xlane + delta_g * c.C) + t0 * kStepsPerTrip * CF + wraps * p.pad) * EB;
  const uint64_t rows_bits = reinterpret_cast<uint64_t>(rows_g);
  const float *rows_s = reinterpret_cast<const float *>(static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits))) |
                                                        static_cast<uint64_t>(sgpr(static_cast<uint32_t>(rows_bits >> 32))) << 32);
  Isa::run(acc, rows_s, addr, CT == 1 ? addr + p.half_offset * EB : 0u, sgpr(overlap(0, head)), sgpr(overlap(head, main_end)),
           sgpr(overlap(main_end, total)), sgpr(to_wrap), sgpr(p.wrap_step), sgpr((kStepsPerTrip * CF + p.pad) * EB));
}

// One group per wave-set (the host launches tap-range shares only when a share's groups fit its waves:
// the partial sums overwrite the window).  Wave w: group w % wave_groups of the share, part w / wave_groups.
template <int R, int CT, bool ONE_GROUP, bool PADDED, typename T, int CGF = 0, bool W16 = false, bool PP = false>
__device__ __forceinline__ void fir_tile_parts(KParams pp, const float *__restrict__ rows, KDesc dp, float *xs,
                                               uint32_t xshift, uint32_t m_lo, uint32_t m_cnt, uint32_t wave, uint32_t lane,
                                               uint32_t zsplit) {
  constexpr int CF = ONE_GROUP ? CT : CT * CGF;
  LaneCtx c;
  uint32_t g, part, gw, wg, parts;
  bool valid;
  f32x2 acc[R];
#pragma unroll
  for (int i = 0; i < R; i++) acc[i] = f32x2{0.f, 0.f};
  {
    const PeriodParams p = load_k(pp);
    c = lane_ctx<CT, ONE_GROUP, PADDED, CGF, PP, PADDED && CGF == 4 && !W16 && !PP>(p, xshift, m_lo, m_cnt, lane);
    wg = p.wave_groups;
    parts = p.ksplit;
    part = 0;
    gw = wave;
    while (gw >= wg) {  // (wave-uniform; parts <= 16)
      gw -= wg;
      part++;
    }
    g = zsplit * wg + gw;
    valid = g < p.groups;
    if constexpr (PP) {
      if (valid) fir_group_pp<R, ONE_GROUP ? 1 : CGF, PADDED, W16>(p, rows, xs, c, g, SPEEXHIP_DIAG_SKIP(p, 4u), part, parts, acc);
    } else {
      if (valid) fir_group_part<R, CT, PADDED, CF, W16>(p, rows, xs, c, g, part, parts, acc);
    }
  }
#ifdef SPEEXHIP_STAMPS
  asm volatile("" ::"v"(acc[0]));
#endif
  STAMP(5);    // (the last wave out of its FIR loop)
  STAMP_FIRST(13);  // (the first wave out of its FIR loop: stored as ~time, so that atomicMax keeps the earliest)
  __syncthreads();  // every wave is done with the window
  STAMP(11);
  // partial sums of part j >= 1, group-wave gw: block (j - 1) * wg + gw of R x 64 pairs, lanes side by side
  f32x2 *sums = reinterpret_cast<f32x2 *>(xs);
  if (valid && part != 0) {
    f32x2 *mine = sums + (static_cast<size_t>(part - 1) * wg + gw) * (R * 64) + lane;
#pragma unroll
    for (int i = 0; i < R; i++) mine[i * 64] = acc[i];
  }
  __syncthreads();
  if (!valid || part != 0) return;
  for (uint32_t j = 1; j < parts; j++) {
    const f32x2 *theirs = sums + (static_cast<size_t>(j - 1) * wg + gw) * (R * 64) + lane;
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] += theirs[i * 64];
  }
  STAMP(12);  // (partial sums added)
  const PeriodParams q = load_k(pp);
  if constexpr (PADDED && CGF == 4 && !W16 && !PP && sizeof(T) == 2 && R == 10) {
    if (SPEEXHIP_DIAG_SKIP(q, 8u)) return;
    const StreamDesc d = load_k(dp);
    store_group_rows<R>(q, d, c, g, c.live, acc);  // (every lane of the wave: see there)
    STAMP(6);
    return;
  }
  if (SPEEXHIP_DIAG_SKIP(q, 8u) || !c.live) return;
  const StreamDesc d = load_k(dp);
  if constexpr (PP)
    store_group_pp<R, ONE_GROUP ? 1 : CGF, T>(q, d, c, g, acc);
  else
    store_group<R, CT, ONE_GROUP, T>(q, d, c, g, acc);
  STAMP(6);
}

// ... with an fp64 accumulator: the partial sums meet in LDS as doubles (twice the room: launch_period_plan)
template <int R, int CT, bool ONE_GROUP, bool PADDED, typename T, int CGF = 0, bool W16 = false>
__device__ __forceinline__ void fir_tile_parts64(KParams pp, const double *__restrict__ rows, KDesc dp, float *xs, uint32_t xshift,
                                                 uint32_t m_lo, uint32_t m_cnt, uint32_t wave, uint32_t lane, uint32_t zsplit) {
  constexpr int CF = ONE_GROUP ? CT : CT * CGF;
  LaneCtx c;
  uint32_t g, part, gw, wg, parts;
  bool valid;
  double acc[R][2];
#pragma unroll
  for (int i = 0; i < R; i++) acc[i][0] = acc[i][1] = 0.0;
  {
    const PeriodParams p = load_k(pp);
    c = lane_ctx<CT, ONE_GROUP, PADDED, CGF>(p, xshift, m_lo, m_cnt, lane);
    wg = p.wave_groups;
    parts = p.ksplit;
    part = 0;
    gw = wave;
    while (gw >= wg) {  // (wave-uniform; parts <= 16)
      gw -= wg;
      part++;
    }
    g = zsplit * wg + gw;
    valid = g < p.groups;
    if (valid) fir_group64<R, CT, PADDED, CF, W16>(p, rows, xs, c, g, SPEEXHIP_DIAG_SKIP(p, 4u), part, parts, acc);
  }
  __syncthreads();  // every wave is done with the window
  double *sums = reinterpret_cast<double *>(xs);
  if (valid && part != 0) {
    double *mine = sums + (static_cast<size_t>(part - 1) * wg + gw) * (2 * R * 64) + lane;
#pragma unroll
    for (int i = 0; i < R; i++) {
      mine[(2 * i) * 64] = acc[i][0];
      mine[(2 * i + 1) * 64] = acc[i][1];
    }
  }
  __syncthreads();
  if (!valid || part != 0) return;
  for (uint32_t j = 1; j < parts; j++) {
    const double *theirs = sums + (static_cast<size_t>(j - 1) * wg + gw) * (2 * R * 64) + lane;
#pragma unroll
    for (int i = 0; i < R; i++) {
      acc[i][0] += theirs[(2 * i) * 64];
      acc[i][1] += theirs[(2 * i + 1) * 64];
    }
  }
  const PeriodParams q = load_k(pp);
  if (SPEEXHIP_DIAG_SKIP(q, 8u) || !c.live) return;
  const StreamDesc d = load_k(dp);
  f32x2 out[R];
#pragma unroll
  for (int i = 0; i < R; i++) out[i] = f32x2{static_cast<float>(acc[i][0]), static_cast<float>(acc[i][1])};
  store_group<R, CT, ONE_GROUP, T>(q, d, c, g, out);
}

// The tap rows on their way into L2 while the window is staged (round 4).  The FIR loop reads its taps with scalar loads,
// one bank in flight per wave, and the waves of a launch walk their rows in step: a row line that is not in L2 costs
// every wave that reads it the trip to HBM -- ~400 lines per wave x ~250 ns in a launch of one generation whose rows
// the previous launch's samples pushed out (or that nobody has read yet: every first call).  32 streams x 131 072 frames
// of stereo 48k -> 11.025k in phase pairs: FIR 155 us behind a staged window, 50 us with the window left unstaged and
// the rows still in L2 from the launch before; rocprofv3, which runs every launch on cold caches, 137 us either way
// (profiles/r04_skips_2ch_pp.txt, r04_pmc_skip.txt; a 32 MB read between two launches does the same, 24 MB does not: the
// eight L2s hold 32 MB -- profiles/r04_cold_state.txt).  One dword per 128-byte line of ALL the rows (<= ~1 MB: a few
// loads per lane, issued once the window is in LDS, waited for before the FIR loop starts: one trip to HBM, ~2 us, for
// the ~400 in a row that the loop would make; workgroups of a later generation find the lines in L2 already).
template <int R>
__device__ __forceinline__ void touch_rows(const PeriodParams &p, const float *__restrict__ rows, uint32_t &sink) {
  if (p.touch == 0) return;  // (the host's rule: launch_period_plan)
  const uint32_t bytes = p.groups * p.l4 * static_cast<uint32_t>(2 * bank_taps(R) * 4);
  // (one statement, its result tied to its input: the loads all land in the one register the caller keeps until they have)
#pragma clang loop unroll(disable)
  for (uint32_t off = threadIdx.x * 128u; off < bytes; off += p.threads * 128u)
    asm volatile("global_load_dword %0, %1, %2" : "+v"(sink) : "v"(off), "s"(rows) : "memory");
}

// (Mono int16 left through an LDS image -- one row per period, whole rows written 16 bytes per lane -- from
//  round 1 to round 3 for launches that fill the chip: 160 -> 138 us for 32 streams of 44.1k -> 48k when a lane's
//  20-byte runs cost five stores.  With the runs packed into dwords at either alignment (store_group) the
//  per-lane stores are faster at every size: 32 streams 122 -> 112 us, 16 streams 72.5 -> 62.7, 8 streams
//  42.4 -> 33.5, 64 streams 229 -> 210, 44.1k -> 8k 176 -> 165, q10 199 -> 192; the image path is gone
//  (profiles/r03_mono_stores_ab.txt).  Its stereo and multi-pair counterparts never paid: 40-byte pieces per
//  lane, stores alone 74 -> 48 us but the launch within noise; 4 ch 431 -> 421 us, 8 ch 621 -> 644, 6 ch 670 -> 735.)
// <= 80 SGPRs: the hardware admits 8 waves per SIMD (two 16-wave workgroups per CU) only then
// (MI355X_MICROARCH.md, residency; measured again with caps of 88, 90 and 96: 206 -> 260 us);
// the compiler alone settles at ~106.  (A second, uncapped build of the kernel for launches of at
// most one workgroup per CU was tried: removing the cap from this kernel gave 13.36 -> 13.0 us on
// one stream, but as a separate __global__ around a shared device body it measured 13.59 vs 13.53 us,
// i.e. nothing, and the refactoring cost the capped kernel 0.17 us -- not kept.)
// The R = 5 instances only ever run one workgroup per CU (launch_period): no 64-VGPR limit for them -- nor for the
// tap-range-share instances (KS), which are launches of one generation by construction.
//
// Workgroup = one tile (blockIdx.x) of one stream (blockIdx.y), optionally one of gridDim.z
// shares of its phase groups.
template <int R, int CT, bool ONE_GROUP, bool PADDED, typename T, int CGF = 0, bool W16 = false, bool KS = false, int AM = 0>
__global__ __launch_bounds__(1024, (R == 10 && !KS) ? 8 : 4) __attribute__((amdgpu_num_sgpr(80))) void resample_period(
    PeriodParams p, const float *__restrict__ rows, DescPack pack) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  // A workgroup that starts beside another one's FIR loop competes with 16 older waves for every
  // issue slot: its few hundred prologue and staging instructions -- the ones that put its window
  // loads in flight -- took 4-5 us there (0.5 us on an idle CU).  They run at raised priority; the
  // FIR loop and everything after it at the default.
  // (unconditionally: behind a test of p.prio the compiler holds back the descriptor's loads -- whose address
  //  needs nothing but blockIdx.y -- until p.prio has arrived, a second memory round trip of ~0.4 us in every
  //  workgroup's prologue; the diagnostics knob lowers the priority again as soon as it is known)
  __builtin_amdgcn_s_setprio(3);
  STAMP(0);
#ifdef SPEEXHIP_STAMPS
  {
    const uint32_t lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && lin_ < 8192)
      g_stamps[lin_ * 16 + 8] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11))) << 32) |
                                __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  }
#endif
  // Everything up to the window geometry is computed BEFORE the first branch (on whatever values an
  // exiting workgroup happens to have: pure arithmetic, no memory access): the kernel arguments and
  // the descriptor then arrive through one batch of scalar loads and a single wait instead of one
  // round trip per early exit (four dependent waits, ~1.0 us from start to the first staging load).
  const StreamDesc d = pack.d[blockIdx.y];
  if (!(p.prio & 1u)) __builtin_amdgcn_s_setprio(0);  // diagnostics: A/B of the raised prologue priority
  const uint32_t m_total = d.m_total;  // periods touched by this call: ceil((k_shift + n_out) / den), from the host
  const uint32_t m_lo = blockIdx.x * p.lane_periods;
  const uint32_t m_cnt = m_lo < m_total ? min(p.lane_periods, m_total - m_lo) : 1u;
  // (most tiles lie wholly inside the call's input: a dozen scalar instructions give their geometry; the
  //  general form -- history in front, silence behind, unaligned buffers -- costs ~150 and was a third of
  //  the 1.2-1.4 us a workgroup took from its first instruction to its first staging load)
  WindowGeom wg;
  if (!window_geom_plain<T>(d, p.channels, p.num, p.tail_frames, m_lo, m_cnt, p.threads, PADDED ? p.pad : 0u,
                            PADDED ? p.period_magic : 0u, &wg))
    wg = window_geom<T>(d, p.channels, p.num, p.tail_frames, m_lo, m_cnt, p.threads, PADDED ? p.pad : 0u,
                        PADDED ? p.period_magic : 0u);
  if (SPEEXHIP_DIAG_SKIP(p, 64u)) return;  // diagnostics: bare dispatch cost
  if (blockIdx.x == p.history_block) {
    if (blockIdx.z == 0) roll_history<T>(p.channels, d, p.threads);
    return;
  }
  if (d.n_out == 0 || blockIdx.x > p.history_block || m_lo >= m_total) return;
  STAMP(1);
  uint32_t row_sink = 0;
  if (!SPEEXHIP_DIAG_SKIP(p, 2u)) {
    // 5 x 16 bytes per lane in flight: a 76 KB window staged by 1024 lanes in one round of loads
    // (the padded commit needs more registers per group: 3 there keeps the kernel at 8 waves per SIMD)
    // (float samples: 4 -- five float groups in flight spill at the 64 VGPRs of 8 waves per SIMD)
    constexpr int UNR = PADDED ? 3 : (sizeof(T) == 4 ? (ONE_GROUP ? 5 : 3) : 5);
    u32x4 w[UNR];
    bool plain = false, plain_padded = false;
    if constexpr (!PADDED) plain = window_is_plain<UNR, T>(wg);  // (wave-uniform)
    if constexpr (PADDED && !W16) plain_padded = window_is_plain_padded<UNR, T>(wg);
    if (plain_padded) {
      window_fetch_plain<UNR, T>(wg, w);
      STAMP(2);
      window_commit_plain_padded<UNR, T>(xs, wg, w);
    } else if (plain) {
      window_fetch_plain<UNR, T>(wg, w);
      STAMP(2);
      if constexpr (W16)
        window_commit_plain16<UNR>(reinterpret_cast<int16_t *>(xs), wg, w);
      else
        window_commit_plain<UNR, T>(xs, wg, w);
    } else {
      window_fetch<UNR, T>(wg, w);
      STAMP(2);
      if constexpr (W16)
        window_commit16<UNR>(reinterpret_cast<int16_t *>(xs), d, wg, w);
      else
        window_commit<UNR, T>(xs, d, wg, w);
    }
    // The tap rows into L2 BEHIND the window: fetched in front of the window's loads, or between those loads and
    // their use, the lines are gone again when the FIR loop asks for them (the samples streaming in replace them);
    // fetched once the window is in LDS they stay (profiles/r04_cold_state.txt).  Instances of the ISA loops only: in
    // the ones that run the C++ loop the statement alone -- never executed -- cost 20 VGPRs and 130-230 bytes of
    // scratch, three channels 44.1k->48k 64 -> 754 us; tests/test_gpu_perf_gate.py caught it.
#ifndef SPEEXHIP_CXX_FIR_LOOP
    if constexpr (AM != 0 || FirLoopAsm<R, CT, ONE_GROUP ? CT : CT * CGF, PADDED, W16>::available) touch_rows<R>(p, rows, row_sink);
#endif
    // (the loads of touch_rows land in row_sink: nothing reads it, the register stays reserved until they have)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(row_sink));
  }
  STAMP(3);
  __syncthreads();
  if (p.prio & 7u) set_fir_priority(p);
  STAMP(4);
  if (SPEEXHIP_DIAG_SKIP(p, 128u)) return;  // diagnostics: prologue + staging only
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (KS: the instances of tap-range shares are kernels of their own -- as a run-time branch of the one kernel the
  //  second path cost the first its registers: 64 VGPRs and 28 bytes of scratch in the BASELINE configs[1] instance)
  if constexpr (KS) {  // every wave of the workgroup works on a group (fir_tile_parts)
    const __attribute__((address_space(4))) KernArgs *ka =
        (const __attribute__((address_space(4))) KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const KDesc dp = &ka->pack.d[blockIdx.y];
    if constexpr (AM == 1)
      fir_tile_parts64<R, CT, ONE_GROUP, PADDED, T, CGF, W16>(&ka->p, reinterpret_cast<const double *>(rows), dp, xs, wg.xshift,
                                                              m_lo, m_cnt, wave, threadIdx.x & 63u, blockIdx.z);
    else
      fir_tile_parts<R, CT, ONE_GROUP, PADDED, T, CGF, W16, AM == 2>(&ka->p, rows, dp, xs, wg.xshift, m_lo, m_cnt, wave,
                                                                    threadIdx.x & 63u, blockIdx.z);
    return;
  } else {
  if (wave >= p.wave_groups) return;  // staging helpers (see launch_period): no phase group of their own
  const __attribute__((address_space(4))) KernArgs *ka =
      (const __attribute__((address_space(4))) KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const KDesc dp = &ka->pack.d[blockIdx.y];
  fir_tile<R, CT, ONE_GROUP, PADDED, T, CGF, W16, AM>(p, d, &ka->p, rows, dp, xs, wg.xshift, m_lo, m_cnt, wave,
                                                  threadIdx.x & 63u, blockIdx.z, gridDim.z);
  }
}

template <int R, int CT, bool ONE_GROUP, bool PADDED, typename T, int CGF = 0, bool W16 = false, bool KS = false, int AM = 0>
hipError_t launch_rc(const PeriodParams &p, const DescPack *pack, dim3 grid, uint32_t threads, size_t lds_bytes,
                     hipStream_t stream) {
#ifdef SPEEXHIP_CXX_FIR_LOOP
  constexpr bool kParts = false;
#else
  constexpr bool kParts = KS && (AM != 0 || FirLoopAsm<R, CT, ONE_GROUP ? CT : CT * CGF, PADDED, W16>::available);
#endif
  if constexpr (KS && !kParts) {  // (no ISA loop for this layout: the host never asks for tap-range shares of it)
    return hipErrorInvalidValue;
  } else {
    static std::atomic<uint64_t> seen{0};
    opt_in_lds_on_this_device(resample_period<R, CT, ONE_GROUP, PADDED, T, CGF, W16, KS, AM>, seen);
    hipLaunchKernelGGL((resample_period<R, CT, ONE_GROUP, PADDED, T, CGF, W16, KS, AM>), grid, dim3(threads), lds_bytes, stream,
                       p, p.rows, *pack);
    return hipGetLastError();
  }
}

}  // namespace
}  // namespace speexhip
