// kernels_period.hip -- the primary fast gfx950 FIR kernel ("period-lane" mapping).
//
// Same algebra as kernels_tiled.hip: with K = k_shift + k = m*den + r every output is
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
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_helpers.h"
#include "device_types.h"
#include "filter_design.h"
#include "kernels.h"

namespace speexhip {
namespace {

// <= 80 SGPRs: the hardware admits 8 waves per SIMD (two 16-wave workgroups per CU) only then
// (MI355X_MICROARCH.md, residency); the compiler alone settles at ~106.
template <int R, int CT, int STEPS, bool PACKED>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) void resample_period(PeriodParams p, const float *__restrict__ rows,
                                                        const StreamDesc *streams, DescPack pack) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const StreamDesc d = PACKED ? pack.d[blockIdx.y] : streams[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {
    if (blockIdx.z == 0) roll_history(p.taps, p.channels, d);
    return;
  }
  if (d.n_out == 0) return;
  const uint32_t K_end = d.k_shift + d.n_out;            // exclusive canonical output index
  const uint32_t m_total = (K_end + p.den - 1) / p.den;  // periods touched by this call
  const uint32_t m_lo = blockIdx.x * p.lane_periods;
  if (m_lo >= m_total) return;
  const uint32_t m_cnt = min(p.lane_periods, m_total - m_lo);
  const uint32_t C = p.channels;

  // ---- stage the input window: interleaved s16 in HBM -> float in LDS ----------------------
  const int64_t hist_elems = static_cast<int64_t>(p.taps - 1) * C;
  const int64_t in_elems = static_cast<int64_t>(d.in_frames) * C;
  const int64_t q_lo =
      (static_cast<int64_t>(d.base_shift) + static_cast<int64_t>(m_lo) * p.num) * C - hist_elems;
  const int64_t q_base = (q_lo >= 0 ? q_lo / 8 : -((-q_lo + 7) / 8)) * 8;
  const uint32_t xshift = static_cast<uint32_t>(q_lo - q_base);
  const uint32_t span = (m_cnt - 1) * p.num + p.tail_frames;
  if (!(p.skip & 2u)) stage_window<4>(xs, d, q_base, (xshift + span * C + 7) / 8, hist_elems, in_elems);
  __syncthreads();

  // ---- wave / lane coordinates ---------------------------------------------------------------
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t cg = lane % p.cgroups;
  const uint32_t pl = lane / p.cgroups;  // period of this lane inside the tile
  const bool lane_live = pl < m_cnt;
  // float index of this lane's first sample of a group with delta_g = 0
  const uint32_t xlane = xshift + min(pl, p.lane_periods - 1) * p.num * C + cg * CT;
  const uint64_t K_lane = static_cast<uint64_t>(m_lo + pl) * p.den;

  const uint32_t g_first = blockIdx.z * p.wave_groups + wave;
  const uint32_t g_step = p.wave_groups * gridDim.z;
  // One round (every wave owns at most one group): outputs are transposed through LDS, which
  // may then reuse the window.  Several rounds (more groups than waves): direct stores.
  const bool via_lds = CT == 2 && p.groups <= g_step && !(p.skip & 16u);
  const uint32_t r_lo = blockIdx.z * p.wave_groups * R;                    // first phase of this workgroup
  const uint32_t r_hi = min(p.den, (blockIdx.z + 1) * p.wave_groups * R);  // one past its last
  const uint32_t run = r_hi > r_lo ? r_hi - r_lo : 0;                      // phases per period here
  for (uint32_t g = g_first; g < p.groups || via_lds; g += g_step) {
    const bool has_group = g < p.groups;
    float acc[R][CT];
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
      for (int ct = 0; ct < CT; ct++) acc[i][ct] = 0.f;
    if (has_group) {
      const uint32_t delta_g = static_cast<uint32_t>((static_cast<uint64_t>(g) * R * p.num) / p.den);
      // `rows` is a __restrict__ kernel argument: provably invariant, so these wave-uniform
      // loads become s_load_dwordx16 and the taps stay in SGPRs (2 steps = 2R taps at a time
      // keeps the kernel under 96 SGPRs: 8 waves per SIMD, two workgroups per CU)
      const float *__restrict__ trow = rows + static_cast<size_t>(g) * p.l4 * (4 * R);
      const float *xp = xs + xlane + delta_g * C;
      const uint32_t n_it = (p.skip & 4u) ? 0 : p.l4 * (4 / STEPS);
      const uint32_t tap_mask = (p.skip & 32u) ? 0u : ~0u;  // diagnostics: 32 = re-read tap block 0
      for (uint32_t it = 0; it < n_it; it++) {
        float tap[STEPS * R];
#pragma unroll
        for (int k = 0; k < STEPS * R; k++) tap[k] = trow[(it & tap_mask) * (STEPS * R) + k];
#pragma unroll
        for (int u = 0; u < STEPS; u++) {
          float x[CT];
          const float *px = xp + (it * STEPS + u) * C;
          if (CT == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(px);
            x[0] = v.x;
            x[CT - 1] = v.y;
          } else {
            x[0] = *px;
          }
#pragma unroll
          for (int i = 0; i < R; i++)
#pragma unroll
            for (int ct = 0; ct < CT; ct++) acc[i][ct] = fmaf(tap[u * R + i], x[ct], acc[i][ct]);
        }
      }
    }
    if (p.skip & 8u) {
      if (via_lds) break;
      continue;
    }

    if (via_lds) {
      // ---- transpose through LDS: image[period][phase - r_lo][channel pair] of packed s16x2,
      //      then each period's run goes out as consecutive dwords (full lines per wave).
      __syncthreads();  // every wave is done reading the window
      uint32_t *img = reinterpret_cast<uint32_t *>(xs);
      if (has_group && pl < p.lane_periods) {
#pragma unroll
        for (int i = 0; i < R; i++) {
          const uint32_t r = g * R + i;
          if (r < r_hi) img[(pl * run + (r - r_lo)) * p.cgroups + cg] = round_pack_pcm(acc[i][0], acc[i][CT - 1]);
        }
      }
      __syncthreads();
      // wave w copies periods w, w+W, ...; lanes sweep a period's run 64 dwords at a time
      const uint32_t row_dw = run * p.cgroups;  // dwords per period in the image
      const uint32_t n_waves = blockDim.x >> 6;
      for (uint32_t pp = wave; pp < m_cnt; pp += n_waves) {
        const uint64_t K_row = static_cast<uint64_t>(m_lo + pp) * p.den + r_lo;
        for (uint32_t w = lane; w < row_dw; w += 64) {
          const uint32_t fr = p.cgroups == 1 ? w : w / p.cgroups;
          const uint32_t cgi = p.cgroups == 1 ? 0 : w - fr * p.cgroups;
          const uint64_t K = K_row + fr;
          if (K < d.k_shift || K >= K_end) continue;
          int16_t *o = d.out + (K - d.k_shift) * C + cgi * 2;
          const uint32_t v = img[pp * row_dw + w];
          if ((reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            *reinterpret_cast<uint32_t *>(o) = v;
          } else {
            o[0] = static_cast<int16_t>(v & 0xffffu);
            o[1] = static_cast<int16_t>(v >> 16);
          }
        }
      }
      break;
    }

    // ---- direct stores (several rounds, or an odd channel count) -----------------------------
    if (lane_live) {
#pragma unroll
      for (int i = 0; i < R; i++) {
        const uint32_t r = g * R + i;
        const uint64_t K = K_lane + r;
        if (r >= p.den || K < d.k_shift || K >= K_end) continue;
        int16_t *o = d.out + (K - d.k_shift) * C + cg * CT;
        if (CT == 2) {
          const uint32_t v = round_pack_pcm(acc[i][0], acc[i][CT - 1]);
          if ((reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            *reinterpret_cast<uint32_t *>(o) = v;
          } else {
            o[0] = static_cast<int16_t>(v & 0xffffu);
            o[1] = static_cast<int16_t>(v >> 16);
          }
        } else {
          o[0] = static_cast<int16_t>(round_pack_pcm(acc[i][0], 0.f) & 0xffffu);
        }
      }
    }
  }
}

template <int R, int CT, int STEPS>
hipError_t launch_rc(const PeriodParams &p, const StreamDesc *d_descs, const DescPack *pack, dim3 grid,
                     uint32_t threads, size_t lds_bytes, hipStream_t stream) {
  if (pack != nullptr) {
    auto kern = resample_period<R, CT, STEPS, true>;
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      lds_opt_in = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, stream, p, p.rows, nullptr, *pack);
  } else {
    DescPack empty;
    std::memset(&empty, 0, sizeof(empty));
    auto kern = resample_period<R, CT, STEPS, false>;
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      lds_opt_in = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, stream, p, p.rows, d_descs, empty);
  }
  return hipGetLastError();
}

const size_t kSlack = 16;  // floats: window starts on the input's 16-byte grid, staged by 8
const uint32_t kR = 10;

}  // namespace

PeriodPlan plan_period(const FilterSpec &f, uint32_t channels, size_t lds_budget) {
  PeriodPlan t;
  t.r = kR;
  t.ct = (channels % 2 == 0) ? 2 : 1;
  t.cgroups = channels / t.ct;
  t.groups = (f.den + t.r - 1) / t.r;
  uint32_t dmax = 0;  // largest shift of a row inside its group
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint64_t d0 = (static_cast<uint64_t>(g) * t.r * f.num) / f.den;
    const uint32_t r_last = std::min<uint32_t>(g * t.r + t.r - 1, f.den - 1);
    dmax = std::max<uint32_t>(dmax, static_cast<uint32_t>((static_cast<uint64_t>(r_last) * f.num) / f.den - d0));
  }
  t.row_len = (f.taps + dmax + 3) / 4 * 4;
  t.l4 = t.row_len / 4;
  t.tail_frames = static_cast<uint32_t>((static_cast<uint64_t>(t.groups - 1) * t.r * f.num) / f.den) + t.row_len;
  t.lane_periods = 64 / t.cgroups;
  t.rows_floats = static_cast<size_t>(t.groups) * t.l4 * 4 * t.r;
  t.window_bytes = ((static_cast<size_t>(t.lane_periods) - 1) * f.num + t.tail_frames) * channels * 4 + kSlack * 4;
  // the same LDS later holds the tile's output image (one packed s16 pair per dword)
  t.window_bytes = std::max(t.window_bytes, static_cast<size_t>(t.lane_periods) * f.den * t.cgroups * 4);
  // needs enough phases to fill the R-wide register tile and a window that fits one CU's LDS
  t.usable = f.den >= 7 && t.cgroups <= 64 && t.lane_periods >= 1 && t.window_bytes <= lds_budget;
  return t;
}

void build_period_rows(const FilterSpec &f, const PeriodPlan &t, std::vector<float> *rows) {
  rows->assign(t.rows_floats, 0.f);
  std::vector<double> h(f.taps);
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint64_t d0 = (static_cast<uint64_t>(g) * t.r * f.num) / f.den;
    for (uint32_t i = 0; i < t.r; i++) {
      const uint32_t r = g * t.r + i;
      if (r >= f.den) continue;  // padding phases of the last group stay zero
      const uint32_t phase = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) % f.den);
      const uint32_t shift = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) / f.den - d0);
      phase_taps(f, phase, h.data());
      for (uint32_t j = 0; j < f.taps; j++) {
        const uint32_t s = j + shift;
        (*rows)[((static_cast<size_t>(g) * t.l4 + s / 4) * 4 + (s & 3)) * t.r + i] = static_cast<float>(h[j]);
      }
    }
  }
}

hipError_t launch_period(const FilterSpec &f, const PeriodPlan &t, const float *d_rows, uint32_t channels,
                         const StreamDesc *h_descs, const StreamDesc *d_descs, const DescPack *pack,
                         uint32_t n_streams, hipStream_t stream) {
  uint32_t max_periods = 0;
  for (uint32_t s = 0; s < n_streams; s++) {
    if (h_descs[s].n_out == 0) continue;
    const uint64_t k_end = static_cast<uint64_t>(h_descs[s].k_shift) + h_descs[s].n_out;
    max_periods = std::max<uint32_t>(max_periods, static_cast<uint32_t>((k_end + f.den - 1) / f.den));
  }
  const uint32_t tiles = (max_periods + t.lane_periods - 1) / t.lane_periods;
  // Few tiles (one short stream): split each tile's phase groups over several workgroups so
  // the launch still covers the chip; many tiles: one workgroup of up to 16 waves per tile.
  uint32_t splits = 1;
  while (splits * 2 <= t.groups && static_cast<uint64_t>(tiles) * n_streams * splits * 2 <= 512 &&
         (t.groups + splits * 2 - 1) / (splits * 2) >= 2)
    splits *= 2;
  static const uint32_t max_waves = std::getenv("SPEEXHIP_WAVES") ? std::atoi(std::getenv("SPEEXHIP_WAVES")) : 16;
  const uint32_t wave_groups = std::min<uint32_t>((t.groups + splits - 1) / splits, max_waves);
  PeriodParams p;
  p.rows = d_rows;
  p.l4 = t.l4;
  p.groups = t.groups;
  p.num = f.num;
  p.den = f.den;
  p.taps = f.taps;
  p.channels = channels;
  p.cgroups = t.cgroups;
  p.lane_periods = t.lane_periods;
  p.wave_groups = wave_groups;
  p.tail_frames = t.tail_frames;
  static const uint32_t skip_mask = std::getenv("SPEEXHIP_SKIP") ? std::atoi(std::getenv("SPEEXHIP_SKIP")) : 0;
  p.skip = skip_mask;
  dim3 grid((max_periods == 0 ? 0 : tiles) + 1, n_streams, splits);
  const uint32_t threads = wave_groups * 64;
  static const int steps = std::getenv("SPEEXHIP_TAPSTEPS") ? std::atoi(std::getenv("SPEEXHIP_TAPSTEPS")) : 4;
  if (t.ct == 2)
    return steps == 2 ? launch_rc<kR, 2, 2>(p, d_descs, pack, grid, threads, t.window_bytes, stream)
                      : launch_rc<kR, 2, 4>(p, d_descs, pack, grid, threads, t.window_bytes, stream);
  return launch_rc<kR, 1, 4>(p, d_descs, pack, grid, threads, t.window_bytes, stream);
}

}  // namespace speexhip
