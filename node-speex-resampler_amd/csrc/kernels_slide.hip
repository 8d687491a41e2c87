// kernels_slide.hip -- host side of the small-ratio fast kernel: shapes, plan, tap rows, launch geometry.
// The kernel itself is kernels_slide_impl.h, instantiated per sample type in kernels_slide_i16.hip / _f32.hip.
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

template <typename T>
hipError_t launch_slide_shape(const SlidePlan &t, const SlideParams &p, const DescPack *pack,
                              dim3 grid, uint32_t threads, size_t lds, hipStream_t stream);
extern template hipError_t launch_slide_shape<int16_t>(const SlidePlan &, const SlideParams &, const DescPack *, dim3, uint32_t, size_t, hipStream_t);
extern template hipError_t launch_slide_shape<float>(const SlidePlan &, const SlideParams &, const DescPack *, dim3, uint32_t, size_t, hipStream_t);

template <typename T>
hipError_t launch_slide64_shape(const SlidePlan &t, const SlideParams &p, const double *rows, const DescPack *pack, dim3 grid, uint32_t threads, size_t lds, hipStream_t stream);
extern template hipError_t launch_slide64_shape<int16_t>(const SlidePlan &, const SlideParams &, const double *, const DescPack *, dim3, uint32_t, size_t,
                                                         hipStream_t);
extern template hipError_t launch_slide64_shape<float>(const SlidePlan &, const SlideParams &, const double *, const DescPack *, dim3, uint32_t, size_t,
                                                       hipStream_t);

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
size_t slide_lds_bytes(const SlidePlan &t, uint32_t waves) {
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
  t.row_len = (f.taps + dmax + 2 * steps - 1) / (2 * steps) * (2 * steps);  // an even number of iterations
  const uint32_t row_elems = steps * channels;
  // row stride: distinct banks for the lanes of a wave (ds_read_b64: stride/2 odd; b32: stride odd)
  t.row_stride = row_elems;
  if (t.pair_ch) {
    if ((t.row_stride / 2) % 2 == 0) t.row_stride += 2;
  } else {
    if (t.row_stride % 2 == 0) t.row_stride += 1;
  }
  if (const char *e = SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_PAD")) t.row_stride = row_elems + static_cast<uint32_t>(std::atoi(e));  // diagnostics
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
                        const StreamDesc *h_descs, const DescPack *pack,
                        uint32_t n_streams, bool float_io, hipStream_t stream, bool fixed_shape) {
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
  static const uint32_t max_waves = SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_WAVES") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_WAVES")) : 8;
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
#ifdef SPEEXHIP_DIAG
  static const uint32_t skip_mask = static_cast<uint32_t>(diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_SKIP"), 0));
  p.skip = skip_mask;
#endif
  const uint32_t tile_periods = p.blocks_per_tile * t.p;
  const uint32_t tiles = (max_periods + tile_periods - 1) / tile_periods;
  // LDS: one row per lane block, + the rows the last lane's window runs into
  size_t lds = slide_lds_bytes(t, waves);
  dim3 grid((max_periods == 0 ? 0 : tiles) + 1, n_streams, 1);
  // Tap-range parts: a launch of a few workgroups of two waves runs every lane's whole filter as one chain --
  // P x den x row_len packed FMAs per wave, alone on its SIMD (48k -> 8k stereo, one stream: 16.7 us for 48 000
  // frames as for 441 000).  Such a launch gives each wave's lane blocks to `parts` waves, a range of the
  // iterations each; the sums meet in LDS behind a barrier (kernels_slide_impl.h).  SPEEXHIP_SLIDE_PARTS=0 turns
  // it off, =n forces n (A/B, tests).
  static const int env_parts = SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_PARTS") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_PARTS")) : -1;
  p.base_waves = waves;
  p.parts = 1;
  {
    const uint32_t pairs = t.row_len / (t.p * f.num) / 2;  // iteration pairs per lane
    const uint64_t chain = static_cast<uint64_t>(t.p) * t.np * t.row_len;  // packed FMAs per wave
    uint32_t parts = 16 / waves;
    if (env_parts > 0) parts = std::min<uint32_t>(parts, static_cast<uint32_t>(env_parts));
    while (parts > 1 && (pairs / parts < 2 || static_cast<size_t>(parts - 1) * waves * t.p * t.np * 64 * 8 > kSlideLdsLimit)) parts--;
    const bool small = static_cast<uint64_t>(tiles) * n_streams * waves <= 4ull * device_compute_units();  // at most one wave per SIMD
    if (!fixed_shape && env_parts != 0 && parts > 1 && (env_parts > 0 || (small && chain >= 1500))) {
      p.parts = parts;
      lds = std::max(lds, static_cast<size_t>(parts - 1) * waves * t.p * t.np * 64 * 8);
    }
  }
  const uint32_t threads = waves * p.parts * 64;
  p.threads = threads;
  return float_io ? launch_slide_shape<float>(t, p, pack, grid, threads, lds, stream)
                  : launch_slide_shape<int16_t>(t, p, pack, grid, threads, lds, stream);
}


// ---- the fp64-accumulate twin (kernels_slide64_impl.h) ---------------------------------------------------------
namespace {
struct Slide64Shape { uint32_t num, den, p; };
// (num, den) -> periods per lane.  Bounds: U*den <= 30 tap doubles per iteration (one bank of SGPR pairs; two banks
// up to 16), a ring of 2U doubles + P*den accumulators + U raw samples within ~110 VGPRs.
const Slide64Shape kShapes64[] = {
    {1, 1, 8}, {1, 2, 8}, {1, 3, 8}, {1, 4, 4}, {1, 5, 4}, {1, 6, 4}, {2, 1, 8}, {2, 3, 4}, {2, 5, 2}, {3, 1, 4},
    {3, 2, 4}, {3, 5, 2}, {4, 1, 4}, {4, 5, 1}, {5, 1, 4}, {5, 2, 2}, {5, 3, 2}, {5, 4, 1}, {5, 6, 1}, {6, 1, 2},
    {6, 5, 1}, {7, 1, 2}, {8, 1, 2}, {8, 3, 1}, {9, 1, 2}, {10, 1, 2}, {12, 1, 1}, {16, 1, 1}, {20, 1, 1}, {24, 1, 1},
    // 7:2 and 9:2 (56k -> 16k, 72k -> 16k): odd channel counts only -- the fp32 kernel's phase-pair shapes of 7:1 and
    // 9:1 cover den = 2 as well, and whatever the slide kernel runs has its fp64 twin
    {7, 2, 2}, {9, 2, 1},
};
}  // namespace

// the fp64 kernel keeps its LDS image in doubles where a sample read feeds >= 4 FMAs (kernels_slide64_impl.h, LD64)
static bool slide64_lds_doubles(const SlidePlan &t, uint32_t den) { return t.p * den >= 4; }

SlidePlan plan_slide64(const FilterSpec &f, uint32_t channels) {
  SlidePlan t;
  t.pair_ch = false;
  t.np = f.den;
  t.cgroups = channels;
  t.num = f.num;
  t.usable = false;
  for (const Slide64Shape &sh : kShapes64)
    if (sh.num == f.num && sh.den == f.den) {
      t.p = sh.p;
      t.usable = channels <= 64;
    }
  if (!t.usable) return t;
  const uint32_t steps = t.p * f.num;
  const uint32_t dmax = static_cast<uint32_t>((static_cast<uint64_t>(f.den - 1) * f.num) / f.den);
  t.row_len = (f.taps + dmax + 2 * steps - 1) / (2 * steps) * (2 * steps);  // an even number of iterations
  // row stride: the lanes of a wave read one element each, lane (block b, channel c) at b*stride + c: the pad with
  // the fewest lanes on one bank -- 64 banks of 4 bytes for the float image; an 8-byte element takes two, and a
  // wave's ds_read_b64 goes through in half-waves
  const uint32_t row_elems = steps * channels;
  const uint32_t blocks = 64 / channels;
  const bool ld64 = slide64_lds_doubles(t, f.den);
  uint32_t best = 0xffffffffu;
  t.row_stride = row_elems;
  for (uint32_t pad = 0; pad < 32; pad++) {
    uint32_t count[64] = {0}, worst = 0;
    for (uint32_t b = 0; b < blocks; b++)
      for (uint32_t c = 0; c < channels; c++) {
        const uint32_t lane = b * channels + c, at = b * (row_elems + pad) + c;
        if (ld64 && lane >= 32) continue;
        worst = std::max(worst, ++count[ld64 ? at % 32 : at % 64]);
      }
    if (worst < best) {
      best = worst;
      t.row_stride = row_elems + pad;
    }
  }
  if (slide_lds_bytes(t, 2) * (ld64 ? 2 : 1) > kSlideLdsLimit) t.usable = false;
  return t;
}

void build_slide64_rows(const FilterSpec &f, const SlidePlan &t, std::vector<double> *rows) {
  // [step][phase], rows of phase r shifted by delta_r, + one iteration of zero padding
  rows->assign(static_cast<size_t>(t.row_len + t.p * f.num) * f.den, 0.0);
  std::vector<double> h(f.taps);
  for (uint32_t r = 0; r < f.den; r++) {
    const uint32_t phase = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) % f.den);
    const uint32_t shift = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) / f.den);
    phase_taps(f, phase, h.data());
    for (uint32_t j = 0; j < f.taps; j++) (*rows)[static_cast<size_t>(j + shift) * f.den + r] = h[j];
  }
}

hipError_t launch_slide64(const FilterSpec &f, const SlidePlan &t, const double *d_rows, uint32_t channels,
                          const StreamDesc *h_descs, const DescPack *pack,
                          uint32_t n_streams, bool float_io, hipStream_t stream, bool fixed_shape) {
  uint32_t max_periods = 0;
  for (uint32_t s = 0; s < n_streams; s++) {
    if (h_descs[s].n_out == 0) continue;
    const uint64_t k_end = static_cast<uint64_t>(h_descs[s].k_shift) + h_descs[s].n_out;
    max_periods = std::max<uint32_t>(max_periods, static_cast<uint32_t>((k_end + f.den - 1) / f.den));
  }
  const uint32_t blocks_per_wave = 64 / t.cgroups;
  static const uint32_t max_waves = SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_WAVES") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_WAVES")) : 8;
  uint32_t waves = max_waves;
  while (waves > 2 && static_cast<uint64_t>(max_periods) * n_streams < 512ull * waves * blocks_per_wave * t.p)
    waves /= 2;
  const size_t eb = slide64_lds_doubles(t, f.den) ? 2 : 1;  // (image in doubles: twice the bytes)
  while (waves > 2 && slide_lds_bytes(t, waves) * eb > kSlideLdsLimit) waves /= 2;  // (fits with 2: plan_slide64)
  SlideParams p;
  p.rows = nullptr;  // (the fp64 rows travel as a kernel argument of their own)
  p.den = f.den;
  p.taps = f.taps;
  p.row_len = t.row_len;
  p.channels = channels;
  p.cgroups = t.cgroups;
  p.blocks_per_wave = blocks_per_wave;
  p.blocks_per_tile = blocks_per_wave * waves;
  p.row_stride = t.row_stride;
  p.row_magic = period_magic_of(t.p * f.num * channels);
#ifdef SPEEXHIP_DIAG
  static const uint32_t skip_mask = static_cast<uint32_t>(diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_SKIP"), 0));
  p.skip = skip_mask;
#endif
  const uint32_t tile_periods = p.blocks_per_tile * t.p;
  const uint32_t tiles = (max_periods + tile_periods - 1) / tile_periods;
  size_t lds = slide_lds_bytes(t, waves) * eb;
  dim3 grid((max_periods == 0 ? 0 : tiles) + 1, n_streams, 1);
  // tap-range parts, by the slide kernel's rule (a wave's chain here: P x den x row_len fp64 FMAs)
  static const int env_parts = SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_PARTS") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SLIDE_PARTS")) : -1;
  p.base_waves = waves;
  p.parts = 1;
  {
    const uint32_t pairs = t.row_len / (t.p * f.num) / 2;
    const uint64_t chain = static_cast<uint64_t>(t.p) * f.den * t.row_len;
    uint32_t parts = 16 / waves;
    if (env_parts > 0) parts = std::min<uint32_t>(parts, static_cast<uint32_t>(env_parts));
    while (parts > 1 && (pairs / parts < 2 || static_cast<size_t>(parts - 1) * waves * t.p * f.den * 64 * 8 > kSlideLdsLimit)) parts--;
    const bool small = static_cast<uint64_t>(tiles) * n_streams * waves <= 4ull * device_compute_units();
    if (!fixed_shape && env_parts != 0 && parts > 1 && (env_parts > 0 || (small && chain >= 1500))) {
      p.parts = parts;
      lds = std::max(lds, static_cast<size_t>(parts - 1) * waves * t.p * f.den * 64 * 8);
    }
  }
  const uint32_t threads = waves * p.parts * 64;
  p.threads = threads;
  return float_io ? launch_slide64_shape<float>(t, p, d_rows, pack, grid, threads, lds, stream)
                  : launch_slide64_shape<int16_t>(t, p, d_rows, pack, grid, threads, lds, stream);
}

}  // namespace speexhip
