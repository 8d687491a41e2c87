// kernels_period.hip -- host side of the primary fast gfx950 FIR kernel ("period-lane" mapping): plans, tap rows,
// launch shapes and the table of fp32 instances.  The kernel itself: kernels_period_impl.h.
#include "kernels_period_impl.h"

namespace speexhip {
// the fp64-accumulate instances (kernels_period64.hip): launch the one of plan `t`'s layout
hipError_t dispatch_period64(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                             dim3 grid, uint32_t threads, bool float_io, hipStream_t stream);
// ... over an int16 LDS window (kernels_period64_w16.hip)
hipError_t dispatch_period64_w16(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                                 dim3 grid, uint32_t threads, bool float_io, hipStream_t stream);
// the instances for frames of five and seven channels (kernels_period_odd.hip)
hipError_t dispatch_period_frames(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                                  dim3 grid, uint32_t threads, bool float_io, hipStream_t stream);  // kernels_period_frames.hip
hipError_t dispatch_period_w16g(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack, dim3 grid, uint32_t threads,
                                hipStream_t stream);  // kernels_period_w16g.hip
hipError_t dispatch_period_odd(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                               dim3 grid, uint32_t threads, bool float_io, hipStream_t stream);
// the phase-pair instances for mono (kernels_period_pp.hip)
hipError_t dispatch_period_pp(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                              dim3 grid, uint32_t threads, bool float_io, hipStream_t stream);
namespace {

const size_t kSlack = 16;  // floats: window starts on the input's 16-byte grid, staged by 8
// Phases per wave: R = 10 (banks of 20 taps = 2 steps) or R = 5 (banks of 30 taps = 6 steps, see
// fir_group; with 20-tap banks as well the one-stream launch measured 12.08 vs 11.95 us on the same
// box).  R = 5 pads its rows by 4 instead of 9 steps and
// gives a tile twice as many wave-sized pieces -- what a launch of a single generation of workgroups
// needs -- at half the FMAs per sample read and 20-byte instead of 40-byte store pieces per lane.
// A ratio with few phases (den <= 80: 44.1k -> 8k = 441:80, 88.2k -> 48k = 147:80, 88.2k -> 16k ...) has at
// most 8 groups of 10: 8 FIR waves per workgroup, 2-4 per SIMD.  Groups of 5 double the waves: 32 streams
// of 44.1k -> 8k stereo q7 193 -> 88 us, 88.2k -> 48k 69 -> 63, 4 channels 44.1k -> 8k 156 -> 127 (same box).
uint32_t default_r(const FilterSpec &f) {
  static const uint32_t forced = [] {
    const char *e = SPEEXHIP_DIAG_ENV("SPEEXHIP_R");
    return e != nullptr ? static_cast<uint32_t>(std::atoi(e)) : 0u;
  }();
  if (forced == 5 || forced == 10) return forced;
  return (f.den + 9) / 10 <= 8 ? 5u : 10u;
}
// ... of a phase-pair plan: a group is 2r phases, so the same reasoning puts the line at den <= 160
uint32_t default_r_pp(const FilterSpec &f) {
  const char *e = SPEEXHIP_DIAG_ENV("SPEEXHIP_R");
  const uint32_t forced = e != nullptr ? static_cast<uint32_t>(std::atoi(e)) : 0u;
  if (forced == 5 || forced == 10) return forced;
  return (f.den + 19) / 20 <= 8 ? 5u : 10u;
}

}  // namespace

bool period_view(const FilterSpec &f, uint32_t channels, FilterSpec *view) {
  if (f.fold != 1 || f.den > 6 || f.den == 0 || plan_slide(f, channels).usable) return false;
  uint32_t k = 5;
  while (k * f.den < 10) k += 5;
  if (static_cast<uint64_t>(f.num) * k > 0x7fffffffull) return false;
  *view = f;
  view->num = f.num * k;
  view->den = f.den * k;
  view->fold = k;
  return true;
}

namespace {
// Launch time: the filter as a plan made on a folded view sees it (the state's FilterSpec carries no table then: a cheap
// copy), and the descriptors with their period counts on the folded den.  k_shift and base_shift need no change: output K
// of a stream is phase (K num) mod den at window start (K num) div den whatever period length K is decomposed by.
struct FoldedLaunch {
  FilterSpec f;
  DescPack pack;
  FoldedLaunch(const FilterSpec &real, uint32_t fold, const StreamDesc *h_descs, const DescPack *src, uint32_t n_streams) : f(real) {
    f.num *= fold;
    f.den *= fold;
    f.fold = fold;
    if (src != nullptr)
      pack = *src;
    else
      std::memcpy(pack.d, h_descs, sizeof(StreamDesc) * n_streams);
    for (uint32_t s = 0; s < n_streams; s++)
      pack.d[s].m_total = static_cast<uint32_t>((static_cast<uint64_t>(pack.d[s].k_shift) + pack.d[s].n_out + f.den - 1) / f.den);
  }
};
}  // namespace

PeriodPlan plan_period(const FilterSpec &f, uint32_t channels, size_t lds_budget, bool w16, bool a64, bool pp) {
  PeriodPlan t = plan_period_r(f, channels, lds_budget, pp ? default_r_pp(f) : default_r(f), w16, a64, pp);
  if (!t.usable && t.r != 10) t = plan_period_r(f, channels, lds_budget, 10, w16, a64, pp);
  return t;
}

PeriodPlan plan_period_w16(const FilterSpec &f, uint32_t channels, size_t lds_budget, const PeriodPlan &t) {
  PeriodPlan w;
  static const bool off = SPEEXHIP_DIAG_ENV("SPEEXHIP_NO_W16") != nullptr;  // diagnostics: A/B
  if (!t.usable || off) return w;
  w = plan_period_r(f, channels, lds_budget, t.r, true, t.a64, t.pp);
  static const bool force = SPEEXHIP_DIAG_ENV("SPEEXHIP_FORCE_W16") != nullptr;  // diagnostics: every layout that has one
  // What the int16 window costs is two conversions per sample read: +20 % vector instructions where a read
  // feeds 10 packed FMAs (R = 10, channel pairs), +40 % at R = 5, and single-channel lanes convert two
  // 2-byte reads per step.  Measured at 32 streams x 2^20 frames (profiles/r03_w16_ab.txt): stereo 48k->11.025k
  // 28 -> 58 periods per tile 530 -> 338 us, 4 channels 14 -> 28: 1032 -> 687, stereo 44.1k->16k 42 -> 64: 363 -> 277,
  // mono 48k->11.025k 58 -> 116: 289 -> 197; but stereo 44.1k->8k (R = 5) 41 -> 64 only 304 -> 272 and mono
  // 44.1k->16k 86 -> 128 nothing (196 -> 198).  So: 5/4 of the periods for R = 10 on channel pairs, 7/4 otherwise.
  // (fp64 loops, round 5: single-channel lanes convert for free -- v_cvt_f64_i32 where v_cvt_f64_f32 stood --, channel
  //  pairs pay two instructions per 2 R v_fma_f64)
  const bool cheap = t.a64 || (t.r == 10 && t.ct == 2);
  // (phase pairs: a tile is 64 periods either way; what the int16 window buys there is a second and third workgroup
  //  per CU -- taken when the float window leaves room for one only)
  const bool pp_fits_more = t.pp && w.usable && t.window_bytes > 80 * 1024 && w.window_bytes <= 80 * 1024;
  if (w.usable && !force && !pp_fits_more && t.float_ok && 4 * w.lane_periods < (cheap ? 5 : 7) * t.lane_periods) w.usable = false;
  return w;
}

// w16: plan for an int16 LDS window (2-byte elements; pad, half_offset ... count elements either way)
PeriodPlan plan_period_r(const FilterSpec &f, uint32_t channels, size_t lds_budget, uint32_t r, bool w16, bool a64, bool pp) {
  PeriodPlan t;
  t.r = r;
  t.w16 = w16;
  t.a64 = a64;
  t.fold = f.fold;
  const size_t eb = w16 ? 2 : 4;  // bytes per LDS element
  // Phase pairs for mono (round 4): the two halves of a packed FMA are two phases of one sample instead of two
  // periods of one tap, so a tile is 64 periods instead of 128 -- half the window for the same lanes (wide windows:
  // mono 48k -> 11.025k has 640 frames per period) -- at twice the tap bytes per FMA and half the FMAs between two
  // waits of the loop.  Planned beside the two-period plans for wide windows; chosen per launch
  // (period_launch_prefers_pp).
  t.pp = pp && !a64 && channels <= 3;  // (lane = (period, channel): mono, stereo, three channels)
  // steps per loop iteration (two banks); the fp64 loop's banks hold half as many taps (doubles in the same SGPRs),
  // and so do the phase-pair loop's (two phases per packed FMA)
  const uint32_t it_steps = 2 * bank_taps(t.r) / t.r / ((a64 || t.pp) ? 2 : 1);
  const uint32_t gw = t.pp ? 2 * t.r : t.r;  // phases per group
  t.ct = (channels % 2 == 0 && !t.pp) ? 2 : 1;
  t.cgroups = channels / t.ct;
  t.groups = (f.den + gw - 1) / gw;
  uint32_t dmax = 0;  // largest shift of a row inside its group
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint64_t d0 = (static_cast<uint64_t>(g) * gw * f.num) / f.den;
    const uint32_t r_last = std::min<uint32_t>(g * gw + gw - 1, f.den - 1);
    dmax = std::max<uint32_t>(dmax, static_cast<uint32_t>((static_cast<uint64_t>(r_last) * f.num) / f.den - d0));
  }
  t.row_len = (f.taps + dmax + it_steps - 1) / it_steps * it_steps;
  t.l4 = t.row_len / it_steps;
  t.tail_frames = static_cast<uint32_t>((static_cast<uint64_t>(t.groups - 1) * gw * f.num) / f.den) + t.row_len;
  t.lane_periods = 64 / t.cgroups;  // revised below once the window size is known
  // + one iteration (40 floats) of zero padding: the tap pipeline prefetches one past the end
  t.rows_floats = static_cast<size_t>(t.groups) * t.row_len * gw + it_steps * gw;
  // Bank padding: the lanes of a wave read the window num*channels floats apart.  Pick the pad
  // (multiple of 4 floats, inserted after every period) with the fewest lanes of a half-wave on
  // the same bank (ds_read_b32) / bank pair (ds_read_b64); 1 = conflict-free.  (Measured on
  // stereo 48k->44.1k, 32 streams: no pad 622 us, pad 4 -- two lanes per bank pair -- 188 us, pad 2
  // -- conflict-free, but 8-byte staging writes -- 199 us, pad 8: 288, pad 16: 411.)
  // (padded float windows of four channel pairs on the fp32 chain run the ROW mapping of lane_ctx: channel pair =
  //  lane / 16, period = lane % 16 -- kernels_period_impl.h; the unpadded instance keeps lane % 4)
  const bool rows = t.ct == 2 && t.cgroups == 4 && !w16 && !a64 && !t.pp && t.r == 10;
  auto worst_bank_load = [&](uint32_t pad) {
    // float image: a lane reads `ct` floats (ds_read_b32 / _b64); int16 image: the dword its one or two
    // samples sit in (ds_read_i16 / _b32, 32 banks)
    const uint32_t unit = w16 ? 2u : t.ct;
    const uint32_t slots = w16 ? 32u : 64 / unit;
    const uint32_t stride = f.num * channels + pad;
    uint32_t count[64] = {0}, worst = 0;
    for (uint32_t lane = 0; lane < 32; lane++) {
      const uint32_t cg = (rows && pad != 0) ? lane >> 4 : lane % t.cgroups, pl = (rows && pad != 0) ? (lane & 15u) : lane / t.cgroups;
      const uint32_t slot = ((pl * stride + cg * t.ct) / unit) % slots;
      worst = std::max(worst, ++count[slot]);
    }
    return worst;
  };
  t.pad = 0;
  {
    uint32_t best = worst_bank_load(0);
    const uint32_t pad_step = w16 ? 8 : 4;  // 16 bytes: the staging writes stay 16-byte aligned
    for (uint32_t pad = pad_step; pad <= 16 * pad_step && best > 1; pad += pad_step) {
      const uint32_t w = worst_bank_load(pad);
      if (w < best) {
        best = w;
        t.pad = pad;
      }
    }
  }
  static const bool no_pad = SPEEXHIP_DIAG_ENV("SPEEXHIP_NO_PAD") != nullptr;
  if (no_pad) t.pad = 0;
  if (SPEEXHIP_DIAG_ENV("SPEEXHIP_PAD")) t.pad = static_cast<uint32_t>(std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_PAD"))) & ((w16 || t.ct == 2) ? ~1u : ~0u);  // diagnostics
  if (t.pad != 0 && t.r != 10) return t;  // (the padded walk below is written for 4-step iterations)
  if (t.pad != 0) {
    // A padded window is only walked cheaply if every period boundary a group's window crosses
    // falls between two iterations: the first one is moved there by starting the group k <= 3
    // frames early (k extra zero taps in front of its rows); further ones follow num/4
    // iterations apart, which needs num to be a multiple of 4.
    const uint32_t row_len = (f.taps + dmax + 3 + 3) / 4 * 4;
    bool ok = true;
    for (uint32_t g = 0; g < t.groups && ok; g++) {
      const uint32_t dg = static_cast<uint32_t>((static_cast<uint64_t>(g) * gw * f.num) / f.den);
      if (dg + row_len + 4 <= f.num) continue;                 // never reaches the boundary
      const uint32_t k = (dg % 4 + 4 - f.num % 4) % 4;  // delta' = dg - k == num (mod 4)
      ok = dg >= k && ((dg - k) + row_len + 4 <= 2 * f.num || f.num % 4 == 0);
    }
    if (ok) {
      t.row_len = row_len;
      t.l4 = row_len / it_steps;  // (it_steps: 4, or 2 with an fp64 accumulator; the boundary tables count 4-step units)
      t.rows_floats = static_cast<size_t>(t.groups) * t.row_len * gw + it_steps * gw;
      t.tail_frames = static_cast<uint32_t>((static_cast<uint64_t>(t.groups - 1) * gw * f.num) / f.den) + t.row_len;
    } else {
      t.pad = 0;
    }
  }
  // (+ 4 frames: the tap/sample pipeline prefetches one bank past the last step; + the padding
  //  of every period the window can touch)
  auto window_bytes_for = [&](uint32_t lane_periods) {
    // (single-channel lanes read their second period at pl + ceil(periods/2): an odd tile still
    //  reads -- and discards -- one period past its last, so size the image for an even count)
    if (t.ct == 1 && !t.pp) lane_periods = (lane_periods + 1) / 2 * 2;
    const size_t pad_floats = static_cast<size_t>(t.pad) * (lane_periods + t.tail_frames / f.num + 2);
    size_t bytes = (((static_cast<size_t>(lane_periods) - 1) * f.num + t.tail_frames + it_steps) * channels + pad_floats) * eb +
                   kSlack * eb;
    return (bytes + 15) / 16 * 16;
  };
  // Periods per tile: all the lanes of a wave if that leaves room for TWO workgroups per CU (one
  // workgroup's staging and stores only overlap FMAs if another one is resident; same box,
  // SPEEXHIP_FULL_TILE=1 against the default: 8-channel 48k->44.1k 785 -> 643 us with 15 of 16
  // periods, stereo 48k->44.1k 225 -> 190 us with 62 of 64).  Otherwise weigh a second workgroup
  // (~20 %) against the lanes it costs.
  // (single-channel lanes carry two periods each, see lane_ctx)
  const uint32_t full = 64 / t.cgroups * ((t.ct == 1 && !t.pp) ? 2 : 1);
  const size_t half_lds = 80 * 1024;
  uint32_t fit_half = 0, fit_all = 0;
  for (uint32_t lp = full; lp >= 1; lp--) {
    const size_t bytes = window_bytes_for(lp);
    if (fit_all == 0 && bytes <= lds_budget) fit_all = lp;
    if (bytes <= half_lds) {
      fit_half = lp;
      break;
    }
  }
  t.lane_periods = (fit_half != 0 && 5 * fit_half >= 4 * fit_all) ? fit_half : fit_all;
  static const bool full_tile = SPEEXHIP_DIAG_ENV("SPEEXHIP_FULL_TILE") != nullptr;  // diagnostics: A/B of the rule above
  if (full_tile) t.lane_periods = fit_all;
  t.window_bytes = t.lane_periods ? window_bytes_for(t.lane_periods) : 0;
  // needs enough phases to fill the R-wide register tile and a window that fits one CU's LDS
  // ... and at least a ninth of each wave at work.  (A quarter until round 5, then an eighth: 96k -> 11.025k and 32k -> 11.025k,
  // num = 1280, fit 28 of a mono tile's 128 periods and 14 of a stereo tile's 64 -- just under it -- and ran the exact
  // kernel: 32 streams x 131 072 frames mono 223 us there against 52 us here, stereo 265 / 81, 32k -> 11.025k 212 / 83 and
  // 245 / 119; one stream 37.6 / 21.8, 51.9 / 21.1, 16.4 / 13.1, 20.3 / 23.3: profiles/r05_wide_windows.txt.)
  // (SPEEXHIP_MIN_FILL=n, diagnostics: at least 1/n of the lanes instead of a quarter -- profiles/r05_wide_windows.txt)
  // (a ninth, late in round 5: seven channels at num = 1280 fit 2 of a tile's 18 periods and still ran the exact kernel --
  //  32 x 131 072 frames of 96k / 32k -> 11.025k 1430 / 1400 us there, 387 / 269 here, one stream 361 -> 133 / 98:
  //  profiles/r05_minfill9.txt)
  static const uint32_t min_fill = SPEEXHIP_DIAG_ENV("SPEEXHIP_MIN_FILL") ? std::max(1, std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_MIN_FILL"))) : 9;
  // (the fp64 plans keep the quarter: at quality 10 the same two ratios, 32 streams, took 459 / 898 / 429 / 886 us on the
  //  exact kernel -- bit-exact there -- against 643 / 977 / 473 / 998 here; profiles/r05_wide_windows.txt)
  const uint32_t fill_rule = a64 && !SPEEXHIP_DIAG_ENV("SPEEXHIP_MIN_FILL") ? 4u : min_fill;
  bool filled = fill_rule * t.lane_periods >= full;
  // (... or its int16-window plan does: 8 channels at num = 1280 with 2 232 taps -- 96k -> 11.025k, quality 8-10 -- fit ONE
  //  period of the float window, a sixteenth of a tile, and three of the int16 one.  The float plan then exists for the
  //  int16 plan to hang off -- int16 calls run over that -- and serves the float calls of such a state itself.  Late in
  //  round 5: 32 x 131 072 frames 3 241 us on the exact kernel, profiles/r05_q10_sweep.txt)
  if (!filled && !w16 && !a64 && !t.pp && t.lane_periods >= 1 && t.window_bytes <= lds_budget) {
    const PeriodPlan i16 = plan_period_r(f, channels, lds_budget, r, true, false, false);
    filled = i16.usable;
  }
  t.usable = f.den >= 7 && t.cgroups <= 64 && filled && t.window_bytes <= lds_budget;
  // (... or not even ONE period of the float window fits -- 16 channels at num = 1280 with 1 120 taps: 155 KB -- and the int16
  //  window holds a ninth of a tile: a plan that exists only for its int16 plan, float_ok = false)
  if (!t.usable && fit_all == 0 && !w16 && !a64 && !t.pp && f.den >= 7 && t.cgroups <= 64) {
    const PeriodPlan i16 = plan_period_r(f, channels, lds_budget, r, true, false, false);
    if (i16.usable) {
      t.usable = true;
      t.float_ok = false;
      t.lane_periods = 1;
      t.window_bytes = 0;
    }
  }
  // an int16 window is read by the ISA loop only: mono, stereo, 4 / 6 / 8 channels (csrc/gen_fir_loop.py); so are the
  // tap rows of an fp64 accumulator
  // (round 5: frames of 5 and 7 single channels have ISA loops too -- int16 window (kernels_period_odd.hip)
  //  and fp64 rows (kernels_period64.hip); three channels likewise)
  // (three channels: the int16 window of the two-period plan too, late in round 5 -- until then only their phase-pair
  //  plans had one: one stream of 2^20 frames 48k / 32k / 96k -> 11.025k 40.9 / 76.5 / 67.1 -> 25.1 / 40.4 / 35.2 us,
  //  32 such streams 1555 / 1496 -> 822 / 794; +6.9 % in the geometric mean of 40 launches, five of them 5-13 % slower:
  //  profiles/r05_w16_3ch.txt.  SPEEXHIP_W16_3CH=0: as before, A/B)
  static const bool w16_3ch = !(SPEEXHIP_DIAG_ENV("SPEEXHIP_W16_3CH") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_W16_3CH")) == 0);
  const bool odd_frame = t.ct == 1 && (t.cgroups == 5 || t.cgroups == 7 || (t.cgroups == 3 && (a64 || w16_3ch)));
  // (frames of 10 / 12 / 16 channels -- 5, 6, 8 channel pairs -- have ISA loops of the fp32 chain since late in round 5:
  //  the int16 window, not the fp64 rows)
  const bool wide_frame = t.ct == 2 && (t.cgroups == 5 || t.cgroups == 6 || t.cgroups == 8) && !a64;
  // (round 6: the int16 window on every other layout too, read by the C++ loop -- kernels_period_w16g.hip; the fp64 rows
  //  stay with the ISA loops)
  if (a64 && !t.pp && !((t.ct == 2 && t.cgroups <= 4) || (t.ct == 1 && t.cgroups == 1) || odd_frame || wide_frame)) t.usable = false;
  if (w16 && !t.pp && t.pad != 0 && t.r != 10) t.usable = false;
  if (t.pp && t.pad != 0 && t.r != 10) t.usable = false;
  return t;
}

// Start-of-group adjustment for a padded window: k frames early so that (num - delta') % 4 == 0
static uint32_t group_shift(const FilterSpec &f, const PeriodPlan &t, uint32_t g) {
  if (t.pad == 0) return 0;
  const uint32_t dg = static_cast<uint32_t>((static_cast<uint64_t>(g) * (t.pp ? 2 * t.r : t.r) * f.num) / f.den);
  if (dg + t.row_len + 4 <= f.num) return 0;  // this group never reaches the period boundary
  return (dg % 4 + 4 - f.num % 4) % 4;
}

namespace {
template <typename E>
void build_period_rows_t(const FilterSpec &f, const PeriodPlan &t, std::vector<E> *rows, uint32_t *tables);
}
void build_period_rows(const FilterSpec &f, const PeriodPlan &t, std::vector<float> *rows) {
  rows->assign(t.rows_floats + 3 * t.groups, 0.f);
  std::vector<uint32_t> tables(3 * t.groups);
  build_period_rows_t<float>(f, t, rows, tables.data());
  std::memcpy(rows->data() + t.rows_floats, tables.data(), tables.size() * sizeof(uint32_t));
}
void build_period_rows64(const FilterSpec &f, const PeriodPlan &t, std::vector<double> *rows) {
  rows->assign(t.rows_floats + (3 * t.groups + 1) / 2, 0.0);
  std::vector<uint32_t> tables(3 * t.groups);
  build_period_rows_t<double>(f, t, rows, tables.data());
  std::memcpy(rows->data() + t.rows_floats, tables.data(), tables.size() * sizeof(uint32_t));
}
namespace {
template <typename E>
void build_period_rows_t(const FilterSpec &f, const PeriodPlan &t, std::vector<E> *rows, uint32_t *tables) {
  // taps, then per group two uint32 tables (bit-copied into the float array):
  //   delta'[g] = (g*R*num) div den - k_g   first input frame of the group inside a period
  //   wrap[g]   = iteration before which the window pointer skips the bank padding (~0u: never)
  //   trips[g]  = head | tail << 4 | iterations << 8 (fir_group): iterations of the group's loop, of which the
  //               first `head` touch only rows 0..R/2-1 and the last `tail` only rows R/2..R-1
  const uint32_t it_steps = 2 * bank_taps(t.r) / t.r / ((t.a64 || t.pp) ? 2 : 1);
  const uint32_t gw = t.pp ? 2 * t.r : t.r;  // phases per group (phase pairs: two per accumulator)
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint32_t dg = static_cast<uint32_t>((static_cast<uint64_t>(g) * gw * f.num) / f.den);
    const uint32_t k = group_shift(f, t, g);
    const uint32_t delta = dg - k;
    uint32_t wrap = 0xffffffffu;
    if (t.pad != 0 && delta + t.row_len + 4 > f.num) wrap = (f.num - delta) / 4;
    tables[g] = delta;
    tables[t.groups + g] = wrap;
  }
  static const bool no_trim = SPEEXHIP_DIAG_ENV("SPEEXHIP_NO_TRIM") != nullptr;  // diagnostics: A/B of the trimming
  std::vector<double> h(f.taps);
  for (uint32_t g = 0; g < t.groups; g++) {
    const uint64_t d0 = (static_cast<uint64_t>(g) * gw * f.num) / f.den - group_shift(f, t, g);
    // iterations in which each half of the rows has a non-zero tap: [first, last]
    uint32_t first[2] = {0xffffffffu, 0xffffffffu}, last[2] = {0, 0};
    for (uint32_t i = 0; i < gw; i++) {
      const uint32_t r = g * gw + i;
      if (r >= f.den) continue;  // padding phases of the last group stay zero
      const uint32_t phase = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) % f.den);
      const uint32_t shift = static_cast<uint32_t>((static_cast<uint64_t>(r) * f.num) / f.den - d0);
      const uint32_t half = (t.r == 10 && i >= gw / 2) ? 1 : 0;
      first[half] = std::min(first[half], shift / it_steps);
      last[half] = std::max(last[half], (shift + f.taps - 1) / it_steps);
      phase_taps(f, phase, h.data());
      for (uint32_t j = 0; j < f.taps; j++) {
        const uint32_t s = j + shift;
        (*rows)[(static_cast<size_t>(g) * t.row_len + s) * gw + i] = static_cast<E>(h[j]);
      }
    }
    uint32_t total = std::max(last[0], last[1]) + 1, head = 0, tail = 0;
    if (t.r == 10 && !no_trim) {
      if (first[1] == 0xffffffffu) {  // the group has no row in its second half at all
        total = last[0] + 1;
      } else {
        head = std::min<uint32_t>(first[1], 15);
        tail = std::min<uint32_t>(total - std::min(total, last[0] + 1), 15);
        if (head + tail > total) head = tail = 0;
      }
    }
    if (no_trim) total = t.l4;
    tables[2 * t.groups + g] = head | tail << 4 | total << 8;
  }
}
}  // namespace

namespace {
uint32_t periods_of_launch(const FilterSpec &f, const StreamDesc *h_descs, uint32_t n_streams) {
  uint32_t max_periods = 0;
  for (uint32_t s = 0; s < n_streams; s++) {
    if (h_descs[s].n_out == 0) continue;
    const uint64_t k_end = static_cast<uint64_t>(h_descs[s].k_shift) + h_descs[s].n_out;
    max_periods = std::max<uint32_t>(max_periods, static_cast<uint32_t>((k_end + f.den - 1) / f.den));
  }
  return max_periods;
}

// Shares a tile's phase groups are split into when one workgroup per tile would leave CUs idle.
uint32_t split_count(const PeriodPlan &t, uint32_t tiles, uint32_t n_streams, uint32_t resident) {
  // (measured on cfg2, one stream: 1/2/4 shares -> 18.5/13.6/14.8 us: split until the launch has
  //  about one workgroup per CU, not more -- every share re-stages the window.
  //  Also tried: splitting each group's TAP range over the spare waves of a share, partial sums
  //  meeting in LDS -- +1 us, the two extra barriers cost more than the occupancy gains.)
  static const uint32_t force_splits = SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLITS") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLITS")) : 0;
  if (force_splits) return std::min<uint32_t>(force_splits, t.groups);
  uint32_t splits = 1;
  while (splits * 2 <= t.groups && static_cast<uint64_t>(tiles) * n_streams * splits * 2 <= resident / 2 &&
         (t.groups + splits * 2 - 1) / (splits * 2) >= 2)
    splits *= 2;
  return splits;
}

// probe != null: the launch's shape only (tiles, shares, waves), nothing is launched
struct PeriodShape {
  uint32_t tiles, splits, wave_groups, ksplit, threads, touch;
  bool rounds_3_4 = false;  // in: the shares as rounds 3-4 counted them (the phase-pair rule was fitted on those shapes)
};
hipError_t launch_period_plan(const FilterSpec &f, const PeriodPlan &t, const float *d_rows, uint32_t channels,
                              const StreamDesc *h_descs, const DescPack *pack,
                              uint32_t n_streams, bool float_io, hipStream_t stream, PeriodShape *probe = nullptr,
                              bool fixed_shape = false);
}  // namespace

// Which window for an int16 launch of a ratio that has both?  The int16 window's tiles hold twice the periods
// (all the lanes of a wave busy where the float window leaves half of them idle) at +20-40 % vector instructions
// for the conversions; a launch of a few tiles wants MORE pieces, not fuller ones.  Measured with tap-range
// shares on both sides (profiles/r03_small_decimators.txt; T = tiles of the float-window plan x streams):
// one stream 48k->11.025k stereo 48 000 / 441 000 / 2^20 frames (T = 2 / 25 / 59) 14.6 / 14.9 / 21.1 us on the
// float window against 17.5 / 17.7 / 18.3 on the int16 one; mono 2^20 frames (T = 29) 16.3 vs 20.8; 4 channels
// 441 000 frames (T = 50) 20.9 vs 17.2; 44.1k->16k (which has r = 5 shares) 2^20 frames (T = 57) 14.5 vs 17.0;
// 8 streams x 131 072 frames: stereo 48k->11.025k (T = 64) 36.9 vs 30.7, mono (T = 32) 28.1 vs 21.0, 44.1k->16k
// (T = 64) 24.2 vs 17.2.  Hence: the int16 window when the float plan fills the chip anyway, from T = 48 when
// the ratio has no r = 5 plan, and from T = 32 when the launch is a batch of at least 4 streams.
bool period_launch_prefers_w16(const FilterSpec &f, const PeriodPlan &t, bool has_fine, const StreamDesc *h_descs,
                               uint32_t n_streams) {
  if (!t.float_ok) return true;  // (there is no float window to prefer)
  if (t.fold != f.fold) {
    const FoldedLaunch v(f, t.fold, h_descs, nullptr, n_streams);
    return period_launch_prefers_w16(v.f, t, has_fine, v.pack.d, n_streams);
  }
  const uint32_t max_periods = periods_of_launch(f, h_descs, n_streams);
  const uint32_t tiles = (max_periods + t.lane_periods - 1) / t.lane_periods;
  if (split_count(t, tiles, n_streams, 2 * device_compute_units()) == 1) return true;
  const uint64_t T = static_cast<uint64_t>(tiles) * n_streams;
  return (T >= 48 && !has_fine) || (n_streams >= 4 && T >= 32);
}

// Phase pairs (mono; plan_period_r, FirLoopAsmPP) against two-period lanes, same box, profiles/r04_pp_ab.txt and
// r04_pp_ab2.txt (launch us, two-period -> phase pairs):
//   32 streams x 131 072 frames: 48k->11.025k 57.0 -> 43.1, 48k->22.05k 44.5 -> 35.7, 44.1k->32k 46.2 -> 31.9,
//     44.1k->8k 50.3 -> 32.2, 44.1k->16k 42.6 -> 29.2, 32k->44.1k 48.0 -> 42.1, 96k->44.1k q5 33.9 -> 27.8;
//     8 streams: 21.8 -> 18.5, 23.2 -> 12.9, 19.1 -> 16.1, 18.8 -> 12.7, 18.4 -> 12.0, 25.1 -> 19.3, 20.1 -> 11.9;
//   32 streams x 2^20 frames: 44.1k->32k 210.5 -> 160.6 and 44.1k->16k 189.6 -> 142.5 (two-period tiles fill 86 of 128
//     lane slots there), 44.1k->8k 165.5 -> 165.2, but 48k->22.05k 195.4 -> 216.5, 96k->44.1k 144.3 -> 161.6 and
//     48k->11.025k (int16 window, 116 periods) 179.5 -> 326.7: a phase-pair loop has 10 packed FMAs between two waits,
//     not 20, and twice the tap bytes (tools/ubench_pair.hip: 105 against 120 TF at full occupancy);
//   one stream: -15 % ... +18 %, no pattern; narrow windows (44.1k->48k, 48k->44.1k): two-period lanes 10-45 % ahead.
// Hence: plans for wide windows only (num >= 320: 128 periods of two-period lanes cannot share a CU with a second
// workgroup), and a launch takes them when it is a batch that two-period tiles leave at most one per CU, or a launch
// of several generations whose two-period plan fills under three quarters of its lane slots.  SPEEXHIP_PP=0 / 1:
// never / whenever planned.
bool period_wants_pp_plans(const FilterSpec &f, uint32_t channels) {
  static const int env_pp = SPEEXHIP_DIAG_ENV("SPEEXHIP_PP") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_PP")) : -1;
  if (channels > 3 || env_pp == 0) return false;
  // (a full tile of the other lanes -- 128 periods of mono, 64 of stereo, 42 of three channels -- is 512 num bytes
  //  of window: from num = 320 it cannot share a CU)
  return env_pp == 1 || static_cast<size_t>(f.num) * 4 * 128 >= 160 * 1024;
}
bool period_launch_prefers_pp(const FilterSpec &f, const PeriodPlan &two, const PeriodPlan &pp, const StreamDesc *h_descs,
                              uint32_t n_streams) {
  static const int env_pp = SPEEXHIP_DIAG_ENV("SPEEXHIP_PP") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_PP")) : -1;
  if (env_pp >= 0) return env_pp != 0;
  if (two.fold != f.fold) {
    const FoldedLaunch v(f, two.fold, h_descs, nullptr, n_streams);
    return period_launch_prefers_pp(v.f, two, pp, v.pack.d, n_streams);
  }
  const uint32_t cus = device_compute_units();
  const uint32_t slots = 64 / two.cgroups * (two.ct == 1 ? 2 : 1);  // periods a full tile of the other plan holds
  const bool unfilled = 4 * two.lane_periods < 3 * slots;
  if (two.cgroups == 1 && two.ct == 1) {  // mono (profiles/r04_pp_ab2.txt)
    const uint32_t max_periods = periods_of_launch(f, h_descs, n_streams);
    const uint64_t tiles = static_cast<uint64_t>((max_periods + two.lane_periods - 1) / two.lane_periods) * n_streams;
    if (tiles <= cus) return n_streams >= 2;  // (one stream: -15 % ... +18 %, no pattern: two-period stays)
    return unfilled;                          // several generations: only where lanes go unused
  }
  // Stereo and three channels (profiles/r04_pp_ab3.txt): phase pairs win where they keep at least as many waves on a
  // SIMD as the other lanes would and at least ~4 -- their loop has 10 packed FMAs between two waits, not 20: at two
  // waves per SIMD stereo 48k->11.025k took 160 us instead of 79 --, in batches of about a generation (48k->22.05k
  // 62.8 -> 40.9 us, 44.1k->8k 72.7 -> 52.2, three channels 44.1k->32k 117.8 -> 66.2; 8 streams 25-44 -> 16-26), and
  // in longer launches only with twice that and lanes the other plan leaves unused (three channels 44.1k->32k, 32 x 2^20
  // frames: 642 -> 459 us; 44.1k->16k at 4 waves per SIMD lost, 578 -> 681).
  if (n_streams < 2) return false;
  PeriodShape so{}, sp{};
  so.rounds_3_4 = sp.rounds_3_4 = true;
  if (launch_period_plan(f, two, nullptr, two.ct * two.cgroups, h_descs, nullptr, n_streams, false, nullptr, &so) != hipSuccess ||
      launch_period_plan(f, pp, nullptr, pp.cgroups, h_descs, nullptr, n_streams, false, nullptr, &sp) != hipSuccess)
    return false;
  auto waves_per_simd = [&](const PeriodPlan &t, const PeriodShape &sh) {
    const uint64_t wgs = static_cast<uint64_t>(sh.tiles) * n_streams * sh.splits;
    const uint64_t per_cu = (wgs + cus - 1) / cus;
    const uint64_t fit = std::max<uint64_t>(1, std::min<uint64_t>((160 * 1024) / std::max<size_t>(t.window_bytes, 1), 32 / std::max<uint32_t>(sh.threads / 64, 1)));
    // (the tap-range shares of an UNSPLIT launch count only against a plan that leaves lanes unused: two half-chains
    //  are not two chains -- stereo 48k->11.025k in phase pairs with them: 90 us against 79 the other way, whose tiles
    //  are full; three channels, 18 of 42 periods per tile the other way: 32 x 2^20 frames 870 -> 400 us)
    static const bool count_ks = SPEEXHIP_DIAG_ENV("SPEEXHIP_PP_COUNT_KS") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_PP_COUNT_KS")) != 0;  // A/B
    return static_cast<double>(std::min(per_cu, fit)) * sh.wave_groups * ((sh.splits > 1 || unfilled || count_ks) ? sh.ksplit : 1u) / 4.0;
  };
  // (Late in round 4: where the other plan has to SPLIT its tiles over two workgroups -- each stages the whole 150 KB window
  //  -- and the phase-pair plan runs unsplit with tap-range shares and its rows fetched behind the window (launch_period_plan):
  //  stereo 48k->11.025k, 32 x 131 072 frames: 76.6 us the other way, 89.6 in phase pairs without the fetch, 46.2 with;
  //  profiles/r04_pp_touch_ab.txt.  Counting the shares in the rule below instead moved six more stereo rows to phase
  //  pairs, +6 % each: profiles/r04_rule3_ab.txt.)
  static const bool split_rule_off = SPEEXHIP_DIAG_ENV("SPEEXHIP_PP_SPLIT_RULE") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_PP_SPLIT_RULE")) == 0;  // A/B
  if (!split_rule_off && so.splits > 1 && sp.splits == 1 && sp.ksplit > 1) return true;
  const double wo = waves_per_simd(two, so), wp = waves_per_simd(pp, sp);
  if (wp < 3.5 || wp < wo) return false;
  const uint64_t wgs_other = static_cast<uint64_t>(so.tiles) * n_streams * so.splits;
  return wgs_other <= 2ull * cus || (unfilled && wp >= 6.0);
}

// Host only (speexhip_debug_launch_shape, tests): the shape launch_period_plan would give this launch.
bool debug_period_shape(const FilterSpec &f, const PeriodPlan &t, uint32_t channels, const StreamDesc *h_descs, uint32_t n_streams,
                        bool float_io, uint32_t out[6]) {
  if (t.fold != f.fold) {
    const FoldedLaunch v(f, t.fold, h_descs, nullptr, n_streams);
    return debug_period_shape(v.f, t, channels, v.pack.d, n_streams, float_io, out);
  }
  PeriodShape sh{};
  if (launch_period_plan(f, t, nullptr, channels, h_descs, nullptr, n_streams, float_io, nullptr, &sh) != hipSuccess) return false;
  out[0] = sh.tiles;
  out[1] = sh.splits;
  out[2] = sh.wave_groups;
  out[3] = sh.ksplit;
  out[4] = sh.threads;
  out[5] = sh.touch;
  return true;
}

// `fine` (may be null / unusable): the same filter planned with R = 5.  A launch that is a single
// generation of workgroups -- one that the R = 10 plan would have to split into shares whose
// workgroups run 8 FIR waves (2 per SIMD) beside 8 staging helpers -- takes it instead: 16 FIR
// waves per workgroup, 4 per SIMD, each with half the phases (cfg2, one stream: 12.95 -> 12.09 us;
// in launches of several generations R = 10 wins: 32 streams 210 vs 225 us).
hipError_t launch_period(const FilterSpec &f, const PeriodPlan &t, const float *d_rows, const PeriodPlan *fine,
                         const float *d_rows_fine, uint32_t channels, const StreamDesc *h_descs,
                         const DescPack *pack, uint32_t n_streams, bool float_io,
                         hipStream_t stream, bool fixed_shape) {
  if (t.fold != f.fold) {  // a plan made on a folded view of this filter (period_view)
    const FoldedLaunch v(f, t.fold, h_descs, pack, n_streams);
    return launch_period(v.f, t, d_rows, fine, d_rows_fine, channels, v.pack.d, &v.pack, n_streams, float_io, stream, fixed_shape);
  }
  if (fine != nullptr && fine->usable && d_rows_fine != nullptr) {
    const uint32_t max_periods = periods_of_launch(f, h_descs, n_streams);
    const uint32_t tiles = (max_periods + t.lane_periods - 1) / t.lane_periods;
    if (split_count(t, tiles, n_streams, 2 * device_compute_units()) > 1)
      return launch_period_plan(f, *fine, d_rows_fine, channels, h_descs, pack, n_streams, float_io, stream, nullptr, fixed_shape);
  }
  return launch_period_plan(f, t, d_rows, channels, h_descs, pack, n_streams, float_io, stream, nullptr, fixed_shape);
}

namespace {
hipError_t launch_period_plan(const FilterSpec &f, const PeriodPlan &plan, const float *d_rows, uint32_t channels,
                              const StreamDesc *h_descs, const DescPack *pack,
                              uint32_t n_streams, bool float_io, hipStream_t stream, PeriodShape *probe, bool fixed_shape) {
  PeriodPlan t = plan;
#ifdef SPEEXHIP_DIAG
  {  // A/B: fewer periods per tile than the plan's (more, smaller workgroups over the same LDS allocation)
    const int tp = diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_TILE_PERIODS"), 0);
    if (tp > 0 && static_cast<uint32_t>(tp) < t.lane_periods && probe == nullptr) t.lane_periods = static_cast<uint32_t>(tp);
  }
#endif
  const uint32_t max_periods = periods_of_launch(f, h_descs, n_streams);
  const uint32_t tiles = (max_periods + t.lane_periods - 1) / t.lane_periods;
  const uint32_t resident = 2 * device_compute_units();  // two workgroups fit per CU
  // One workgroup per (tile, stream); the hardware dispatcher refills a CU the moment a workgroup
  // retires.  (A persistent walk of the tile list with the next tile's window prefetched into
  // registers lived here until round 1's last measurements at sustained clocks: 7 % slower at 32
  // streams -- 228 vs 212 us --, 25 % slower on the 8-channel configuration.  Round 2 built it once
  // more -- 512 resident workgroups walking tiles tile_step apart, the next tile's input lines
  // touched into L2 during the FIR loop, no slot turnover, the history roll folded into workgroup 0 --
  // and measured 224 vs 205 us at 32 streams, 703 vs 604 us on 8 channels: a tile's FIR time varies by
  // +-20 % (profiles/r02_stamps_cfg2_s32.txt), so a static share of 7 tiles per workgroup ends with
  // the slowest of 512, where the dispatcher hands the next tile to whichever CU is free; and what a
  // fresh workgroup costs between two FIR loops is hidden behind the other workgroup's FIR anyway.
  // Round 3 took the static share out of the argument: 512 workgroups drawing their next tile from a
  // device counter (the atomic issued at the top of a tile, in flight behind its staging loads; the
  // number handed round through LDS behind two bare barriers; no wait for the tile's stores).  Results
  // equal; 32 streams 195 -> 204-206 us, 64 streams 395 -> 408, mono 123 -> 143, FIR only 162 -> 169
  // (profiles/r03_walk_ab.txt).  The reason is phase: the two workgroups the dispatcher keeps on a CU
  // start half a tile apart and stay there (stamps: neighbour offset 10-12 us of a 25 us cycle), so one
  // stages and stores while the other is in its FIR loop -- and a workgroup alone on its CU runs its loop
  // 1.65x faster than each of two (12.4 vs 20.5 us); walkers start together and stay in step: both in
  // their FIR loops, then both in their memory phases with nothing issuing FMAs.  What a walk could save
  // once staggered (slot turnover 1.5 us + half the prologue per tile) is ~3 % of a launch; the one-stream
  // launch lost 0.4 us to the loop around the kernel body.  Removed.)
  // When one workgroup per tile leaves CUs idle (one short stream), the phase groups of a tile are
  // split over several workgroups.
  // (Phase costs of the single-stream launch, R = 10, rocprofv3 with parts skipped,
  //  profiles/r01_phases_cfg2_s1.txt: bare dispatch 1.7, descriptor + geometry 1.0, window staging 1.9,
  //  FIR loop 6.6, stores 3.0 -- 14.8 us if serial against 13.2 us measured: a launch that is a
  //  single generation of workgroups overlaps very little.)
  // (Round 3 tried to give a one-generation launch twice the waves by sharing every R = 5 group between TWO waves,
  //  half of the group's tap range each, partial sums meeting in LDS behind two barriers: 8 groups per workgroup,
  //  4 workgroups per tile, two resident per CU = 8 waves per SIMD instead of 4, with the R = 5 instances brought
  //  under 64 VGPRs on 20-tap banks for it.  Results equal within +-1 LSB, and slower: cfg2 one stream 12.05 ->
  //  12.73 us, float 13.9 -> 14.8, mono 13.6 -> 14.0.  The dispatcher starts the 448 workgroups of 16 waves over
  //  3.25 us (224: 0.65 us) -- ~2 200 waves per us --, every workgroup stages the whole 76 KB window again, and the
  //  20-tap banks alone cost R = 5 0.2 us (44.1k->8k at 32 streams 304 -> 320 us).  A second form kept the workgroup
  //  count and let the eight HELPER waves of a workgroup that owns <= 8 groups compute the second half of each
  //  group's trips instead of leaving: mono one stream 13.22 -> 13.94 us, a 441 000-frame stereo call 8.52 -> 9.02,
  //  float 9.37 -> 10.10; only launches of a few tiles gained (16 384 frames 7.14 -> 6.36 us): the two barriers and
  //  the pass through LDS cost more than the halved loop saves.  Removed; profiles/r03_ab_ksplit.txt.)
  static const uint32_t max_waves = SPEEXHIP_DIAG_ENV("SPEEXHIP_WAVES") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_WAVES")) : 16;
  uint32_t splits = split_count(t, tiles, n_streams, resident);
  // split_count doubles the shares by workgroup count alone.  Where that leaves a share more phase groups than a
  // workgroup has waves (8k -> 44.1k: 45 groups in 2 shares, the waves walk two groups each: 29.5 us for one
  // stream of 441 000 frames against 20.8 in 5 shares), a few more shares are weighed with a small cost model
  // fitted to the launches of profiles/r03_small_decimators.txt: a wave alone on its SIMD spends ~19 cycles per
  // packed FMA, w of them together 4.75 w; a workgroup takes its window in at ~11 bytes per cycle; workgroups
  // beyond one per CU queue.  A candidate must beat the incumbent by 10 %.
  static const bool model_off = SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLIT_MODEL") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLIT_MODEL")) == 0;  // A/B (0: neither model)
  if (splits > 1 && !model_off && !SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLITS")) {
    const double cus = device_compute_units();
    auto cost = [&](uint32_t s) {
      const double wg_per_cu = std::ceil(static_cast<double>(tiles) * n_streams * s / cus);
      const uint32_t gps = (t.groups + s - 1) / s;                 // groups per share
      const double walks = std::ceil(static_cast<double>(gps) / max_waves);
      double chain = static_cast<double>(t.r) * t.row_len, waves = std::min<uint32_t>(gps, max_waves);
      if (gps * 2 <= max_waves && chain >= 1800) {                  // tap-range shares (below)
        const double parts = std::min<double>(max_waves / gps, std::max<uint32_t>(t.l4 / 4, 1));
        chain /= parts;
        waves = gps * parts;
      }
      const double per_simd = wg_per_cu * waves / 4;
      return walks * chain * std::max(19.0, 4.75 * per_simd) + wg_per_cu * t.window_bytes / 11.0;
    };
    uint32_t best = splits;
    for (uint32_t s2 = splits + 1; s2 <= 2 * splits + 1 && s2 <= 16 && (t.groups + s2 - 1) / s2 >= 2; s2++)
      if (cost(s2) < 0.9 * cost(best)) best = s2;
    splits = best;
  }
  // Windows of more than half the LDS (one workgroup per CU; round 5).  There the launch runs in whole GENERATIONS of
  // workgroups -- the history-roll block of every (stream, share) is a workgroup too and asks for the same LDS -- and
  // the doubling above neither counts those nor knows three shares: 8 streams x 131 072 frames of 4-channel 32k ->
  // 11.025k is 8 tiles + 1 per stream, x 4 shares = 288 workgroups on 256 CUs, 70 us against 48 in 3 shares (216); 32
  // mono streams in 2 tiles each took 4 shares, 384 workgroups, 85 us against 49 in 2.  The shares by a model of the
  // generations instead, fitted to profiles/r05_wide_grid.txt (tools/ab.sh over SPEEXHIP_SPLITS: 27 launches x 7 split counts): a
  // workgroup stages its window at ~5 bytes per cycle (all CUs at once) and then spends 4.75 cycles per packed FMA and
  // SIMD on the groups of its share, its slowest wave at least `walks` chains; within +-15 % of the measured launches,
  // argmin within 6 % of the best measured on 25 of the 27.  A candidate has to beat fewer shares by 10 %.
  static const bool wide_model_off = SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLIT_MODEL") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLIT_MODEL")) <= 1;  // A/B: 1 = rounds 3-4
  // Only where the count above does not fit one generation: inside one, its shares (and the tap-range shares they allow)
  // are the better-fitted choice -- the model in their place lost 25-40 % on one-stream and mono launches
  // (profiles/r05_ab_split_model.txt, first table).
  if (t.window_bytes > 80 * 1024 && !wide_model_off && !SPEEXHIP_DIAG_ENV("SPEEXHIP_SPLITS") && max_periods != 0 &&
      !(probe != nullptr && probe->rounds_3_4) &&
      static_cast<uint64_t>(tiles) * n_streams * splits + n_streams > device_compute_units()) {
    // (of the history-roll blocks only those of share 0 do anything; the others leave at once)
    const double cus = device_compute_units();
    const double chain = static_cast<double>(t.r) * t.row_len * (t.a64 ? 2 : 1) * 4.75 * (t.w16 ? 1.2 : 1.0);
    auto cost = [&](uint32_t s) {
      const double gens = std::ceil((static_cast<double>(tiles) * n_streams * s + n_streams) / cus);
      const uint32_t gps = (t.groups + s - 1) / s;
      const double walks = std::ceil(static_cast<double>(gps) / max_waves);
      // (tap-range shares, below: the waves a share leaves idle take pieces of its groups' chains)
      const double parts = (gps * 2 <= max_waves && static_cast<uint64_t>(t.r) * t.row_len * (t.a64 ? 2 : 1) >= 1800) ? max_waves / gps : 1;
      return gens * (t.window_bytes / 5.0 + std::max(gps / 4.0, walks / parts) * chain);
    };
    uint32_t best = 1;
    for (uint32_t s2 = 2; s2 <= 16 && (t.groups + s2 - 1) / s2 >= 2; s2++)
      if (cost(s2) < 0.9 * cost(best)) best = s2;
    splits = best;
  }
  const uint32_t wave_groups = std::min<uint32_t>((t.groups + splits - 1) / splits, max_waves);
  PeriodParams p;
  p.rows = d_rows;
  p.delta = d_rows == nullptr ? nullptr  // (a probe of the launch shape: period_launch_prefers_pp)
            : t.a64           ? reinterpret_cast<const uint32_t *>(reinterpret_cast<const double *>(d_rows) + t.rows_floats)
                              : reinterpret_cast<const uint32_t *>(d_rows + t.rows_floats);
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
  p.pad = t.pad;
  p.half_periods = (t.ct == 1 && !t.pp) ? (t.lane_periods + 1) / 2 : 0;
  p.half_offset = p.half_periods * (f.num * channels + t.pad);
  p.wrap_step = f.num % 4 == 0 ? f.num / 4 : 0x40000000u;
  p.period_magic = period_magic_of(f.num * channels);
  p.history_block = max_periods == 0 ? 0 : tiles;
  // (The tail: a launch of several generations ends ragged -- profiles/r02_stamps_cfg2_s32.txt: the last
  //  tenth of the workgroups finish over 22 of 204 us; 64 streams take 404 us against 209 for 32, so ~14 us
  //  of a launch are start + tail.  Tried: the tiles of the last 1-12 streams -- the last ones dispatched --
  //  as 2 or 4 shares of their phase groups each, like the tiles of a one-stream launch.  16 streams
  //  114.3 -> 111.1 us with one stream shared out, 32 streams within the +-1.5 % of repeated runs for 1-3
  //  streams and slower beyond (8: 219 vs 213 us; 4 shares always slower), 64 streams 404 -> 407: a share
  //  stages the whole window again and leaves its CU with half the FIR waves; removed.)
  p.threads = 0;  // set below
  // bit 0: prologue + staging raised, bit 1: stores raised, bit 2 (measured slower, see set_fir_priority): the two
  // workgroups of a CU at different FIR priorities
  static const int env_prio = SPEEXHIP_DIAG_ENV("SPEEXHIP_PRIO") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_PRIO")) : 3;
  p.prio = static_cast<uint32_t>(env_prio);
#ifdef SPEEXHIP_DIAG
  static const uint32_t skip_mask = static_cast<uint32_t>(diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_SKIP"), 0));
  p.skip = skip_mask;
#endif
  // touch_rows (kernels_period_impl.h): every workgroup fetches the tap rows into L2 once its window is in LDS.  Pays
  // where the rows are NOT in L2 when the FIR loop starts -- a first call, or a launch whose own samples (and the
  // launch before it) replace the L2s' 32 MB -- and the table is large; costs a trip to HBM per workgroup (~2 us,
  // hidden where a second workgroup shares the CU) where they are.  On warm / cold caches (a 64 MB read between two
  // launches), without -> with, profiles/r04_cold_touch.txt: stereo 48k->11.025k 32 x 131 072 frames 80.9 / 119.8 ->
  // 82.8 / 82.8 us, mono 48k->22.05k 30.0 / 53.0 -> 31.0 / 31.6, 4 channels 145 / 148 -> 136 / 135, 3 channels 32 x 2^20
  // frames 496 -> 393, stereo 327 -> 277; one stream of cfg2 14.0 / 16.6 -> 14.2 / 15.4, of 44.1k->48k q10 35.8 / 40.5
  // -> 37.9 / 38.6.  So: launches that move >= 24 MB (their rows never survive to the next launch) with >= 128 KB of
  // rows (6 channels 44.1k->8k, 250 KB: 111.8 -> 95.9 us; three channels 44.1k->16k, 150 KB: 60.7 -> 55.9; cfg2's 90 KB
  // at 32 streams 190.5 / 189.0, cfg4's 546.8 / 551.1: profiles/r04_touch_ab3.txt); a caller whose launches are smaller
  // but far apart can force it (SPEEXHIP_TOUCH=1).  And: unsplit phase-pair launches with tap-range shares of about a
  // generation or more whose rows are >= 256 KB -- those rows never survived from one launch to the next (stereo
  // 48k->11.025k, 415 KB, 20.7 MB moved: 89.6 us without, 46.2 with; at 210 KB, stereo 48k->22.05k: 40.1 -> 42.3, so not
  // there: profiles/r04_pp_touch_ab.txt, r04_rule4_ab.txt).
  static const int env_touch = SPEEXHIP_DIAG_ENV("SPEEXHIP_TOUCH") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_TOUCH")) : -1;  // A/B
  {
    const size_t rows_bytes = t.rows_floats * (t.a64 ? 8 : 4);
    uint64_t moved = 0;  // bytes in + out
    for (uint32_t i = 0; i < n_streams; i++)
      moved += (static_cast<uint64_t>(h_descs[i].in_frames) + h_descs[i].n_out) * channels * (float_io ? 4 : 2);
    // (A/B: SPEEXHIP_TOUCH_RULE=2 fetches in every launch of half a generation or more -- stereo 48k->11.025k the same
    //  46 us, ten more rows of the sweep +3...7 % on warm caches: profiles/r04_rule2_ab.txt)
    static const bool wide_rule = SPEEXHIP_DIAG_ENV("SPEEXHIP_TOUCH_RULE") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_TOUCH_RULE")) == 2;
    const bool half_generation = 2ull * tiles * n_streams * splits >= device_compute_units();
    static const bool pp_rule = !(SPEEXHIP_DIAG_ENV("SPEEXHIP_TOUCH_RULE") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_TOUCH_RULE")) == 1);  // A/B: 1 = by bytes only
    // (set below: the tap-range shares; an unsplit launch of 8-wave groups takes them whenever its chain is long)
    const bool pp_shares = t.pp && splits == 1 && t.r == 10 && wave_groups * 2 <= max_waves;
    const bool wanted = env_touch >= 0 ? env_touch != 0
                                       : rows_bytes >= 128 * 1024 && (moved >= (24ull << 20) || (wide_rule && half_generation) ||
                                                                     (pp_rule && pp_shares && half_generation && rows_bytes >= 256 * 1024));
    p.touch = wanted ? 1u : 0u;
  }
  // A workgroup that owns only a share of the groups still stages the whole window: lend it the
  // waves it has no groups for, they leave after the staging barrier.
  static const int env_helpers = SPEEXHIP_DIAG_ENV("SPEEXHIP_HELPERS") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_HELPERS")) : 1;
  const bool helpers = env_helpers != 0 && splits > 1;
  // Tap-range shares (fir_tile_parts): a split launch whose waves each carry a long chain -- R x row_len packed
  // FMAs, alone on their SIMD -- gives every group as many waves as the workgroup has room for.  From 1 800 FMAs
  // per wave (44.1k -> 16k in R = 5 shares, 1 900, gains 2.3 us of 10.8; 44.1k -> 48k q10, 1 400, loses 0.2 us of
  // 8.7; BASELINE configs[1]'s one-generation plan, R = 5 x 140 = 700, lost 0.7 us of 13.2 with the same scheme), with
  // at least 4 trips per part, the partial sums inside the window, an ISA loop for the layout (frames of 1, 2,
  // 4, 6, 8 channels) and one group per wave-set.  SPEEXHIP_KSPLIT=0 turns it off, =n forces n parts (A/B, tests).
  static const int env_ksplit = SPEEXHIP_DIAG_ENV("SPEEXHIP_KSPLIT") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_KSPLIT")) : -1;
  p.ksplit = 1;
#ifdef SPEEXHIP_CXX_FIR_LOOP
  const bool isa_layout = false;  // (the A/B library without the ISA loop has no tap-range shares either)
#else
  const bool isa_layout = t.pp || (t.ct == 1 && (t.cgroups == 1 || t.cgroups == 3 || t.cgroups == 5 || t.cgroups == 7)) || (t.ct == 2 && (t.cgroups <= 6 || t.cgroups == 8));
#endif
  // (Round 4, late: an UNSPLIT launch whose workgroups have at most 8 waves takes the shares too.  The FIR loop waits
  //  ~500 cycles for every bank of taps -- a scalar load that misses to L2 -- and only other waves cover that: stamps of
  //  32 x 131 072 frames of 3-channel 48k -> 11.025k, 8 waves per workgroup, one or two workgroups per CU: every
  //  wave spends 355 000 cycles on 6 480 packed FMAs, 55 cycles each, whether it shares its CU or not
  //  (profiles/r04_stamps_3ch_pp.txt).  Two waves per group: 157 -> 92 us; mono 48k -> 22.05k 35.5 -> 26.2,
  //  3 channels 44.1k -> 16k 89 -> 61: profiles/r04_ks_ab.txt.)
  static const bool unsplit_ks_off = SPEEXHIP_DIAG_ENV("SPEEXHIP_KS_UNSPLIT") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_KS_UNSPLIT")) == 0;  // A/B
  // (R = 10 only: the R = 5 instances with shares take 76 VGPRs -- one 16-wave workgroup per CU where two of 8 waves
  //  ran: stereo 44.1k->8k in phase pairs 52.5 -> 57 us)
  const bool unsplit_ks = splits == 1 && t.r == 10 && !unsplit_ks_off;
  if (!fixed_shape && (splits > 1 || unsplit_ks || env_ksplit > 0) && env_ksplit != 0 && isa_layout && wave_groups * splits >= t.groups && wave_groups * 2 <= max_waves) {
    uint32_t parts = env_ksplit > 0 ? static_cast<uint32_t>(env_ksplit) : max_waves / wave_groups;
    parts = std::min<uint32_t>(parts, max_waves / wave_groups);
    const uint32_t trips = t.l4;  // trips per group row
    // (the partial sums overwrite the window: R x 64 pairs of floats per wave, or of doubles with an fp64 accumulator)
    while (parts > 1 && (trips / parts < 4 ||
                         static_cast<size_t>(parts - 1) * wave_groups * t.r * 64 * (t.a64 ? 16 : 8) > t.window_bytes))
      parts--;
    // (two parts on a chip already more than half full buy nothing: there the launch is throughput, not one
    //  wave's latency -- 32 mono streams x 131 072 frames of 48k -> 22.05k in 2 shares: 44.8 us without, 48.7 with)
    static const bool crowded_off = SPEEXHIP_DIAG_ENV("SPEEXHIP_KS_CROWDED") && std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_KS_CROWDED")) == 0;  // A/B
    const bool crowded = !crowded_off && static_cast<uint64_t>(tiles) * n_streams * splits * 2 > device_compute_units();
    // (a wave's chain in vector instructions: one packed FMA per tap, two v_fma_f64 with an fp64 accumulator)
    if (parts > 1 && (env_ksplit > 0 || (static_cast<uint64_t>(t.r) * t.row_len * (t.a64 ? 2 : 1) >= 1800 && (unsplit_ks || !(crowded && parts < 3)))))
      p.ksplit = parts;
  }
  const uint32_t threads =
      (p.ksplit > 1 ? wave_groups * p.ksplit : helpers ? std::max<uint32_t>(wave_groups, max_waves) : wave_groups) * 64;
  p.threads = threads;
  if (probe != nullptr) {
    *probe = PeriodShape{tiles, splits, wave_groups, p.ksplit, threads, p.touch, probe->rounds_3_4};
    return hipSuccess;
  }
  // Grid: x = tiles + 1 (the extra block rolls the history), padded to a multiple of 8
  // when a tile is split: workgroups whose linear ids differ by a multiple of 8 share an XCD, so
  // the `splits` workgroups that stage the same input window hit in that XCD's L2 instead of
  // each fetching it from HBM (measured 2.7x read amplification without this).
  uint32_t grid_x = (max_periods == 0 ? 0 : tiles) + 1;
  if (splits > 1 && n_streams == 1) grid_x = (grid_x + 7) / 8 * 8;
  const dim3 grid(grid_x, n_streams, splits);
  // diagnostics: one line per launch shape on stderr
  static const bool verbose = SPEEXHIP_DIAG_ENV("SPEEXHIP_PLAN_VERBOSE") != nullptr;
  if (verbose) {
    static uint64_t last = 0;
    const uint64_t key = (static_cast<uint64_t>(tiles) << 40) ^ (static_cast<uint64_t>(n_streams) << 24) ^ (splits << 16) ^ (t.r << 8) ^ p.ksplit ^ (t.w16 ? 1u << 31 : 0u);
    if (key != last) {
      last = key;
      std::fprintf(stderr, "period launch: r=%u%s groups=%u lane_periods=%u tiles=%u streams=%u splits=%u wave_groups=%u parts=%u threads=%u window=%zu B\n",
                   t.r, t.w16 ? " (int16 window)" : "", t.groups, t.lane_periods, tiles, n_streams, splits, wave_groups, p.ksplit, threads, t.window_bytes);
    }
  }
  // (an int16 window -- t.w16 -- exists for int16 calls on the layouts the ISA loop is generated for: ONE or CGV)
  if (t.w16 && float_io) return hipErrorInvalidValue;
  if (!t.float_ok) return hipErrorInvalidValue;  // (a plan that only carries its int16 plan: engine.cpp never launches it)
  if (t.a64)  // kernels_period64.hip / kernels_period64_w16.hip
    return t.w16 ? dispatch_period64_w16(t, p, pack, grid, threads, float_io, stream)
                 : dispatch_period64(t, p, pack, grid, threads, float_io, stream);
  if (t.pp) return dispatch_period_pp(t, p, pack, grid, threads, float_io, stream);   // kernels_period_pp.hip
  if (t.ct == 1 && (t.cgroups == 3 || t.cgroups == 5 || t.cgroups == 7)) return dispatch_period_odd(t, p, pack, grid, threads, float_io, stream);
  if (t.ct == 2 && (t.cgroups == 5 || t.cgroups == 6 || t.cgroups == 8)) return dispatch_period_frames(t, p, pack, grid, threads, float_io, stream);
  // (round 6: the int16 window of the layouts without an ISA loop -- frames of 9, 11, 13-15, 17 ... channels, C++ loop)
  if (t.w16 && !(t.cgroups == 1 || (t.ct == 2 && t.cgroups <= 4))) return dispatch_period_w16g(t, p, pack, grid, threads, stream);
  // (LRC: the kernel of the layout, or its tap-range-shares twin)
#define SPEEXHIP_LRCW(RV, CTV, ONE, PADV, TV, CGV, W)                                                                                    \
  (p.ksplit > 1 ? launch_rc<RV, CTV, ONE, PADV, TV, CGV, W, true>(p, pack, grid, threads, t.window_bytes, stream)              \
                : launch_rc<RV, CTV, ONE, PADV, TV, CGV, W, false>(p, pack, grid, threads, t.window_bytes, stream))
#define SPEEXHIP_LRC(RV, CTV, ONE, PADV, TV, CGV) SPEEXHIP_LRCW(RV, CTV, ONE, PADV, TV, CGV, false)
#define SPEEXHIP_PERIOD_CASE_R(RV, CTV, ONE, PADV)                                                                                      \
  return float_io ? SPEEXHIP_LRC(RV, CTV, ONE, PADV, float, 0)                                                                           \
         : (ONE && t.w16) ? SPEEXHIP_LRCW(RV, CTV, ONE, PADV, int16_t, 0, ONE)                                                         \
                          : SPEEXHIP_LRC(RV, CTV, ONE, PADV, int16_t, 0)
  // 4 / 6 / 8 channels: channel pairs per frame as a compile-time constant (lane_ctx)
#define SPEEXHIP_PERIOD_CASE_CG(RV, PADV, CGV)                                                                                          \
  return float_io ? SPEEXHIP_LRC(RV, 2, false, PADV, float, CGV)                                                                         \
         : t.w16  ? SPEEXHIP_LRCW(RV, 2, false, PADV, int16_t, CGV, true)                                                              \
                  : SPEEXHIP_LRC(RV, 2, false, PADV, int16_t, CGV)
  const bool padded = t.pad != 0;
  if (t.ct == 2 && t.cgroups >= 2 && t.cgroups <= 4) {
    if (t.r == 5) {
      if (t.cgroups == 2) SPEEXHIP_PERIOD_CASE_CG(5, false, 2);
      if (t.cgroups == 3) SPEEXHIP_PERIOD_CASE_CG(5, false, 3);
      SPEEXHIP_PERIOD_CASE_CG(5, false, 4);
    }
    if (!padded) {
      if (t.cgroups == 2) SPEEXHIP_PERIOD_CASE_CG(10, false, 2);
      if (t.cgroups == 3) SPEEXHIP_PERIOD_CASE_CG(10, false, 3);
      SPEEXHIP_PERIOD_CASE_CG(10, false, 4);
    }
    if (t.cgroups == 2) SPEEXHIP_PERIOD_CASE_CG(10, true, 2);
    if (t.cgroups == 3) SPEEXHIP_PERIOD_CASE_CG(10, true, 3);
    SPEEXHIP_PERIOD_CASE_CG(10, true, 4);
  }
#undef SPEEXHIP_PERIOD_CASE_CG
  if (t.r == 5) {  // never padded (plan_period_r)
    if (t.ct == 2 && t.cgroups == 1) SPEEXHIP_PERIOD_CASE_R(5, 2, true, false);
    if (t.ct == 2) SPEEXHIP_PERIOD_CASE_R(5, 2, false, false);
    if (t.cgroups == 1) SPEEXHIP_PERIOD_CASE_R(5, 1, true, false);
    SPEEXHIP_PERIOD_CASE_R(5, 1, false, false);
  }
#define SPEEXHIP_PERIOD_CASE(CTV, ONE, PADV) SPEEXHIP_PERIOD_CASE_R(10, CTV, ONE, PADV)
  if (t.ct == 2) {
    if (t.cgroups == 1 && !padded) SPEEXHIP_PERIOD_CASE(2, true, false);
    if (t.cgroups == 1) SPEEXHIP_PERIOD_CASE(2, true, true);
    if (!padded) SPEEXHIP_PERIOD_CASE(2, false, false);
    SPEEXHIP_PERIOD_CASE(2, false, true);
  }
  if (t.cgroups == 1 && !padded) SPEEXHIP_PERIOD_CASE(1, true, false);
  if (t.cgroups == 1) SPEEXHIP_PERIOD_CASE(1, true, true);
  if (!padded) SPEEXHIP_PERIOD_CASE(1, false, false);
  SPEEXHIP_PERIOD_CASE(1, false, true);
#undef SPEEXHIP_PERIOD_CASE
#undef SPEEXHIP_PERIOD_CASE_R
#undef SPEEXHIP_LRC
#undef SPEEXHIP_LRCW
}
}  // namespace

#ifdef SPEEXHIP_STAMPS
extern "C" __attribute__((visibility("default"))) int speexhip_debug_stamps(unsigned long long *dst, size_t n, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long)) != hipSuccess) return 2;
  if (clear) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess) return 3;
    if (hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 16) != hipSuccess) return 4;
  }
  return 0;
}
#endif

// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(period)

}  // namespace speexhip
