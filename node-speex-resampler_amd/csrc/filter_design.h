// filter_design.h -- host-side filter design for the resampler (product code).
//
// Builds, in double-precision libm on the host, exactly the tables the reference builds in
// update_filter() (reference deps/speex/resample.c:605-702) and, for the fast GPU path, the
// per-phase tap rows derived from them.  GPU transcendental functions are never used: the
// table bits must match the reference's ("Exactness of the table").
#pragma once
#include <cstdint>
#include <vector>

namespace speexhip {

enum KernelKind : int {
  kDirectSingle = 0,       // resample.c:331  fp32 running sum
  kDirectDouble = 1,       // resample.c:389  4 fp64 partial sums of fp32 products
  kInterpolateSingle = 2,  // resample.c:438  4 fp32 sums + cubic blend
  kInterpolateDouble = 3,  // resample.c:501  4 fp64 sums + cubic blend
};

struct FilterSpec {
  uint32_t in_rate = 0, out_rate = 0;
  uint32_t num = 0, den = 0;  // in/out reduced by their gcd (resample.c:1125-1128)
  int quality = 0;
  uint32_t taps = 0;          // filt_len
  uint32_t oversample = 0;
  int int_advance = 0, frac_advance = 0;  // resample.c:613-614
  float cutoff = 0.f;
  KernelKind kind = kDirectSingle;
  uint32_t table_len = 0;     // floats in the reference-layout table
  std::vector<float> table;   // reference layout: den*taps (direct) or taps*oversample+8
  bool direct() const { return kind == kDirectSingle || kind == kDirectDouble; }
  // Round 6 -- the PERIOD KERNEL'S VIEW of a ratio with a small denominator (kernels.h, period_view): num and den are
  // `fold` times the filter's own.  A resampler's outputs repeat with period den; they repeat with period fold * den just
  // as well -- output K has phase (K * num) mod den and window start (K * num) div den either way -- and the period
  // kernel wants den >= 7 phases to fill its register tile.  So 7:6 runs as 35:30, 11:1 as 110:10: the same taps per
  // output, in groups of five phases.  Phases of a folded spec are `fold` times the table's: phase_taps / phase_blend
  // divide them back.  1 everywhere else.
  uint32_t fold = 1;
};

// Returns a SPEEXHIP_ERR_* code.  `fill_table=false` computes the geometry only.
int design_filter(uint32_t in_rate, uint32_t out_rate, int quality, FilterSpec *spec,
                  bool fill_table = true);

// Same with the ratio given separately from the nominal rates (speex_resampler_init_frac /
// set_rate_frac, resample.c:799, 1107): num/den = ratio reduced by its gcd, the rates are
// only reported back.
int design_filter_frac(uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate,
                       int quality, FilterSpec *spec, bool fill_table = true);

// Phase numerator carried over to a new denominator (resample.c:1130-1139): frac*new/old
// without 32-bit overflow, clamped below new_den.  false = overflow (RESAMPLER_ERR_OVERFLOW).
bool scale_phase(uint32_t *frac, uint32_t new_den, uint32_t old_den);

// Cubic blend weights of the interpolated kernels for one output phase
// (resample.c:454-458 + cubic_coef :318-328), bit-exact float arithmetic.
void phase_blend(const FilterSpec &f, uint32_t phase, int *offset, float w[4]);

// Effective FIR taps of output phase `phase` as one row of `taps` doubles:
//   direct kinds:       the table row itself
//   interpolated kinds: sum_t w[t] * table[4 + (j+1)*oversample - offset - 2 + t]
// (the algebraic collapse of the reference's four accumulators; fast path only).
void phase_taps(const FilterSpec &f, uint32_t phase, double *row);

}  // namespace speexhip
