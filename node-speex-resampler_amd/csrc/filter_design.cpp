// filter_design.cpp -- see filter_design.h.  Compile with -ffp-contract=off: the widths and
// the order of the float/double operations below reproduce the reference's table bits.
#include "filter_design.h"

#include <climits>
#include <cmath>

#include "../../include/speexhip_resampler.h"

namespace speexhip {
namespace {

// Kaiser windows sampled on [0,1] (+ guard points), resample.c:148-192.  These are the
// algorithm's numeric constants.
struct Window {
  const double *pts;
  int density;  // samples per unit of the normalised abscissa
};

const double kKaiser6[36] = {
    0.99733006, 1.00000000, 0.99733006, 0.98935595, 0.97618418, 0.95799003, 0.93501423, 0.90755855,
    0.87598009, 0.84068475, 0.80211977, 0.76076565, 0.71712752, 0.67172623, 0.62508937, 0.57774224,
    0.53019925, 0.48295561, 0.43647969, 0.39120616, 0.34752997, 0.30580127, 0.26632152, 0.22934058,
    0.19505503, 0.16360756, 0.13508755, 0.10953262, 0.08693120, 0.06722600, 0.05031820, 0.03607231,
    0.02432151, 0.01487334, 0.00752000, 0.00000000};
const double kKaiser8[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200, 0.91223751, 0.87580811,
    0.83439927, 0.78875245, 0.73966538, 0.68797126, 0.63451750, 0.58014482, 0.52566725, 0.47185369,
    0.41941150, 0.36897272, 0.32108304, 0.27619388, 0.23465776, 0.19672670, 0.16255380, 0.13219758,
    0.10562887, 0.08273982, 0.06335451, 0.04724088, 0.03412321, 0.02369490, 0.01563093, 0.00959968,
    0.00527363, 0.00233883, 0.00050000, 0.00000000};
const double kKaiser10[36] = {
    0.99537781, 1.00000000, 0.99537781, 0.98162644, 0.95908712, 0.92831446, 0.89005583, 0.84522401,
    0.79486424, 0.74011713, 0.68217934, 0.62226347, 0.56155915, 0.50119680, 0.44221549, 0.38553619,
    0.33194107, 0.28205962, 0.23636152, 0.19515633, 0.15859932, 0.12670280, 0.09935205, 0.07632451,
    0.05731132, 0.04193980, 0.02979584, 0.02044510, 0.01345224, 0.00839739, 0.00488951, 0.00257636,
    0.00115101, 0.00035515, 0.00000000, 0.00000000};
const double kKaiser12[68] = {
    0.99859849, 1.00000000, 0.99859849, 0.99440475, 0.98745105, 0.97779076, 0.96549770, 0.95066529,
    0.93340547, 0.91384741, 0.89213598, 0.86843014, 0.84290116, 0.81573067, 0.78710866, 0.75723148,
    0.72629970, 0.69451601, 0.66208321, 0.62920216, 0.59606986, 0.56287762, 0.52980938, 0.49704014,
    0.46473455, 0.43304576, 0.40211431, 0.37206735, 0.34301800, 0.31506490, 0.28829195, 0.26276832,
    0.23854851, 0.21567274, 0.19416736, 0.17404546, 0.15530766, 0.13794294, 0.12192957, 0.10723616,
    0.09382272, 0.08164178, 0.07063950, 0.06075685, 0.05193064, 0.04409466, 0.03718069, 0.03111947,
    0.02584161, 0.02127838, 0.01736250, 0.01402878, 0.01121463, 0.00886058, 0.00691064, 0.00531256,
    0.00401805, 0.00298291, 0.00216702, 0.00153438, 0.00105297, 0.00069463, 0.00043489, 0.00025272,
    0.00013031, 0.0000527734, 0.00001000, 0.00000000};

struct Grade {  // resample.c:226-238
  int base_taps, oversample;
  float down_bw, up_bw;
  Window window;
};
const Grade kGrades[11] = {
    {8, 4, 0.830f, 0.860f, {kKaiser6, 32}},     {16, 4, 0.850f, 0.880f, {kKaiser6, 32}},
    {32, 4, 0.882f, 0.910f, {kKaiser6, 32}},    {48, 8, 0.895f, 0.917f, {kKaiser8, 32}},
    {64, 8, 0.921f, 0.940f, {kKaiser8, 32}},    {80, 16, 0.922f, 0.940f, {kKaiser10, 32}},
    {96, 16, 0.940f, 0.945f, {kKaiser10, 32}},  {128, 16, 0.950f, 0.950f, {kKaiser10, 32}},
    {160, 16, 0.960f, 0.960f, {kKaiser10, 32}}, {192, 32, 0.968f, 0.968f, {kKaiser12, 64}},
    {256, 32, 0.975f, 0.975f, {kKaiser12, 64}}};

// resample.c:240-258: cubic interpolation between window samples; the fractional position
// and its powers are float, the weights and the blend are double.
double window_value(float at, const Window &w) {
  const float scaled = at * w.density;
  const int k = static_cast<int>(std::floor(scaled));
  const float u = scaled - k;
  const double w3 = -0.1666666667 * u + 0.1666666667 * (u * u * u);
  const double w2 = u + 0.5 * (u * u) - 0.5 * (u * u * u);
  const double w0 = -0.3333333333 * u + 0.5 * (u * u) - 0.1666666667 * (u * u * u);
  const double w1 = 1.f - w3 - w2 - w0;
  return w0 * w.pts[k] + w1 * w.pts[k + 1] + w2 * w.pts[k + 2] + w3 * w.pts[k + 3];
}

// resample.c:288-298 (FLOATING_POINT): windowed sinc, narrowed to float.
float windowed_sinc(float cutoff, float x, int taps, const Window &w) {
  const float scaled = x * cutoff;
  if (std::fabs(static_cast<double>(x)) < 1e-6) return cutoff;
  if (std::fabs(static_cast<double>(x)) > .5 * taps) return 0;
  return static_cast<float>(cutoff * std::sin(M_PI * scaled) / (M_PI * scaled) *
                            window_value(static_cast<float>(std::fabs(2. * x / taps)), w));
}

uint32_t gcd(uint32_t a, uint32_t b) {
  while (b != 0) {
    const uint32_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// resample.c:593-603
bool mul_ratio(uint32_t *dst, uint32_t v, uint32_t num, uint32_t den) {
  const uint32_t q = v / den, r = v % den;
  if (r > UINT32_MAX / num || q > UINT32_MAX / num || q * num > UINT32_MAX - r * num / den)
    return false;
  *dst = r * num / den + q * num;
  return true;
}

}  // namespace

int design_filter(uint32_t in_rate, uint32_t out_rate, int quality, FilterSpec *f, bool fill_table) {
  if (in_rate == 0 || out_rate == 0) return SPEEXHIP_ERR_INVALID_ARG;
  return design_filter_frac(in_rate, out_rate, in_rate, out_rate, quality, f, fill_table);
}

bool scale_phase(uint32_t *frac, uint32_t new_den, uint32_t old_den) {
  if (!mul_ratio(frac, *frac, new_den, old_den)) return false;
  if (*frac >= new_den) *frac = new_den - 1;  // "safety net", resample.c:1136-1138
  return true;
}

int design_filter_frac(uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate,
                       int quality, FilterSpec *f, bool fill_table) {
  if (ratio_num == 0 || ratio_den == 0 || quality > 10 || quality < 0) return SPEEXHIP_ERR_INVALID_ARG;
  const Grade &g = kGrades[quality];
  const uint32_t common = gcd(ratio_num, ratio_den);
  f->in_rate = in_rate;
  f->out_rate = out_rate;
  f->num = ratio_num / common;
  f->den = ratio_den / common;
  f->quality = quality;
  f->int_advance = static_cast<int>(f->num / f->den);
  f->frac_advance = static_cast<int>(f->num % f->den);
  f->oversample = g.oversample;
  f->taps = g.base_taps;
  if (f->num > f->den) {  // decimation: longer filter, coarser table (resample.c:618-635)
    f->cutoff = g.down_bw * f->den / f->num;
    if (!mul_ratio(&f->taps, f->taps, f->num, f->den)) return SPEEXHIP_ERR_ALLOC_FAILED;
    f->taps = ((f->taps - 1) & (~0x7u)) + 8;
    for (uint32_t k = 2; k <= 16; k *= 2)
      if (k * f->den < f->num) f->oversample >>= 1;
    if (f->oversample < 1) f->oversample = 1;
  } else {
    f->cutoff = g.up_bw;
  }
  // resample.c:647-648 (uint32 wrap-around products, as there)
  const bool direct = static_cast<uint32_t>(f->taps * f->den) <=
                          static_cast<uint32_t>(f->taps * f->oversample + 8) &&
                      INT_MAX / sizeof(float) / f->den >= f->taps;
  uint32_t len;
  if (direct) {
    len = f->taps * f->den;
    f->kind = quality > 8 ? kDirectDouble : kDirectSingle;
  } else {
    if ((INT_MAX / sizeof(float) - 8) / f->oversample < f->taps) return SPEEXHIP_ERR_ALLOC_FAILED;
    len = f->taps * f->oversample + 8;
    f->kind = quality > 8 ? kInterpolateDouble : kInterpolateSingle;
  }
  f->table_len = len;
  f->table.clear();
  if (!fill_table) return SPEEXHIP_ERR_SUCCESS;
  f->table.assign(len, 0.f);
  const int n = static_cast<int>(f->taps);
  if (direct) {  // resample.c:671-678
    for (uint32_t p = 0; p < f->den; p++)
      for (int32_t j = 0; j < n; j++)
        f->table[static_cast<size_t>(p) * n + j] = windowed_sinc(
            f->cutoff, ((j - static_cast<int32_t>(f->taps) / 2 + 1) - (static_cast<float>(p)) / f->den),
            n, g.window);
  } else {  // resample.c:690-691
    const int32_t last = static_cast<int32_t>(f->oversample * f->taps + 4);
    for (int32_t i = -4; i < last; i++)
      f->table[i + 4] =
          windowed_sinc(f->cutoff, (i / static_cast<float>(f->oversample) - f->taps / 2), n, g.window);
  }
  return SPEEXHIP_ERR_SUCCESS;
}

void phase_blend(const FilterSpec &f, uint32_t phase, int *offset, float w[4]) {
  // (a folded view, FilterSpec::fold: its phases and its den are `fold` times the filter's -- the reference's
  //  arithmetic below runs on the filter's own)
  const uint32_t den = f.den / f.fold;
  phase /= f.fold;
  *offset = static_cast<int>(phase * f.oversample / den);
  const float t = (static_cast<float>((phase * f.oversample) % den)) / den;
  w[0] = -0.16667f * t + 0.16667f * t * t * t;
  w[1] = t + 0.5f * t * t - 0.5f * t * t * t;
  w[3] = -0.33333f * t + 0.5f * t * t - 0.16667f * t * t * t;
  w[2] = 1. - w[0] - w[1] - w[3];
}

void phase_taps(const FilterSpec &f, uint32_t phase, double *row) {
  const int n = static_cast<int>(f.taps);
  if (f.direct()) {
    for (int j = 0; j < n; j++) row[j] = f.table[static_cast<size_t>(phase / f.fold) * n + j];
    return;
  }
  int offset;
  float w[4];
  phase_blend(f, phase, &offset, w);
  for (int j = 0; j < n; j++) {
    const float *t = f.table.data() + 4 + (j + 1) * f.oversample - offset - 2;
    row[j] = static_cast<double>(w[0]) * t[0] + static_cast<double>(w[1]) * t[1] +
             static_cast<double>(w[2]) * t[2] + static_cast<double>(w[3]) * t[3];
  }
}

}  // namespace speexhip
