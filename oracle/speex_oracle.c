/*
 * oracle/speex_oracle.c -- CPU restatement of the Speex resampler hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under node-speex-resampler_amd/ (the
 * product) may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the reported CPU baseline.
 *
 * What it restates (all citations relative to /root/reference/):
 *   - filter design         deps/speex/resample.c:148-298, 605-702
 *   - the four FIR kernels  deps/speex/resample.c:331-558
 *   - stream bookkeeping    deps/speex/resample.c:878-902, 968-1036, 1061-1082 (int16 entry)
 *                           and 927-963, 1038-1059 (float entry)
 *   - mid-stream changes    deps/speex/resample.c:703-782 (history re-alignment, "magic"
 *                           samples), 904-922, 1084-1220 (set_rate[_frac], set_quality,
 *                           latencies, skip_zeros, reset_mem)
 *   - float build typedefs  deps/speex/arch.h:131-209 (FLOATING_POINT)
 * as built by scripts/build_emscripten.sh:18-19 (-D FLOATING_POINT -D OUTSIDE_SPEEX).
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file bit-for-bit
 * against tests/golden/ vectors that were produced by the reference itself
 * (native build oracle/_ref and the shipped WASM, which agree), see
 * tests/golden/make_golden.py.
 *
 * Build with -O2 -ffp-contract=off (no FMA contraction, no fast-math): the
 * summation order and the float/double widths below are part of the contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_ALLOC 1
#define ORC_ERR_INVALID 3
#define ORC_ERR_OVERFLOW 5

#define ORC_BLOCK_IN 160   /* st->buffer_size, resample.c:835 */
#define ORC_BLOCK_OUT 1024 /* FIXED_STACK_ALLOC without VAR_ARRAYS, resample.c:108-112 */

enum { K_DIRECT_SINGLE = 0, K_DIRECT_DOUBLE = 1, K_INTERP_SINGLE = 2, K_INTERP_DOUBLE = 3 };

typedef struct {
  const double *samples; /* window sampled on [0,1], 4 guard points */
  int steps;             /* "oversample" of the window table */
} orc_window;

/* Kaiser window tables: numeric constants of the algorithm (resample.c:148-192). */
static const double W12[68] = {
    0.99859849, 1.00000000, 0.99859849, 0.99440475, 0.98745105, 0.97779076, 0.96549770,
    0.95066529, 0.93340547, 0.91384741, 0.89213598, 0.86843014, 0.84290116, 0.81573067,
    0.78710866, 0.75723148, 0.72629970, 0.69451601, 0.66208321, 0.62920216, 0.59606986,
    0.56287762, 0.52980938, 0.49704014, 0.46473455, 0.43304576, 0.40211431, 0.37206735,
    0.34301800, 0.31506490, 0.28829195, 0.26276832, 0.23854851, 0.21567274, 0.19416736,
    0.17404546, 0.15530766, 0.13794294, 0.12192957, 0.10723616, 0.09382272, 0.08164178,
    0.07063950, 0.06075685, 0.05193064, 0.04409466, 0.03718069, 0.03111947, 0.02584161,
    0.02127838, 0.01736250, 0.01402878, 0.01121463, 0.00886058, 0.00691064, 0.00531256,
    0.00401805, 0.00298291, 0.00216702, 0.00153438, 0.00105297, 0.00069463, 0.00043489,
    0.00025272, 0.00013031, 0.0000527734, 0.00001000, 0.00000000};
static const double W10[36] = {
    0.99537781, 1.00000000, 0.99537781, 0.98162644, 0.95908712, 0.92831446, 0.89005583,
    0.84522401, 0.79486424, 0.74011713, 0.68217934, 0.62226347, 0.56155915, 0.50119680,
    0.44221549, 0.38553619, 0.33194107, 0.28205962, 0.23636152, 0.19515633, 0.15859932,
    0.12670280, 0.09935205, 0.07632451, 0.05731132, 0.04193980, 0.02979584, 0.02044510,
    0.01345224, 0.00839739, 0.00488951, 0.00257636, 0.00115101, 0.00035515, 0.00000000,
    0.00000000};
static const double W8[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200, 0.91223751,
    0.87580811, 0.83439927, 0.78875245, 0.73966538, 0.68797126, 0.63451750, 0.58014482,
    0.52566725, 0.47185369, 0.41941150, 0.36897272, 0.32108304, 0.27619388, 0.23465776,
    0.19672670, 0.16255380, 0.13219758, 0.10562887, 0.08273982, 0.06335451, 0.04724088,
    0.03412321, 0.02369490, 0.01563093, 0.00959968, 0.00527363, 0.00233883, 0.00050000,
    0.00000000};
static const double W6[36] = {
    0.99733006, 1.00000000, 0.99733006, 0.98935595, 0.97618418, 0.95799003, 0.93501423,
    0.90755855, 0.87598009, 0.84068475, 0.80211977, 0.76076565, 0.71712752, 0.67172623,
    0.62508937, 0.57774224, 0.53019925, 0.48295561, 0.43647969, 0.39120616, 0.34752997,
    0.30580127, 0.26632152, 0.22934058, 0.19505503, 0.16360756, 0.13508755, 0.10953262,
    0.08693120, 0.06722600, 0.05031820, 0.03607231, 0.02432151, 0.01487334, 0.00752000,
    0.00000000};

static const orc_window WIN6 = {W6, 32}, WIN8 = {W8, 32}, WIN10 = {W10, 32}, WIN12 = {W12, 64};

/* quality -> (taps, table oversampling, bandwidths, window): resample.c:226-238 */
typedef struct {
  int taps, os;
  float bw_down, bw_up;
  const orc_window *win;
} orc_quality;
static const orc_quality QUAL[11] = {
    {8, 4, 0.830f, 0.860f, &WIN6},     {16, 4, 0.850f, 0.880f, &WIN6},
    {32, 4, 0.882f, 0.910f, &WIN6},    {48, 8, 0.895f, 0.917f, &WIN8},
    {64, 8, 0.921f, 0.940f, &WIN8},    {80, 16, 0.922f, 0.940f, &WIN10},
    {96, 16, 0.940f, 0.945f, &WIN10},  {128, 16, 0.950f, 0.950f, &WIN10},
    {160, 16, 0.960f, 0.960f, &WIN10}, {192, 32, 0.968f, 0.968f, &WIN12},
    {256, 32, 0.975f, 0.975f, &WIN12}};

typedef struct orc_state {
  uint32_t in_rate, out_rate, num, den, channels;
  int quality;
  uint32_t taps, os;
  int step_int, step_frac;
  float cutoff;
  int kind;
  uint32_t table_len, table_cap; /* floats in use / allocated */
  float *table;
  uint32_t line;     /* floats of history+staging per channel, grow-only ("mem_alloc_size"):
                        at least taps-1+ORC_BLOCK_IN */
  float *lines;      /* channels x line */
  int32_t *pos;      /* per channel: resample.c "last_sample" */
  uint32_t *phase;   /* per channel: resample.c "samp_frac_num" */
  uint32_t *pending; /* per channel: resample.c "magic_samples": already-buffered input frames
                        (right after the history) left over from a filter-length change */
  int started;       /* a block has been processed: filter changes must re-align the history */
  int live;          /* construction finished (resample.c "initialised") */
  int zero;          /* the last filter change failed: resampler_basic_zero is installed
                        (resample.c:561-591, 785-791) */
  uint32_t in_stride, out_stride; /* per-channel entry points, resample.c:842-843, 1170-1188 */
} orc_state;

/* Window value at x in [0,1] by 4-point cubic interpolation of the table
 * (resample.c:240-258).  `t` and its powers are float, the blend is double. */
static double window_at(float x, const orc_window *w) {
  float scaled = x * w->steps;
  int cell = (int)floor(scaled);
  float t = scaled - cell;
  double c3 = -0.1666666667 * t + 0.1666666667 * (t * t * t);
  double c2 = t + 0.5 * (t * t) - 0.5 * (t * t * t);
  double c0 = -0.3333333333 * t + 0.5 * (t * t) - 0.1666666667 * (t * t * t);
  double c1 = 1.f - c3 - c2 - c0;
  return c0 * w->samples[cell] + c1 * w->samples[cell + 1] + c2 * w->samples[cell + 2] +
         c3 * w->samples[cell + 3];
}

/* One tap of the windowed sinc (resample.c:288-298, float build). */
static float tap_value(float cutoff, float x, int taps, const orc_window *w) {
  float xc = x * cutoff;
  if (fabs(x) < 1e-6) return cutoff;
  if (fabs(x) > .5 * taps) return 0;
  return cutoff * sin(M_PI * xc) / (M_PI * xc) * window_at(fabs(2. * x / taps), w);
}

/* value*num/den without 32-bit overflow, or failure (resample.c:593-603). */
static int scale_u32(uint32_t *res, uint32_t value, uint32_t num, uint32_t den) {
  uint32_t whole = value / den, rest = value % den;
  if (rest > UINT32_MAX / num || whole > UINT32_MAX / num ||
      whole * num > UINT32_MAX - rest * num / den)
    return 1;
  *res = rest * num / den + whole * num;
  return 0;
}

static uint32_t gcd_u32(uint32_t a, uint32_t b) {
  while (b) {
    uint32_t r = a % b;
    a = b;
    b = r;
  }
  return a;
}

/* History re-alignment after the filter length changed from old_taps (resample.c:703-782).
 * A channel's line holds  history (taps-1 frames) ++ pending frames.  Restated per case:
 *   not started : everything is silence;
 *   longer      : put `pending` zeros in front (the reference's "remove the magic samples as
 *                 if nothing had happened"), then either left-pad with zeros up to the new
 *                 history length and move the position by half the padding, or -- if that
 *                 augmented line is already long enough -- drop its first q frames and keep
 *                 its last q frames as pending input, q = half the excess;
 *   shorter     : drop the first d frames, the last d (plus the old pending ones) become
 *                 pending input, d = half the difference. */
static int fit_lines(orc_state *o, uint32_t old_taps) {
  const uint32_t need = o->taps - 1 + ORC_BLOCK_IN;
  if (need > o->line) {
    float *fresh = (float *)calloc((size_t)o->channels * need, sizeof(float));
    if (!fresh) return ORC_ERR_ALLOC;
    for (uint32_t c = 0; c < o->channels && o->lines; c++)
      memcpy(fresh + (size_t)c * need, o->lines + (size_t)c * o->line, sizeof(float) * o->line);
    free(o->lines);
    o->lines = fresh;
    o->line = need;
  }
  if (!o->started) {
    memset(o->lines, 0, sizeof(float) * (size_t)o->channels * o->line);
    return ORC_OK;
  }
  if (o->taps == old_taps) return ORC_OK;
  for (uint32_t c = 0; c < o->channels; c++) {
    float *x = o->lines + (size_t)c * o->line;
    const uint32_t p = o->pending[c];
    if (o->taps > old_taps) {
      const uint32_t have = old_taps - 1 + p; /* frames held */
      const uint32_t aug = old_taps + 2 * p;  /* "olen", :741 */
      float *b = (float *)calloc((size_t)aug + o->taps, sizeof(float));
      if (!b) return ORC_ERR_ALLOC;
      memcpy(b + p, x, sizeof(float) * have); /* b = p zeros ++ held frames: aug-1 frames */
      o->pending[c] = 0;
      if (o->taps > aug) { /* :748-758 */
        const uint32_t lead = o->taps - aug;
        memset(x, 0, sizeof(float) * lead);
        memcpy(x + lead, b, sizeof(float) * (aug - 1));
        o->pos[c] += lead / 2;
      } else { /* :759-764 */
        const uint32_t q = (aug - o->taps) / 2;
        memcpy(x, b + q, sizeof(float) * (o->taps - 1 + q));
        o->pending[c] = q;
      }
      free(b);
    } else { /* :766-782 */
      const uint32_t d = (old_taps - o->taps) / 2;
      memmove(x, x + d, sizeof(float) * (o->taps - 1 + d + p));
      o->pending[c] = d + p;
    }
  }
  return ORC_OK;
}

/* Filter design (resample.c:605-702) followed by the history re-alignment (:703-782). */
static int design(orc_state *o) {
  const orc_quality *q = &QUAL[o->quality];
  const uint32_t old_taps = o->taps;
  o->step_int = o->num / o->den;
  o->step_frac = o->num % o->den;
  o->os = q->os;
  o->taps = q->taps;
  if (o->num > o->den) { /* decimating: stretch the filter, :618-635 */
    o->cutoff = q->bw_down * o->den / o->num;
    if (scale_u32(&o->taps, o->taps, o->num, o->den)) goto fail;
    o->taps = ((o->taps - 1) & (~0x7u)) + 8;
    if (2 * o->den < o->num) o->os >>= 1;
    if (4 * o->den < o->num) o->os >>= 1;
    if (8 * o->den < o->num) o->os >>= 1;
    if (16 * o->den < o->num) o->os >>= 1;
    if (o->os < 1) o->os = 1;
  } else {
    o->cutoff = q->bw_up;
  }
  /* smaller table wins, :647-648 (uint32 wrap-around arithmetic as in the reference) */
  int direct = (uint32_t)(o->taps * o->den) <= (uint32_t)(o->taps * o->os + 8) &&
               INT32_MAX / sizeof(float) / o->den >= o->taps;
  uint32_t want;
  if (direct) {
    want = o->taps * o->den;
  } else {
    if ((INT32_MAX / sizeof(float) - 8) / o->os < o->taps) goto fail;
    want = o->taps * o->os + 8;
  }
  if (want > o->table_cap) { /* grow-only allocation, :659-667 */
    float *t = (float *)realloc(o->table, sizeof(float) * want);
    if (!t) goto fail;
    o->table = t;
    o->table_cap = want;
  }
  o->table_len = want;
  if (direct) { /* one row of taps per output phase, :671-678 */
    for (uint32_t ph = 0; ph < o->den; ph++)
      for (int32_t j = 0; j < (int32_t)o->taps; j++)
        o->table[ph * o->taps + j] = tap_value(
            o->cutoff, ((j - (int32_t)o->taps / 2 + 1) - ((float)ph) / o->den), o->taps, q->win);
    o->kind = o->quality > 8 ? K_DIRECT_DOUBLE : K_DIRECT_SINGLE;
  } else { /* one oversampled prototype with 4 guard taps each side, :690-691 */
    for (int32_t i = -4; i < (int32_t)(o->os * o->taps + 4); i++)
      o->table[i + 4] =
          tap_value(o->cutoff, (i / (float)o->os - o->taps / 2), o->taps, q->win);
    o->kind = o->quality > 8 ? K_INTERP_DOUBLE : K_INTERP_SINGLE;
  }
  o->zero = 0; /* a real kernel is installed at :682-698, before the history is fitted */
  {
    const int rc = fit_lines(o, old_taps);
    if (rc != ORC_OK) {
      o->zero = 1;
      o->taps = old_taps;
    }
    return rc;
  }
fail:
  o->zero = 1;        /* :785 resampler_basic_zero */
  o->taps = old_taps; /* :786-790: the history still belongs to the old filter length */
  return ORC_ERR_ALLOC;
}

void orc_free(orc_state *o) {
  if (!o) return;
  free(o->table);
  free(o->lines);
  free(o->pos);
  free(o->phase);
  free(o->pending);
  free(o);
}

/* resample.c:1107-1145: new ratio (reduced by the gcd), phase numerators rescaled to the
 * new denominator, then the filter is redesigned if the state is live. */
int orc_set_rate_frac(orc_state *o, uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate,
                      uint32_t out_rate) {
  if (ratio_num == 0 || ratio_den == 0) return ORC_ERR_INVALID;
  if (o->in_rate == in_rate && o->out_rate == out_rate && o->num == ratio_num &&
      o->den == ratio_den)
    return ORC_OK;
  const uint32_t old_den = o->den;
  const uint32_t g = gcd_u32(ratio_num, ratio_den);
  o->in_rate = in_rate;
  o->out_rate = out_rate;
  o->num = ratio_num / g;
  o->den = ratio_den / g;
  if (old_den > 0) {
    for (uint32_t c = 0; c < o->channels; c++) {
      if (scale_u32(&o->phase[c], o->phase[c], o->den, old_den)) return ORC_ERR_OVERFLOW;
      if (o->phase[c] >= o->den) o->phase[c] = o->den - 1;
    }
  }
  return o->live ? design(o) : ORC_OK;
}

/* resample.c:1084-1087 */
int orc_set_rate(orc_state *o, uint32_t in_rate, uint32_t out_rate) {
  return orc_set_rate_frac(o, in_rate, out_rate, in_rate, out_rate);
}

/* resample.c:1153-1163 */
int orc_set_quality(orc_state *o, int quality) {
  if (quality > 10 || quality < 0) return ORC_ERR_INVALID;
  if (o->quality == quality) return ORC_OK;
  o->quality = quality;
  return o->live ? design(o) : ORC_OK;
}

/* resample.c:799-866 */
orc_state *orc_new_frac(uint32_t channels, uint32_t ratio_num, uint32_t ratio_den,
                        uint32_t in_rate, uint32_t out_rate, int quality, int *err) {
  int e = ORC_OK;
  orc_state *o = NULL;
  if (channels == 0 || ratio_num == 0 || ratio_den == 0 || quality > 10 || quality < 0) {
    e = ORC_ERR_INVALID;
  } else if (!(o = (orc_state *)calloc(1, sizeof(*o)))) {
    e = ORC_ERR_ALLOC;
  } else {
    o->channels = channels;
    o->in_stride = o->out_stride = 1; /* :842-843 */
    o->quality = -1;
    o->cutoff = 1.f;
    o->pos = (int32_t *)calloc(channels, sizeof(int32_t));
    o->phase = (uint32_t *)calloc(channels, sizeof(uint32_t));
    o->pending = (uint32_t *)calloc(channels, sizeof(uint32_t));
    if (!(o->pos && o->phase && o->pending)) {
      e = ORC_ERR_ALLOC;
    } else {
      orc_set_quality(o, quality);
      orc_set_rate_frac(o, ratio_num, ratio_den, in_rate, out_rate);
      e = design(o);
      o->live = 1;
    }
    if (e != ORC_OK) {
      orc_free(o);
      o = NULL;
    }
  }
  if (err) *err = e;
  return o;
}

/* resample.c:794-797 */
orc_state *orc_new(uint32_t channels, uint32_t in_rate, uint32_t out_rate, int quality,
                   int *err) {
  return orc_new_frac(channels, in_rate, out_rate, in_rate, out_rate, quality, err);
}

/* float -> int16 with round-half-up in double and saturation (arch.h:208-209). */
static int16_t to_pcm(float v) {
  if (v < -32767.5f) return -32768;
  if (v > 32766.5f) return 32767;
  return (int16_t)floor(.5 + v);
}

/* Cubic blend weights for the interpolated table (resample.c:318-328):
 * three in float, the third (index 2) via a double expression. */
static void blend_weights(float f, float w[4]) {
  w[0] = -0.16667f * f + 0.16667f * f * f * f;
  w[1] = f + 0.5f * f * f - 0.5f * f * f * f;
  w[3] = -0.33333f * f + 0.5f * f * f - 0.16667f * f * f * f;
  w[2] = 1. - w[0] - w[1] - w[3];
}

/* One output sample at window start x[0..taps) for the current phase. */
static float fir_sample(const orc_state *o, const float *x, uint32_t phase) {
  const int n = (int)o->taps;
  switch (o->kind) {
    case K_DIRECT_SINGLE: { /* resample.c:346-352: float products, float running sum */
      const float *h = o->table + (size_t)phase * n;
      float s = 0;
      for (int j = 0; j < n; j++) s += h[j] * x[j];
      return s;
    }
    case K_DIRECT_DOUBLE: { /* :409-417,422: float products into 4 double lanes */
      const float *h = o->table + (size_t)phase * n;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (int j = 0; j < n; j += 4) {
        a0 += h[j] * x[j];
        a1 += h[j + 1] * x[j + 1];
        a2 += h[j + 2] * x[j + 2];
        a3 += h[j + 3] * x[j + 3];
      }
      double s = a0 + a1 + a2 + a3;
      return (float)s;
    }
    case K_INTERP_SINGLE: { /* :454-476 */
      const int shift = phase * o->os / o->den;
      const float f = ((float)((phase * o->os) % o->den)) / o->den;
      float a0 = 0, a1 = 0, a2 = 0, a3 = 0, w[4];
      for (int j = 0; j < n; j++) {
        const float v = x[j];
        const float *t = o->table + 4 + (j + 1) * o->os - shift;
        a0 += v * t[-2];
        a1 += v * t[-1];
        a2 += v * t[0];
        a3 += v * t[1];
      }
      blend_weights(f, w);
      return w[0] * a0 + w[1] * a1 + w[2] * a2 + w[3] * a3;
    }
    default: { /* K_INTERP_DOUBLE, :517-539,545: float products, double lanes,
                  double blend narrowed to float */
      const int shift = phase * o->os / o->den;
      const float f = ((float)((phase * o->os) % o->den)) / o->den;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      float w[4];
      for (int j = 0; j < n; j++) {
        const float v = x[j];
        const float *t = o->table + 4 + (j + 1) * o->os - shift;
        a0 += v * t[-2];
        a1 += v * t[-1];
        a2 += v * t[0];
        a3 += v * t[1];
      }
      blend_weights(f, w);
      float s = w[0] * a0 + w[1] * a1 + w[2] * a2 + w[3] * a3;
      return s;
    }
  }
}

/* One run of the FIR over a channel line holding history ++ nin input frames, at most `cap`
 * outputs, followed by the history shift (resample.c:878-902 around the kernel loop shape at
 * :344-379).  Outputs go to `out` with `stride`: rounded to int16 (arch.h:208-209, as
 * resample.c:1022 does from its stack buffer) or as they are (float entry).  Returns the
 * number made; *used = input frames that entered the history. */
static uint32_t run_block(orc_state *o, uint32_t c, uint32_t nin, uint32_t cap, void *out,
                          uint32_t stride, int as_pcm, uint32_t *used_out) {
  float *x = o->lines + (size_t)c * o->line;
  int32_t pos = o->pos[c];
  uint32_t phase = o->phase[c];
  uint32_t made = 0;
  o->started = 1;
  while (!(pos >= (int32_t)nin || made >= cap)) {
    const float v = o->zero ? 0.f : fir_sample(o, x + pos, phase); /* resample.c:561-591 */
    if (as_pcm)
      ((int16_t *)out)[(size_t)made * stride] = to_pcm(v);
    else
      ((float *)out)[(size_t)made * stride] = v;
    made++;
    pos += o->step_int;
    phase += o->step_frac;
    if (phase >= o->den) {
      phase -= o->den;
      pos++;
    }
  }
  uint32_t used = nin;
  if (pos < (int32_t)nin) used = pos; /* output-bound: only `pos` frames count */
  o->pos[c] = pos - (int32_t)used;
  o->phase[c] = phase;
  for (uint32_t j = 0; j + 1 < o->taps; j++) x[j] = x[j + used];
  *used_out = used;
  return made;
}

/* The pending ("magic") frames are input that is already in the line: run them as a block of
 * their own and keep what was not consumed (resample.c:904-922). */
static uint32_t drain_pending(orc_state *o, uint32_t c, uint32_t cap, void *out, uint32_t stride,
                              int as_pcm) {
  float *x = o->lines + (size_t)c * o->line;
  uint32_t used = 0;
  const uint32_t made = run_block(o, c, o->pending[c], cap, out, stride, as_pcm, &used);
  o->pending[c] -= used;
  for (uint32_t i = 0; i < o->pending[c]; i++) x[o->taps - 1 + i] = x[o->taps - 1 + i + used];
  return made;
}

/* One channel of one call through the int16 entry point: blocks of at most (line - history)
 * input frames -- 160 unless the filter has been shortened since -- and at most 1024 outputs,
 * pending frames first (resample.c:968-1036).  Strided int16 in/out. */
static void run_channel(orc_state *o, uint32_t c, const int16_t *in, uint32_t in_stride, uint32_t stride,
                        uint32_t *in_len, int16_t *out, uint32_t *out_len) {
  float *x = o->lines + (size_t)c * o->line;
  const uint32_t hist = o->taps - 1;
  const uint32_t block_in = o->line - hist;
  uint32_t in_left = *in_len, out_left = *out_len;
  while (in_left && out_left) {
    uint32_t room = out_left > ORC_BLOCK_OUT ? ORC_BLOCK_OUT : out_left;
    uint32_t from_pending = 0, used = 0, made = 0;
    if (o->pending[c]) {
      from_pending = drain_pending(o, c, room, out, stride, 1);
      room -= from_pending;
      out_left -= from_pending;
    }
    if (!o->pending[c]) {
      const uint32_t nin = in_left > block_in ? block_in : in_left;
      for (uint32_t j = 0; j < nin; j++) x[hist + j] = in ? (float)in[(size_t)j * in_stride] : 0.f;
      made = run_block(o, c, nin, room, out + (size_t)from_pending * stride, stride, 1, &used);
    }
    in_left -= used;
    out_left -= made;
    out += (size_t)(from_pending + made) * stride;
    if (in) in += (size_t)used * in_stride;
  }
  *in_len -= in_left;
  *out_len -= out_left;
}

/* resample.c:1061-1082: channels outermost, lengths restored before each. */
int orc_process_interleaved_int(orc_state *o, const int16_t *in, uint32_t *in_len,
                                int16_t *out, uint32_t *out_len) {
  const uint32_t want_in = *in_len, want_out = *out_len;
  for (uint32_t c = 0; c < o->channels; c++) {
    *in_len = want_in;
    *out_len = want_out;
    run_channel(o, c, in ? in + c : NULL, o->channels, o->channels, in_len, out + c, out_len);
  }
  return o->zero ? ORC_ERR_ALLOC : ORC_OK; /* :1081 */
}

/* resample.c:968-1036 as a public call: one channel, the state's strides (:1170-1188) */
int orc_process_int(orc_state *o, uint32_t c, const int16_t *in, uint32_t *in_len, int16_t *out,
                    uint32_t *out_len) {
  run_channel(o, c, in, o->in_stride, o->out_stride, in_len, out, out_len);
  return o->zero ? ORC_ERR_ALLOC : ORC_OK; /* :1035 */
}

/* One channel of one call through the FLOAT entry point (resample.c:927-963): input frames are
 * copied as they are, the FIR values are written unrounded, and -- unlike the int16 entry
 * point -- a block's output is limited only by the room left (resample.c:943), not by 1024;
 * pending frames are drained once, up front, even when the call brings no input. */
static void run_channel_float(orc_state *o, uint32_t c, const float *in, uint32_t in_stride, uint32_t stride,
                              uint32_t *in_len, float *out, uint32_t *out_len) {
  float *x = o->lines + (size_t)c * o->line;
  const uint32_t hist = o->taps - 1;
  const uint32_t block_in = o->line - hist;
  uint32_t in_left = *in_len, out_left = *out_len;
  if (o->pending[c]) {
    const uint32_t m = drain_pending(o, c, out_left, out, stride, 0);
    out_left -= m;
    out += (size_t)m * stride;
  }
  if (!o->pending[c]) {
    while (in_left && out_left) {
      const uint32_t nin = in_left > block_in ? block_in : in_left;
      uint32_t used = 0;
      for (uint32_t j = 0; j < nin; j++) x[hist + j] = in ? in[(size_t)j * in_stride] : 0.f;
      const uint32_t made = run_block(o, c, nin, out_left, out, stride, 0, &used);
      in_left -= used;
      out_left -= made;
      out += (size_t)made * stride;
      if (in) in += (size_t)used * in_stride;
    }
  }
  *in_len -= in_left;
  *out_len -= out_left;
}

/* resample.c:1038-1059 */
int orc_process_interleaved_float(orc_state *o, const float *in, uint32_t *in_len, float *out,
                                  uint32_t *out_len) {
  const uint32_t want_in = *in_len, want_out = *out_len;
  for (uint32_t c = 0; c < o->channels; c++) {
    *in_len = want_in;
    *out_len = want_out;
    run_channel_float(o, c, in ? in + c : NULL, o->channels, o->channels, in_len, out + c, out_len);
  }
  return o->zero ? ORC_ERR_ALLOC : ORC_OK; /* :1058 */
}

/* resample.c:927-963 as a public call */
int orc_process_float(orc_state *o, uint32_t c, const float *in, uint32_t *in_len, float *out,
                      uint32_t *out_len) {
  run_channel_float(o, c, in, o->in_stride, o->out_stride, in_len, out, out_len);
  return o->zero ? ORC_ERR_ALLOC : ORC_OK; /* :962 */
}

/* resample.c:1170-1188 */
void orc_set_input_stride(orc_state *o, uint32_t stride) { o->in_stride = stride; }
void orc_set_output_stride(orc_state *o, uint32_t stride) { o->out_stride = stride; }
uint32_t orc_get_input_stride(const orc_state *o) { return o->in_stride; }
uint32_t orc_get_output_stride(const orc_state *o) { return o->out_stride; }
/* one channel's position: last_sample, samp_frac_num, magic_samples */
void orc_channel_position(const orc_state *o, uint32_t c, int32_t *pos, uint32_t *phase, uint32_t *pending) {
  *pos = o->pos[c];
  *phase = o->phase[c];
  *pending = o->pending[c];
}

/* resample.c:1089-1093, 1147-1151, 1165-1168 */
void orc_get_rate(const orc_state *o, uint32_t *in_rate, uint32_t *out_rate) {
  *in_rate = o->in_rate;
  *out_rate = o->out_rate;
}
void orc_get_ratio(const orc_state *o, uint32_t *num, uint32_t *den) {
  *num = o->num;
  *den = o->den;
}
int orc_get_quality(const orc_state *o) { return o->quality; }

/* resample.c:1190-1198 */
int orc_input_latency(const orc_state *o) { return o->taps / 2; }
int orc_output_latency(const orc_state *o) {
  return ((o->taps / 2) * o->den + (o->num >> 1)) / o->num;
}

/* resample.c:1200-1206: start half a filter in, so the first output is not the filter's ramp-up */
int orc_skip_zeros(orc_state *o) {
  for (uint32_t c = 0; c < o->channels; c++) o->pos[c] = o->taps / 2;
  return ORC_OK;
}

/* resample.c:1208-1220 */
int orc_reset_mem(orc_state *o) {
  for (uint32_t c = 0; c < o->channels; c++) {
    o->pos[c] = 0;
    o->pending[c] = 0;
    o->phase[c] = 0;
  }
  /* the reference clears the first channels*(taps-1) floats of its buffer as one run; with a
     per-channel stride that is only every channel's history when there is one channel.  The
     frames it leaves behind in the other channels' histories are a quirk we restate. */
  {
    const size_t n = (size_t)o->channels * (o->taps - 1);
    memset(o->lines, 0, sizeof(float) * n);
  }
  return ORC_OK;
}

/* ---- introspection for the tests (no reference counterpart) ---- */
void orc_info(const orc_state *o, uint32_t out[8]) {
  out[0] = o->num;
  out[1] = o->den;
  out[2] = o->taps;
  out[3] = o->os;
  out[4] = (uint32_t)o->kind;
  out[5] = o->table_len;
  out[6] = (uint32_t)o->step_int;
  out[7] = (uint32_t)o->step_frac;
}
const float *orc_table(const orc_state *o) { return o->table; }
void orc_position(const orc_state *o, int32_t *pos, uint32_t *phase) {
  *pos = o->pos[0];
  *phase = o->phase[0];
}
/* Last taps-1 consumed frames of channel c, as the float values the next call sees. */
void orc_history(const orc_state *o, uint32_t c, float *dst) {
  memcpy(dst, o->lines + (size_t)c * o->line, sizeof(float) * (o->taps - 1));
}
/* Pending ("magic") frame count of channel 0 and, if dst, channel c's pending frames. */
uint32_t orc_pending(const orc_state *o, uint32_t c, float *dst) {
  if (dst)
    memcpy(dst, o->lines + (size_t)c * o->line + (o->taps - 1), sizeof(float) * o->pending[c]);
  return o->pending[c];
}
uint32_t orc_block_in(const orc_state *o) { return o->line - (o->taps - 1); }

/* resample.c:1222-1239 */
const char *orc_strerror(int err) {
  switch (err) {
    case 0: return "Success.";
    case 1: return "Memory allocation failed.";
    case 2: return "Bad resampler state.";
    case 3: return "Invalid argument.";
    case 4: return "Input and output buffers overlap.";
    default: return "Unknown error. Bad error code or strange version mismatch.";
  }
}
