// kernels_period64_w16.hip -- the fp64-accumulate period kernel over an int16 LDS window (round 5): the wide windows of
// quality 9 / 10 decimators (48k -> 11.025k, 44.1k -> 16k / 8k ...) held only a fraction of a tile's periods as floats and had no
// int16 window like their fp32 siblings (kernels_period.hip, W16).  Same kernel, same rows as doubles; the loop is
// FirLoopAsm64<..., W16 = true>: samples sign-extended and widened behind the LDS read (v_cvt_f64_i32), one v_fma_f64 per
// tap and half -- deps/speex/resample.c:389-435, :501-558 with exact products and fp64 sums.  int16 calls only, and only
// while the histories hold PCM values (engine.cpp, Batch::float_seen_).
#ifdef SPEEXHIP_STAMPS
#undef SPEEXHIP_STAMPS  // (the diagnostics stamps belong to the fp32 translation unit)
#endif
#include "kernels_period_impl.h"

namespace speexhip {

hipError_t dispatch_period64_w16(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                             dim3 grid, uint32_t threads, bool float_io, hipStream_t stream) {
#define SPEEXHIP_P64_T(RV, CTV, ONE, PADV, TV, CGV)                                                                           \
  (p.ksplit > 1 ? launch_rc<RV, CTV, ONE, PADV, TV, CGV, true, true, 1>(p, pack, grid, threads, t.window_bytes, stream)      \
                : launch_rc<RV, CTV, ONE, PADV, TV, CGV, true, false, 1>(p, pack, grid, threads, t.window_bytes, stream))
#define SPEEXHIP_P64(RV, CTV, ONE, PADV, CGV) \
  return SPEEXHIP_P64_T(RV, CTV, ONE, PADV, int16_t, CGV)
  const bool padded = t.pad != 0;
  if (!t.a64 || !t.w16 || float_io || (padded && t.r != 10)) return hipErrorInvalidValue;
  if (t.ct == 1 && t.cgroups == 1) {
    if (t.r == 5) SPEEXHIP_P64(5, 1, true, false, 0);
    if (!padded) SPEEXHIP_P64(10, 1, true, false, 0);
    SPEEXHIP_P64(10, 1, true, true, 0);
  }
  // frames of three, five, seven channels on single-channel lanes (round 5: until then quality 9 / 10 on those layouts
  // kept the fp32 chain)
#define SPEEXHIP_P64_ODD(CGV)                         \
  if (t.ct == 1 && t.cgroups == CGV) {                \
    if (t.r == 5) SPEEXHIP_P64(5, 1, false, false, CGV);  \
    if (!padded) SPEEXHIP_P64(10, 1, false, false, CGV);  \
    SPEEXHIP_P64(10, 1, false, true, CGV);                \
  }
  SPEEXHIP_P64_ODD(3)
  SPEEXHIP_P64_ODD(5)
  SPEEXHIP_P64_ODD(7)
#undef SPEEXHIP_P64_ODD
  if (t.ct == 1) return hipErrorInvalidValue;
  if (t.cgroups == 1) {
    if (t.r == 5) SPEEXHIP_P64(5, 2, true, false, 0);
    if (!padded) SPEEXHIP_P64(10, 2, true, false, 0);
    SPEEXHIP_P64(10, 2, true, true, 0);
  }
#define SPEEXHIP_P64_CG(CGV)                          \
  if (t.cgroups == CGV) {                             \
    if (t.r == 5) SPEEXHIP_P64(5, 2, false, false, CGV);  \
    if (!padded) SPEEXHIP_P64(10, 2, false, false, CGV);  \
    SPEEXHIP_P64(10, 2, false, true, CGV);                \
  }
  SPEEXHIP_P64_CG(2)
  SPEEXHIP_P64_CG(3)
  SPEEXHIP_P64_CG(4)
#undef SPEEXHIP_P64_CG
#undef SPEEXHIP_P64
#undef SPEEXHIP_P64_T
  return hipErrorInvalidValue;
}


// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(period64_w16)
}  // namespace speexhip
