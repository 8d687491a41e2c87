// kernels_period64.hip -- the period kernel's instances with an fp64 accumulator (round 4): what FAST mode runs
// for the reference's double kernels (quality 9 and 10: resampler_basic_direct_double, deps/speex/resample.c:389-435,
// resampler_basic_interpolate_double, :501-558) on the ratios of the period kernel (den >= 7).  Same kernel, same
// window, same launch shapes (kernels_period.hip plans them); the FIR loop is FirLoopAsm64 (csrc/gen_fir_loop.py):
// taps as doubles in SGPR pairs, samples widened behind the LDS read, v_fma_f64 -- exact products, fp64 sums, i.e.
// wider than the reference's fp64 sums of fp32-rounded products, where the fp32 FMA chain was narrower.  Layouts:
// mono, stereo, 4 / 6 / 8 channels and (round 5) 3 / 5 / 7 channels -- the ISA loop's; others keep the fp32 chain.
#ifdef SPEEXHIP_STAMPS
#undef SPEEXHIP_STAMPS  // (the diagnostics stamps belong to the fp32 translation unit)
#endif
#include "kernels_period_impl.h"

namespace speexhip {

hipError_t dispatch_period64(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack,
                             dim3 grid, uint32_t threads, bool float_io, hipStream_t stream) {
#define SPEEXHIP_P64_T(RV, CTV, ONE, PADV, TV, CGV)                                                                           \
  (p.ksplit > 1 ? launch_rc<RV, CTV, ONE, PADV, TV, CGV, false, true, 1>(p, pack, grid, threads, t.window_bytes, stream)      \
                : launch_rc<RV, CTV, ONE, PADV, TV, CGV, false, false, 1>(p, pack, grid, threads, t.window_bytes, stream))
#define SPEEXHIP_P64(RV, CTV, ONE, PADV, CGV) \
  return float_io ? SPEEXHIP_P64_T(RV, CTV, ONE, PADV, float, CGV) : SPEEXHIP_P64_T(RV, CTV, ONE, PADV, int16_t, CGV)
  const bool padded = t.pad != 0;
  if (!t.a64 || t.w16 || (padded && t.r != 10)) return hipErrorInvalidValue;
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
SPEEXHIP_WARM_UNIT(period64)
}  // namespace speexhip
