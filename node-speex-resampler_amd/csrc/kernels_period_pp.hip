// kernels_period_pp.hip -- the period kernel's phase-pair instances (round 4): lane = (period, channel) for mono, stereo
// and three channels, two PHASES per packed FMA (FirLoopAsmPP, csrc/gen_fir_loop.py) where the other instances give
// the halves of a packed FMA to two periods (odd channel counts) or to the two channels of a pair.  Half the LDS window
// per tile for the same 64 lanes: what the wide windows of down-sampling ratios need.  Same arithmetic per output as the fp32 chain (one FMA per tap, same order):
// deps/speex/resample.c:331-384 / :438-496 with the effective taps, +-1 LSB.
#include "kernels_period_impl.h"  // (diagnostics build: this translation unit has stamps of its own, read by
                                  //  speexhip_debug_stamps_pp below)

namespace speexhip {

hipError_t dispatch_period_pp(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack, dim3 grid, uint32_t threads,
                              bool float_io, hipStream_t stream) {
  if (!t.pp || t.a64 || t.ct != 1 || t.cgroups > 3 || (t.pad != 0 && t.r != 10) || (t.w16 && float_io)) return hipErrorInvalidValue;
  // ONE / CGV: mono is the one-group layout, stereo and three channels carry their frame as a compile-time constant
#define SPEEXHIP_PP_KS(RV, ONE, CGV, PADV, TV, W)                                                                               \
  return p.ksplit > 1 ? launch_rc<RV, 1, ONE, PADV, TV, CGV, W, true, 2>(p, pack, grid, threads, t.window_bytes, stream)         \
                      : launch_rc<RV, 1, ONE, PADV, TV, CGV, W, false, 2>(p, pack, grid, threads, t.window_bytes, stream)
#define SPEEXHIP_PP(RV, ONE, CGV, PADV)                          \
  {                                                              \
    if (float_io) SPEEXHIP_PP_KS(RV, ONE, CGV, PADV, float, false); \
    if (t.w16) SPEEXHIP_PP_KS(RV, ONE, CGV, PADV, int16_t, true);   \
    SPEEXHIP_PP_KS(RV, ONE, CGV, PADV, int16_t, false);             \
  }
#define SPEEXHIP_PP_FRAME(ONE, CGV)               \
  {                                               \
    if (t.r == 5) SPEEXHIP_PP(5, ONE, CGV, false)    \
    if (t.pad == 0) SPEEXHIP_PP(10, ONE, CGV, false) \
    SPEEXHIP_PP(10, ONE, CGV, true)                  \
  }
  if (t.cgroups == 1) SPEEXHIP_PP_FRAME(true, 0)
  if (t.cgroups == 2) SPEEXHIP_PP_FRAME(false, 2)
  SPEEXHIP_PP_FRAME(false, 3)
#undef SPEEXHIP_PP_FRAME
#undef SPEEXHIP_PP
#undef SPEEXHIP_PP_KS
}

#ifdef SPEEXHIP_STAMPS
extern "C" __attribute__((visibility("default"))) int speexhip_debug_stamps_pp(unsigned long long *dst, size_t n, int clear) {
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
SPEEXHIP_WARM_UNIT(period_pp)
}  // namespace speexhip
