// kernels_period_frames.hip -- the period kernel's instances for frames of TEN, TWELVE and SIXTEEN channels (5, 6, 8 channel
// pairs per frame; late in round 5).  The reference's path is generic in the channel count (deps/speex/resample.c:968-1036 loops over
// st->nb_channels); here frames beyond 8 channels ran the C++ FIR loop with the frame stride in a register until now -- no int16
// window, no tap-range shares: 32 streams x 131 072 frames of 48k -> 11.025k took 407 / 463 / 787 us (10 / 12 / 16 ch) beside 141
// for 8.  Same arithmetic per output as every fp32-chain instance: resample.c:331-384 / :438-496 with the effective taps.
#ifdef SPEEXHIP_STAMPS
#undef SPEEXHIP_STAMPS  // (the diagnostics stamps belong to the fp32 translation unit)
#endif
#include "kernels_period_impl.h"

namespace speexhip {

hipError_t dispatch_period_frames(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack, dim3 grid, uint32_t threads,
                                  bool float_io, hipStream_t stream) {
  if (t.pp || t.a64 || t.ct != 2 || (t.cgroups != 5 && t.cgroups != 6 && t.cgroups != 8) || (t.pad != 0 && t.r != 10) ||
      (t.w16 && float_io))
    return hipErrorInvalidValue;
#define SPEEXHIP_FR_KS(RV, CGV, PADV, TV, W)                                                                                  \
  return p.ksplit > 1 ? launch_rc<RV, 2, false, PADV, TV, CGV, W, true, 0>(p, pack, grid, threads, t.window_bytes, stream)    \
                      : launch_rc<RV, 2, false, PADV, TV, CGV, W, false, 0>(p, pack, grid, threads, t.window_bytes, stream)
#define SPEEXHIP_FR(RV, CGV, PADV)                             \
  {                                                            \
    if (float_io) SPEEXHIP_FR_KS(RV, CGV, PADV, float, false); \
    if (t.w16) SPEEXHIP_FR_KS(RV, CGV, PADV, int16_t, true);   \
    SPEEXHIP_FR_KS(RV, CGV, PADV, int16_t, false);             \
  }
#define SPEEXHIP_FR_FRAME(CGV)                 \
  {                                            \
    if (t.r == 5) SPEEXHIP_FR(5, CGV, false)   \
    if (t.pad == 0) SPEEXHIP_FR(10, CGV, false) \
    SPEEXHIP_FR(10, CGV, true)                 \
  }
  if (t.cgroups == 5) SPEEXHIP_FR_FRAME(5)
  if (t.cgroups == 6) SPEEXHIP_FR_FRAME(6)
  SPEEXHIP_FR_FRAME(8)
#undef SPEEXHIP_FR_FRAME
#undef SPEEXHIP_FR
#undef SPEEXHIP_FR_KS
}

// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(period_frames)
}  // namespace speexhip
