// kernels_period_odd.hip -- the period kernel's instances for frames of THREE, FIVE and SEVEN channels (round 5; three
// channels: their two-period plan -- the phase-pair plans of wide windows have had their own instances since round 4).  Odd channel
// counts run single-channel lanes that carry two periods each (kernels_period_impl.h, lane_ctx); until round 5 these two
// layouts had no ISA loop (csrc/gen_fir_loop.py knew frames of 1, 2, 4, 6, 8 floats) and with it no int16 window and no
// tap-range shares: 32 streams x 131 072 frames of 48k -> 11.025k took 216 us (5 ch) and 344 us (7 ch) beside 139 / 142 us
// for 6 / 8 channels, and seven channels of 44.1k -> 48k 13 % longer than eight (profiles/r05_odd_channels.txt).  Same
// arithmetic per output as every fp32-chain instance: deps/speex/resample.c:331-384 / :438-496 with the effective taps.
#ifdef SPEEXHIP_STAMPS
#undef SPEEXHIP_STAMPS  // (the diagnostics stamps belong to the fp32 translation unit)
#endif
#include "kernels_period_impl.h"

namespace speexhip {

hipError_t dispatch_period_odd(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack, dim3 grid, uint32_t threads,
                               bool float_io, hipStream_t stream) {
  if (t.pp || t.a64 || t.ct != 1 || (t.cgroups != 3 && t.cgroups != 5 && t.cgroups != 7) || (t.pad != 0 && t.r != 10) ||
      (t.w16 && float_io))
    return hipErrorInvalidValue;
#define SPEEXHIP_ODD_KS(RV, CGV, PADV, TV, W)                                                                                 \
  return p.ksplit > 1 ? launch_rc<RV, 1, false, PADV, TV, CGV, W, true, 0>(p, pack, grid, threads, t.window_bytes, stream)    \
                      : launch_rc<RV, 1, false, PADV, TV, CGV, W, false, 0>(p, pack, grid, threads, t.window_bytes, stream)
#define SPEEXHIP_ODD(RV, CGV, PADV)                          \
  {                                                          \
    if (float_io) SPEEXHIP_ODD_KS(RV, CGV, PADV, float, false); \
    if (t.w16) SPEEXHIP_ODD_KS(RV, CGV, PADV, int16_t, true);   \
    SPEEXHIP_ODD_KS(RV, CGV, PADV, int16_t, false);             \
  }
#define SPEEXHIP_ODD_FRAME(CGV)               \
  {                                           \
    if (t.r == 5) SPEEXHIP_ODD(5, CGV, false)    \
    if (t.pad == 0) SPEEXHIP_ODD(10, CGV, false) \
    SPEEXHIP_ODD(10, CGV, true)                  \
  }
  if (t.cgroups == 3) SPEEXHIP_ODD_FRAME(3)
  if (t.cgroups == 5) SPEEXHIP_ODD_FRAME(5)
  SPEEXHIP_ODD_FRAME(7)
#undef SPEEXHIP_ODD_FRAME
#undef SPEEXHIP_ODD
#undef SPEEXHIP_ODD_KS
}

// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(period_odd)
}  // namespace speexhip
