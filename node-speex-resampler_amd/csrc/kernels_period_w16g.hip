// kernels_period_w16g.hip -- the period kernel's int16-window instances for the layouts WITHOUT a generated ISA loop (round 6):
// frames of 9, 11, 13, 14, 15, 17, 18 ... channels.  The reference's path is generic in the channel count
// (deps/speex/resample.c:968-1036 loops over st->nb_channels); here such frames ran the C++ FIR loop over a FLOAT window only,
// so their wide-window decimators held half the periods per tile of their neighbours: 32 streams x 131 072 frames of
// 48k -> 11.025k at 0.10-0.16 of the vector peak for 9 / 11 / 13 / 15 / 17 channels beside 0.33-0.39 for 8 / 10 / 12 / 16
// (profiles/r06_sweep_frames.txt).  The C++ loop now converts the samples it reads from an int16 image (fir_group);
// same arithmetic per output as every fp32-chain instance: resample.c:331-384 / :438-496 with the effective taps.
#ifdef SPEEXHIP_STAMPS
#undef SPEEXHIP_STAMPS  // (the diagnostics stamps belong to the fp32 translation unit)
#endif
#include "kernels_period_impl.h"

namespace speexhip {

hipError_t dispatch_period_w16g(const PeriodPlan &t, const PeriodParams &p, const DescPack *pack, dim3 grid, uint32_t threads,
                                hipStream_t stream) {
  if (!t.w16 || t.pp || t.a64 || p.ksplit > 1 || (t.pad != 0 && t.r != 10)) return hipErrorInvalidValue;
#define SPEEXHIP_W16G(RV, CTV, PADV) return launch_rc<RV, CTV, false, PADV, int16_t, 0, true, false, 0>(p, pack, grid, threads, t.window_bytes, stream)
  if (t.r == 5) {  // never padded (plan_period_r)
    if (t.ct == 2) SPEEXHIP_W16G(5, 2, false);
    SPEEXHIP_W16G(5, 1, false);
  }
  if (t.ct == 2) {
    if (t.pad == 0) SPEEXHIP_W16G(10, 2, false);
    SPEEXHIP_W16G(10, 2, true);
  }
  if (t.pad == 0) SPEEXHIP_W16G(10, 1, false);
  SPEEXHIP_W16G(10, 1, true);
#undef SPEEXHIP_W16G
}

// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(period_w16g)
}  // namespace speexhip
