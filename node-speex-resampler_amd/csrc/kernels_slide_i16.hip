// kernels_slide_i16.hip -- the slide kernel's instantiations for int16_t samples (see kernels_slide_impl.h).
#include "kernels_slide_impl.h"

namespace speexhip {
template hipError_t launch_slide_shape<int16_t>(const SlidePlan &, const SlideParams &, const DescPack *, dim3,
                                           uint32_t, size_t, hipStream_t);
// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(slide_i16)

}  // namespace speexhip
