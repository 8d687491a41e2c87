// kernels_slide_f32.hip -- the slide kernel's instantiations for float samples (see kernels_slide_impl.h).
#include "kernels_slide_impl.h"

namespace speexhip {
template hipError_t launch_slide_shape<float>(const SlidePlan &, const SlideParams &, const DescPack *, dim3,
                                           uint32_t, size_t, hipStream_t);
// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(slide_f32)

}  // namespace speexhip
