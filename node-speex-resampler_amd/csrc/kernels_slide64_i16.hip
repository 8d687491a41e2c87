// kernels_slide64_i16.hip -- the fp64-accumulate slide kernel (kernels_slide64_impl.h) for int16 calls
// (speex_resampler_process_interleaved_int, deps/speex/resample.c:1061-1082)
#include "kernels_slide64_impl.h"
namespace speexhip {
template hipError_t launch_slide64_shape<int16_t>(const SlidePlan &, const SlideParams &, const double *,                                                   const DescPack *, dim3, uint32_t, size_t, hipStream_t);
// warm-up (engine.cpp, warm_device): one empty launch loads this translation unit's code object onto the device
SPEEXHIP_WARM_UNIT(slide64_i16)

}
