// device_probe.hip -- which kind of box is this?  The pool's MI355X boxes differ by 4-6 % in the clock they hold
// under this library's load, so bench lines and the perf gate of different leases are only
// comparable next to the clock they ran at.  speexhip_debug_device_clock() runs ~0.3 ms of packed fp32 FMAs with LDS
// reads on every CU -- the FIR loop's mix -- and reports the shader clock the chip held meanwhile: s_memtime (shader
// cycles) over s_memrealtime (100 MHz), the same two counters tools/stamps.py and tools/ubench_loop read.
// Diagnostics: nothing on the processing path calls this.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "../../include/speexhip_resampler.h"

namespace speexhip {
namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void clock_probe(unsigned long long *out, int iters) {
  __shared__ float xs[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) xs[i] = static_cast<float>(i & 255) * 1e-3f;
  __syncthreads();
  f32x2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = f32x2{static_cast<float>(threadIdx.x), static_cast<float>(i)};
  const f32x2 tap = f32x2{1.0000001f, 0.9999999f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t at = (threadIdx.x * 37u) & 4094u;
  for (int it = 0; it < iters; it++) {
    const f32x2 x = f32x2{xs[at], xs[at + 1]};
    at = (at + 74u) & 4094u;
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(tap), "v"(x));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
  if (threadIdx.x == 0 || s == 12345.678f) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}
}  // namespace
}  // namespace speexhip

extern "C" SPEEXHIP_API int speexhip_debug_device_clock(double *ghz_median, double *ghz_min) {
  if (ghz_median == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  const int blocks = 512;
  unsigned long long *d = nullptr;
  if (hipMalloc(&d, 2 * blocks * sizeof(unsigned long long)) != hipSuccess) return SPEEXHIP_ERR_DEVICE;
  std::vector<unsigned long long> h(2 * blocks);
  hipError_t e = hipSuccess;
  for (int round = 0; round < 3 && e == hipSuccess; round++) {  // (the first rounds bring the clocks up)
    hipLaunchKernelGGL(speexhip::clock_probe, dim3(blocks), dim3(1024), 0, 0, d, 6000);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return SPEEXHIP_ERR_DEVICE;
  std::vector<double> ghz;
  for (int b = 0; b < blocks; b++)
    if (h[2 * b + 1] != 0) ghz.push_back(static_cast<double>(h[2 * b]) / (static_cast<double>(h[2 * b + 1]) * 10.0));
  if (ghz.empty()) return SPEEXHIP_ERR_DEVICE;
  std::sort(ghz.begin(), ghz.end());
  *ghz_median = ghz[ghz.size() / 2];
  if (ghz_min != nullptr) *ghz_min = ghz.front();
  return SPEEXHIP_ERR_SUCCESS;
}

// speexhip_debug_pcie_peak: the link's own rate on this box, the roofline of the host-fed legs of bench.py (round 6).
// Plain pinned hipMemcpyAsync of `bytes` host -> device, device -> host, and both at once on two non-blocking streams;
// best of `reps`, in GB/s.  (Through torch's stream pool the two directions of the both-ways case ran one after the
// other -- 28 + 28 GB/s where this measures 43-48 each way, profiles/r06_ubench_pcie.txt -- hence a probe of the
// library's own.)  Diagnostics: nothing on the processing path calls this.
extern "C" SPEEXHIP_API int speexhip_debug_pcie_peak(uint64_t bytes, int reps, double out_gbs[3]) {
  if (out_gbs == nullptr || bytes == 0 || reps <= 0) return SPEEXHIP_ERR_INVALID_ARG;
  char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
  hipStream_t s1 = nullptr, s2 = nullptr;
  bool ok = hipMalloc(&d_in, bytes) == hipSuccess && hipMalloc(&d_out, bytes) == hipSuccess &&
            hipHostMalloc(&h_in, bytes, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&h_out, bytes, hipHostMallocDefault) == hipSuccess &&
            hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess;
  if (ok) ok = hipMemset(d_out, 1, bytes) == hipSuccess;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int which = 0; ok && which < 3; which++) {
    double best = 1e30;
    for (int r = 0; ok && r < reps + 2; r++) {  // (two warm-up rounds)
      const double t0 = now();
      if (which != 1) ok = ok && hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1) == hipSuccess;
      if (which != 0) ok = ok && hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s2) == hipSuccess;
      ok = ok && hipStreamSynchronize(s1) == hipSuccess && hipStreamSynchronize(s2) == hipSuccess;
      if (r >= 2) best = std::min(best, now() - t0);
    }
    out_gbs[which] = static_cast<double>(bytes) / best / 1e9;  // (both ways: per direction)
  }
  if (s1 != nullptr) (void)hipStreamDestroy(s1);
  if (s2 != nullptr) (void)hipStreamDestroy(s2);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  (void)hipHostFree(h_in);
  (void)hipHostFree(h_out);
  if (!ok) {
    (void)hipGetLastError();
    return SPEEXHIP_ERR_DEVICE;
  }
  return SPEEXHIP_ERR_SUCCESS;
}
