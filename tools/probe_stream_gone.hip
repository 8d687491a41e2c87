// tools/probe_stream_gone.hip -- what this runtime does with a stream handle after hipStreamDestroy (round 4,
// design input for Batch::quiesce / chain_to): does destroying a busy stream wait for its work, and what do
// hipStreamSynchronize / hipEventRecord / hipStreamQuery return for the stale handle?  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_stream_gone.hip -o tools/probe_stream_gone
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, int *flag) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (threadIdx.x == 0) *flag = 1;
}
int main() {
  int *flag;
  CHECK(hipHostMalloc(&flag, 4, 0));
  hipEvent_t ev;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int round = 0; round < 3; round++) {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    *flag = 0;
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 500000ull /* 5 ms at 100 MHz */, flag);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t ed = hipStreamDestroy(s);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("round %d: hipStreamDestroy of a stream with 5 ms of work: %s after %.3f ms, kernel finished by then: %d\n", round,
           hipGetErrorName(ed), ms, *flag);
    hipError_t e = hipStreamSynchronize(s);
    printf("  hipStreamSynchronize(stale) = %d %s\n", (int)e, hipGetErrorName(e));
    (void)hipGetLastError();
    e = hipEventRecord(ev, s);
    printf("  hipEventRecord(ev, stale)   = %d %s\n", (int)e, hipGetErrorName(e));
    (void)hipGetLastError();
    e = hipStreamQuery(s);
    printf("  hipStreamQuery(stale)       = %d %s\n", (int)e, hipGetErrorName(e));
    (void)hipGetLastError();
    CHECK(hipDeviceSynchronize());
    printf("  after hipDeviceSynchronize: flag %d\n", *flag);
  }
  return 0;
}
