// tools/ubench_sync.hip -- what the wait after a tiny launch costs (design input for the small host-buffer calls):
// an empty one-wave kernel + (a) hipStreamSynchronize, (b) a hipStreamQuery spin, (c) a spin on a word of pinned
// host memory that the kernel writes last (system-scope store), (d) hipEventRecord + hipEventSynchronize.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_sync.hip -o tools/ubench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void tiny(volatile unsigned *flag, unsigned v, float *sink) {
  if (sink != nullptr && threadIdx.x == 0) sink[0] = 1.f;
  if (flag != nullptr && threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(const_cast<unsigned *>(flag), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned *flag; CHECK(hipHostMalloc(&flag, 64, hipHostMallocCoherent)); *flag = 0;
  float *sink; CHECK(hipMalloc(&sink, 64));
  hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int N = 2000;
  auto stat = [&](const char *name, std::vector<double> &t) {
    std::sort(t.begin(), t.end());
    printf("%-44s median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10]);
  };
  for (int warm = 0; warm < 200; warm++) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink); }
  CHECK(hipStreamSynchronize(s));
  std::vector<double> t;
  t.clear();
  for (int i = 0; i < N; i++) { double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink); CHECK(hipStreamSynchronize(s)); t.push_back(now_us() - t0); }
  stat("launch + hipStreamSynchronize", t);
  t.clear();
  for (int i = 0; i < N; i++) { double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink); while (hipStreamQuery(s) == hipErrorNotReady) {} t.push_back(now_us() - t0); }
  stat("launch + hipStreamQuery spin", t);
  t.clear();
  for (int i = 0; i < N; i++) { double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink); CHECK(hipEventRecord(ev, s)); CHECK(hipEventSynchronize(ev)); t.push_back(now_us() - t0); }
  stat("launch + hipEventRecord + EventSynchronize", t);
  t.clear();
  for (int i = 0; i < N; i++) {
    const unsigned want = i + 1;
    double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, flag, want, sink);
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {}
    t.push_back(now_us() - t0);
  }
  stat("launch + spin on a pinned word (kernel writes)", t);
  CHECK(hipStreamSynchronize(s));
  t.clear();
  for (int i = 0; i < N; i++) {
    const unsigned want = 100000 + i;
    double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink);
    CHECK(hipStreamWriteValue32(s, flag, want, 0));
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {}
    t.push_back(now_us() - t0);
  }
  stat("launch + hipStreamWriteValue32 + spin", t);
  CHECK(hipStreamSynchronize(s));
  t.clear();
  for (int i = 0; i < N; i++) {
    const unsigned want = 200000 + i;
    double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, flag, want, nullptr);
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {}
    t.push_back(now_us() - t0);
  }
  stat("launch + signal kernel + spin", t);
  CHECK(hipStreamSynchronize(s));
  t.clear();
  for (int i = 0; i < N; i++) { double t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, 0u, sink); t.push_back(now_us() - t0); if (i % 64 == 63) CHECK(hipStreamSynchronize(s)); }
  stat("hipLaunchKernelGGL alone (host side)", t);
  CHECK(hipStreamSynchronize(s));
  return 0;
}
