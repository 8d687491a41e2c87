// tools/ubench_ctl.hip -- what do small control-plane operations of ONE stream cost while ANOTHER stream has a
// long kernel running on every CU?  (design input for Batch::install_filter / ~Batch, round 3; not product)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ctl.hip -o tools/ubench_ctl
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(float *out, long long cycles) {
  const long long t0 = clock64();
  float a = threadIdx.x;
  while (clock64() - t0 < cycles) a = a * 1.0001f + 1.f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void tiny(float *p) { p[threadIdx.x] = 1.f; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t busy, ctl;
  CHECK(hipStreamCreateWithFlags(&busy, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&ctl, hipStreamNonBlocking));
  float *big, *d, *pin;
  CHECK(hipMalloc(&big, 8192 * 1024 * 4));
  CHECK(hipMalloc(&d, 1 << 20));
  CHECK(hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault));
  std::vector<float> page(1 << 18, 1.f);
  for (int bytes : {2048, 8192, 16384, 16385, 20480, 32768, 65536, 1 << 20}) {
    for (int pass = 0; pass < 2; pass++) {
      const bool loaded = pass == 1;
      auto start = [&]() { if (loaded) hipLaunchKernelGGL(spin, dim3(8192), dim3(1024), 0, busy, big, 12000000LL); };  // ~5 ms, 4 generations
      auto finish = [&](const char *what, double t0) {
        const double t1 = now_us();
        const bool inflight = loaded && hipStreamQuery(busy) == hipErrorNotReady;
        (void)hipGetLastError();
        hipStreamSynchronize(busy);
        printf("%-44s %8d B  %s: %9.1f us%s\n", what, bytes, loaded ? "beside a 5 ms kernel" : "idle device        ", t1 - t0,
               loaded ? (inflight ? "" : "   (the kernel had ENDED: waited for it)") : "");
      };
      double t0;
      start(); t0 = now_us(); CHECK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, ctl)); CHECK(hipStreamSynchronize(ctl)); finish("H2D async from pinned + stream sync", t0);
      start(); t0 = now_us(); CHECK(hipMemcpyAsync(d, page.data(), bytes, hipMemcpyHostToDevice, ctl)); CHECK(hipStreamSynchronize(ctl)); finish("H2D async from pageable + stream sync", t0);
      start(); t0 = now_us(); CHECK(hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, ctl)); CHECK(hipStreamSynchronize(ctl)); finish("D2H async to pinned + stream sync", t0);
      start(); t0 = now_us(); CHECK(hipMemcpyAsync(page.data(), d, bytes, hipMemcpyDeviceToHost, ctl)); CHECK(hipStreamSynchronize(ctl)); finish("D2H async to pageable + stream sync", t0);
      start(); t0 = now_us(); CHECK(hipMemsetAsync(d, 0, bytes, ctl)); CHECK(hipStreamSynchronize(ctl)); finish("hipMemsetAsync + stream sync", t0);
      start(); t0 = now_us(); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, ctl, d); CHECK(hipStreamSynchronize(ctl)); finish("one-wave kernel + stream sync", t0);
      if (bytes != 2048 && bytes != 65536) continue;
      start(); t0 = now_us(); { float *q; CHECK(hipMalloc(&q, bytes)); finish("hipMalloc", t0); start(); t0 = now_us(); CHECK(hipFree(q)); finish("hipFree", t0); }
      start(); t0 = now_us(); { float *q; CHECK(hipHostMalloc(&q, bytes, hipHostMallocDefault)); finish("hipHostMalloc", t0); start(); t0 = now_us(); CHECK(hipHostFree(q)); finish("hipHostFree", t0); }
      start(); t0 = now_us(); std::memcpy(pin, page.data(), bytes); finish("(host memcpy into pinned)", t0);
    }
  }
  return 0;
}
