// tools/ubench_pcie.hip -- what the PCIe link of this box gives a host-fed call (design input for the pinned-input
// path of round 6, not product): pinned copies one way and both ways at once (copy engines), and KERNELS that read
// pinned host memory, write it, and do both at once -- interleaved per thread, or in two phases like a launch of one
// generation of workgroups (all stage, then all store).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// grid-stride 16-byte copy src -> dst
__global__ void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) {
    uint4 v = src[i];
    v.x += 1;
    dst[i] = v;
  }
}
// a workgroup stages `tile16` x 16 bytes into LDS, barrier, then stores them: one generation when the grid is resident at once
__global__ void stage_then_store(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, uint32_t tile16) {
  extern __shared__ uint4 lds[];
  for (size_t t0 = (size_t)blockIdx.x * tile16; t0 < n16; t0 += (size_t)gridDim.x * tile16) {
    for (uint32_t j = threadIdx.x; j < tile16 && t0 + j < n16; j += blockDim.x) lds[j] = src[t0 + j];
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < tile16 && t0 + j < n16; j += blockDim.x) {
      uint4 v = lds[j];
      v.x += 1;
      dst[t0 + j] = v;
    }
    __syncthreads();
  }
}
int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 4;
  const size_t n = mb << 20, n16 = n / 16;
  char *din, *dout, *pin_in, *pin_out;
  CHECK(hipMalloc(&din, n)); CHECK(hipMalloc(&dout, n));
  CHECK(hipHostMalloc(&pin_in, n)); CHECK(hipHostMalloc(&pin_out, n));
  memset(pin_in, 1, n); memset(pin_out, 2, n);
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto bench = [&](const char *name, double bytes, auto fn) {
    for (int i = 0; i < 5; i++) fn();
    double best = 1e30, sum = 0;
    for (int i = 0; i < 20; i++) { double t0 = now(); fn(); double t = now() - t0; best = t < best ? t : best; sum += t; }
    printf("%-86s best %8.1f us  mean %8.1f us  %6.1f GB/s (best)\n", name, best, sum / 20, bytes / best / 1e3);
  };
  printf("buffers of %zu MiB\n", mb);
  bench("copy engine: pinned H2D", n, [&] { CHECK(hipMemcpyAsync(din, pin_in, n, hipMemcpyHostToDevice, s1)); CHECK(hipStreamSynchronize(s1)); });
  bench("copy engine: pinned D2H", n, [&] { CHECK(hipMemcpyAsync(pin_out, dout, n, hipMemcpyDeviceToHost, s1)); CHECK(hipStreamSynchronize(s1)); });
  bench("copy engine: pinned H2D on s1 || D2H on s2 (bytes = both)", 2.0 * n, [&] {
    CHECK(hipMemcpyAsync(din, pin_in, n, hipMemcpyHostToDevice, s1));
    CHECK(hipMemcpyAsync(pin_out, dout, n, hipMemcpyDeviceToHost, s2));
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  for (int wgs : {64, 256, 1024}) {
    char name[160];
    snprintf(name, sizeof name, "kernel %4d x 256: reads pinned -> writes HBM", wgs);
    bench(name, n, [&] { hipLaunchKernelGGL(copy16, dim3(wgs), dim3(256), 0, s1, (const uint4 *)pin_in, (uint4 *)dout, n16); CHECK(hipStreamSynchronize(s1)); });
    snprintf(name, sizeof name, "kernel %4d x 256: reads HBM -> writes pinned", wgs);
    bench(name, n, [&] { hipLaunchKernelGGL(copy16, dim3(wgs), dim3(256), 0, s1, (const uint4 *)din, (uint4 *)pin_out, n16); CHECK(hipStreamSynchronize(s1)); });
    snprintf(name, sizeof name, "kernel %4d x 256: reads pinned -> writes pinned, interleaved (bytes = both)", wgs);
    bench(name, 2.0 * n, [&] { hipLaunchKernelGGL(copy16, dim3(wgs), dim3(256), 0, s1, (const uint4 *)pin_in, (uint4 *)pin_out, n16); CHECK(hipStreamSynchronize(s1)); });
  }
  bench("kernel reads pinned -> HBM on s1 || kernel HBM -> pinned on s2 (bytes = both)", 2.0 * n, [&] {
    hipLaunchKernelGGL(copy16, dim3(256), dim3(256), 0, s1, (const uint4 *)pin_in, (uint4 *)dout, n16);
    hipLaunchKernelGGL(copy16, dim3(256), dim3(256), 0, s2, (const uint4 *)din, (uint4 *)pin_out, n16);
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  // tiles of 76 KB staged through LDS by 1024-lane workgroups, like the period kernel's windows
  const uint32_t tile16 = 76 * 1024 / 16;
  for (int wgs : {16, 32, 64, 128, 512}) {
    char name[160];
    snprintf(name, sizeof name, "kernel %4d x 1024, 76 KB tiles through LDS, pinned -> pinned (bytes = both)", wgs);
    bench(name, 2.0 * n, [&] { hipLaunchKernelGGL(stage_then_store, dim3(wgs), dim3(1024), tile16 * 16, s1, (const uint4 *)pin_in, (uint4 *)pin_out, n16, tile16); CHECK(hipStreamSynchronize(s1)); });
  }
  bench("empty wait: hipStreamSynchronize on an idle stream", 0, [&] { CHECK(hipStreamSynchronize(s1)); });
  return 0;
}
