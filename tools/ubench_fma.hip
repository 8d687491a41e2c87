// tools/ubench_fma.hip -- measure fp32 vector FMA issue rates on gfx950 (design input for the
// FIR kernels): v_fma_f32 vs v_pk_fma_f32, at 1/2/4 waves per SIMD.  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fma.hip -o tools/ubench_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <bool PACKED>
__global__ void fma_loop(float *out, int iters, float b, float c) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 a[16];
  for (int i = 0; i < 16; i++) a[i] = v2{(float)threadIdx.x * 1e-6f + i, (float)i};
  v2 bb = v2{b, b * 0.5f}, cc = v2{c, c};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (PACKED) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bb), "v"(cc));
      } else {
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(bb.x), "v"(cc.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(bb.y), "v"(cc.y));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int packed = 0; packed < 2; packed++)
    for (int threads : {256, 512, 1024}) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        if (packed) hipLaunchKernelGGL(fma_loop<true>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(fma_loop<false>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double fmas = (double)cus * threads * iters * 32.0;  // 16 x 2 scalar FMAs per iteration
      printf("%s threads/CU=%4d (waves/SIMD=%d): %.3f ms  %.1f TFLOP/s  %.1f FMA/clk/CU @2.4GHz\n",
             packed ? "v_pk_fma_f32" : "v_fma_f32   ", threads, threads / 256, best,
             2 * fmas / best / 1e9, fmas / (best * 1e-3) / cus / 2.4e9);
    }
  return 0;
}
