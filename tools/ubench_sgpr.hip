// tools/ubench_sgpr.hip -- does v_pk_fma_f32 issue slower with an SGPR-pair source than with VGPR
// sources?  (design input: the FIR loops feed their taps as SGPR operands).  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sgpr.hip -o tools/ubench_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 0: all VGPR; 1: SGPR pair, low half broadcast; 2: SGPR pair alternating halves (the FIR loop's form)
__global__ __launch_bounds__(1024) void k(float *out, int iters, float b, float c) {
  f32x2 a[20];
  for (int i = 0; i < 20; i++) a[i] = f32x2{(float)threadIdx.x * 1e-6f + i, (float)i};
  f32x2 bb = f32x2{b, b * 0.5f}, cc = f32x2{c, c * 0.25f};
  f32x2 sb = f32x2{b, b * 0.5f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 20; i++) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bb), "v"(cc));
      else if (MODE == 1 || (i & 1) == 0)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(sb), "v"(cc));
      else
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "s"(sb), "v"(cc));
    }
  }
  float s = 0;
  for (int i = 0; i < 20; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 2 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(k<0>, dim3(cus * 2), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
  for (int mode = 0; mode < 3; mode++)
    for (int blocks_per_cu : {1, 2}) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus * blocks_per_cu), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus * blocks_per_cu), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(cus * blocks_per_cu), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double fmas = (double)cus * blocks_per_cu * 1024 * iters * 40.0;
      printf("%-44s waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n",
             mode == 0 ? "v_pk_fma_f32 all-VGPR" : mode == 1 ? "v_pk_fma_f32 SGPR pair (low half)" : "v_pk_fma_f32 SGPR pair (alternating halves)",
             4 * blocks_per_cu, best, 2 * fmas / best / 1e9);
    }
  return 0;
}
