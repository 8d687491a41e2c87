// tools/ubench_fma64.hip -- issue rates of the fp64 vector instructions the fp64-accumulate fast path is
// made of (round 4): v_fma_f64 with a wave-uniform SGPR-pair tap, v_cvt_f64_f32, and the mix of a FIR step
// (1 conversion per 10 FMAs).  Design input, not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fma64.hip -o tools/ubench_fma64
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// MODE 0: v_fma_f64 acc, s[tap], v[x], acc   1: v_fma_f64 all-VGPR   2: 10 FMAs + 1 v_cvt_f64_f32   3: v_cvt_f64_f32 only
// 4: v_pk_fma_f32 (reference line)
template <int MODE>
__global__ __launch_bounds__(1024) void loop64(double *out, int iters, double tap_in, float xin) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  double a[16];
  v2 p[16];
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-6 + i; p[i] = v2{(float)i, 1.f}; }
  double tap = tap_in;
  asm volatile("" : "+s"(tap));
  double x = xin + threadIdx.x;
  float xf = xin + threadIdx.x;
  v2 xp = v2{xf, xf};
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(tap), "v"(x));
    } else if (MODE == 1) {
      double tv = tap;
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(tv), "v"(x));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 10; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(tap), "v"(x));
      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(xf));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(xf));
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(tap), "v"(xp));
    }
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, double *out, int cus, int per_iter) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int threads : {256, 512, 1024}) {
    for (int wgs : {1, 2}) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(loop64<MODE>, dim3(cus * wgs), dim3(threads), 0, 0, out, iters, 1.0000001, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double instr = (double)cus * wgs * (threads / 64) * iters * per_iter;  // wave instructions
      printf("%-28s waves/SIMD=%d: %8.3f ms  %.2f cycles per wave instruction per SIMD @2.4GHz  (%.1f G lane-ops/s)\n", name,
             threads / 256 * wgs, best, best * 1e-3 * 2.4e9 / (instr / (cus * 4.0)), instr * 64 / best / 1e6);
    }
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.gcnArchName, cus);
  double *out;
  CHECK(hipMalloc(&out, sizeof(double) * cus * 2 * 1024));
  if (run<0>("v_fma_f64 sgpr tap", out, cus, 16)) return 1;
  if (run<1>("v_fma_f64 vgpr tap", out, cus, 16)) return 1;
  if (run<2>("10 v_fma_f64 + 1 cvt_f64_f32", out, cus, 11)) return 1;
  if (run<3>("v_cvt_f64_f32", out, cus, 16)) return 1;
  if (run<4>("v_pk_fma_f32 sgpr tap", out, cus, 16)) return 1;
  return 0;
}
