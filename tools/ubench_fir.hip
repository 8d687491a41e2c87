// tools/ubench_fir.hip -- isolates the period-lane FIR inner loop (not part of the product):
// 40 v_pk_fma_f32 per iteration with wave-uniform taps, toggling (a) taps from scalar loads vs
// loop-invariant SGPRs, (b) the 4 LDS sample reads vs a register sample.  Grid and LDS
// footprint as in the real kernel (3584 workgroups x 1024 lanes, 76 KB LDS -> 2 per CU).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fir.hip -o tools/ubench_fir
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int R = 10;

template <bool LOAD_TAPS, bool LDS_READS, int STEPS, bool VGPR_TAPS = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80)))
void fir(const float* __restrict__ rows, float* __restrict__ out, int l4, int num) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int i = threadIdx.x; i < 19000; i += blockDim.x) xs[i] = (float)((i * 2654435761u) >> 17) - 16384.f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const float* __restrict__ trow = rows + (size_t)wave * l4 * 4 * R;
  const float2* xp = reinterpret_cast<const float2*>(xs) + lane * num + wave * 9;
  float2 acc[R];
#pragma unroll
  for (int i = 0; i < R; i++) acc[i] = make_float2(0.f, 0.f);
  float2 xr = make_float2((float)lane, 1.0f);
  const int n_it = l4 * (4 / STEPS);
  for (int it = 0; it < n_it; ++it) {
    float taps[STEPS * R];
#pragma unroll
    for (int k = 0; k < STEPS * R; k++) taps[k] = LOAD_TAPS ? trow[it * STEPS * R + k] : trow[k];
    if (VGPR_TAPS) {
#pragma unroll
      for (int k = 0; k < STEPS * R; k++) asm volatile("" : "+v"(taps[k]));  // force the taps into VGPRs
    }
#pragma unroll
    for (int u = 0; u < STEPS; u++) {
      const float2 x = LDS_READS ? xp[it * STEPS + u] : xr;
#pragma unroll
      for (int i = 0; i < R; i++) {
        acc[i].x = fmaf(taps[u * R + i], x.x, acc[i].x);
        acc[i].y = fmaf(taps[u * R + i], x.y, acc[i].y);
      }
    }
    if (!LDS_READS) asm volatile("" : "+v"(xr.x), "+v"(xr.y));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < R; i++) s += acc[i].x + acc[i].y;
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool A, bool B, int STEPS, bool V = false>
int run(const char* name, const float* rows, float* out, int blocks, int threads) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto k = fir<A, B, STEPS, V>;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 76376, 0, rows, out, 35, 147);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double fma = (double)blocks * (threads / 64) * 35 * 40 * 128.0;
  printf("%-44s blocks=%d thr=%d: %8.1f us  %.1f TFLOP/s\n", name, blocks, threads, best * 1e3, 2 * fma / best / 1e9);
  return 0;
}

int main() {
  float *rows, *out;
  CHECK(hipMalloc(&rows, 16 * 35 * 40 * 4 + 4096));
  {
    std::vector<float> h(16 * 35 * 40 + 1024);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 26)); }
    CHECK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMalloc(&out, (size_t)3584 * 1024 * 4));
  for (int blocks : {3584}) {
    const int thr = 1024;
    run<true, true, 4>("taps s_load x40, LDS reads (real loop)", rows, out, blocks, thr);
    run<false, true, 4>("taps loop-invariant, LDS reads", rows, out, blocks, thr);
    run<true, false, 4>("taps s_load x40, register sample", rows, out, blocks, thr);
    run<false, false, 4>("taps loop-invariant, register sample", rows, out, blocks, thr);
    run<false, false, 4, true>("taps loop-invariant in VGPRs, register sample", rows, out, blocks, thr);
    run<false, true, 4, true>("taps loop-invariant in VGPRs, LDS reads", rows, out, blocks, thr);
  }
  return 0;
}
