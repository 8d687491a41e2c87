// tools/ubench_pair.hip -- design input for the period kernel (not part of the product): the FIR
// inner loop under the two packings of v_pk_fma_f32 with wave-uniform SGPR taps,
//   CH  : lane = one period x one channel PAIR, one tap per instruction (4 B of taps per v_pk_fma_f32;
//         the shipping kernel) -- a 64-period tile, 76 KB window
//   PH  : lane = one period x one channel, the instruction covers a PHASE pair (8 B of taps per
//         v_pk_fma_f32, sample broadcast) -- a 32-period stereo tile, 38 KB window
// at several workgroup shapes.  Same bank pipeline as kernels_period.hip (wait A | loads B | FMAs A | ...).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_pair.hip -o tools/ubench_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int R = 10;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void fma_ch(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x, bool hi) {
  if (hi) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}
// acc.{x,y} += tap_pair.{x,y} * x.(hi ? y : x)
__device__ __forceinline__ void fma_ph(f32x2 &acc, const f32x2 &tap_pair, const f32x2 &x, bool hi) {
  if (hi) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}

// PH = false: 10 accumulator pairs (channel pair), per bank 2 steps: 10 tap pairs + 2 ds_read_b64
// PH = true : 5 accumulator pairs (phase pairs),  per bank 2 steps: 10 tap pairs + 1 ds_read2_b32
template <bool PH, int GROUPS_PER_WAVE, int ROW_MASK = 15, bool PIN = true>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80)))
void fir(const float *__restrict__ rows, float *__restrict__ out, int l4, int num, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) xs[i] = (float)((i * 2654435761u) >> 17) - 16384.f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float s = 0;
  for (int gi = 0; gi < GROUPS_PER_WAVE; gi++) {
    const int g = wave * GROUPS_PER_WAVE + gi;
    const float *__restrict__ trow = rows + (size_t)(g & ROW_MASK) * l4 * 4 * R;  // ROW_MASK: 15 = a row per wave, 0 = all waves on one row (scalar cache hits)
    // CH: lane = period, frame stride 2 floats.  PH: lane = (period = lane/2, channel = lane&1)
    const float *xp = PH ? xs + (lane >> 1) * num * 2 + (lane & 1) + (g & 15) * 18 : xs + lane * num * 2 + (g & 15) * 18;
    constexpr int NA = PH ? R / 2 : R;
    f32x2 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) acc[i] = f32x2{0.f, 0.f};
    f32x2 ta[R], tb[R], xa[2], xb[2];
    auto load_bank = [&](f32x2 (&t)[R], f32x2 (&x)[2], const float *tp, const float *sp) {
#pragma unroll
      for (int j = 0; j < R; j++) t[j] = *reinterpret_cast<const f32x2 *>(tp + 2 * j);
      if (PH) {
        x[0].x = sp[0];
        x[0].y = sp[2];
      } else {
        x[0] = *reinterpret_cast<const f32x2 *>(sp);
        x[1] = *reinterpret_cast<const f32x2 *>(sp + 2);
      }
    };
    auto touch_bank = [&](const f32x2 (&t)[R], const f32x2 (&x)[2]) {
      if (PH)
        asm volatile("" ::"s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]), "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]),
                     "s"(t[8]), "s"(t[9]), "v"(x[0]));
      else
        asm volatile("" ::"s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]), "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]),
                     "s"(t[8]), "s"(t[9]), "v"(x[0]), "v"(x[1]));
    };
    auto fma_bank = [&](const f32x2 (&t)[R], const f32x2 (&x)[2]) {
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int i = 0; i < NA; i++) {
          if (PH) fma_ph(acc[i], t[u * NA + i], x[0], u != 0);
          else    fma_ch(acc[i], t[(u * R + i) >> 1], x[u], ((u * R + i) & 1) != 0);
        }
    };
    load_bank(ta, xa, trow, xp);
    for (int left = l4; left != 0; left--) {
      if (PIN) touch_bank(ta, xa);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      load_bank(tb, xb, trow + 2 * R, xp + 4);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      fma_bank(ta, xa);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      if (PIN) touch_bank(tb, xb);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      trow += 4 * R;
      xp += 8;
      load_bank(ta, xa, trow, xp);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      fma_bank(tb, xb);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    touch_bank(ta, xa);
#pragma unroll
    for (int i = 0; i < NA; i++) s += acc[i].x + acc[i].y;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool PH, int GPW, int ROW_MASK = 15, bool PIN = true>
int run(const char *name, const float *rows, float *out, int blocks, int threads, int lds_bytes) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto k = fir<PH, GPW, ROW_MASK, PIN>;
  CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  float best = 1e30f;
  for (int rep = 0; rep < 8; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, rows, out, 35, 147, lds_bytes / 4);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // useful FMAs: per wave per group 35 iterations x 40 (step, phase) slots x 128 (lane, half) MACs... both packings:
  // one v_pk_fma_f32 = 128 MACs; CH issues 40 per iteration, PH 20
  const double fma = (double)blocks * (threads / 64) * GPW * 35 * (PH ? 20 : 40) * 128.0;
  printf("%-58s blocks=%5d thr=%4d lds=%6d: %8.1f us  %6.1f TFLOP/s\n", name, blocks, threads, lds_bytes, best * 1e3, 2 * fma / best / 1e9);
  return 0;
}

int main() {
  float *rows, *out;
  CHECK(hipMalloc(&rows, 16 * 35 * 40 * 4 + 8192));
  {
    std::vector<float> h(16 * 35 * 40 + 2048);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 26)); }
    CHECK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMalloc(&out, (size_t)8192 * 1024 * 4));
  // warm the clocks
  for (int i = 0; i < 40; i++) run<false, 1>("warmup", rows, out, 3584, 1024, 76376);
  printf("---- many generations (the 32-stream regime)\n");
  run<false, 1>("CH 16 waves x 1 group, 76 KB (2 WG/CU)  [shipping]", rows, out, 3584, 1024, 76376);
  run<false, 1, 0>("CH 16 waves, all waves on ONE tap row (scalar-cache hits)", rows, out, 3584, 1024, 76376);
  run<false, 1, 3>("CH 16 waves, 4 distinct tap rows", rows, out, 3584, 1024, 76376);
  run<false, 1, 15, false>("CH 16 waves, order left to the compiler (no sched_barrier)", rows, out, 3584, 1024, 76376);
  run<true, 1>("PH 16 waves x 1 group, 38 KB (2 WG/CU)", rows, out, 7168, 1024, 38400);
  run<true, 2>("PH  8 waves x 2 groups, 38 KB (4 WG/CU)", rows, out, 7168, 512, 38400);
  run<true, 4>("PH  4 waves x 4 groups, 38 KB (4 WG/CU by LDS)", rows, out, 7168, 256, 38400);
  run<true, 2>("PH  8 waves x 2 groups, 50 KB (3 WG/CU)", rows, out, 7168, 512, 51200);
  run<false, 2>("CH  8 waves x 2 groups, 76 KB (2 WG/CU, 4 waves/SIMD)", rows, out, 3584, 512, 76376);
  printf("---- one generation (the single-stream regime): 224-256 workgroups\n");
  run<false, 1>("CH  8 waves x 1 group, 76 KB, 224 WG (2 waves/SIMD) [shipping]", rows, out, 224, 512, 76376);
  run<true, 1>("PH 16 waves x 1 group, 38 KB, 224 WG (4 waves/SIMD)", rows, out, 224, 1024, 38400);
  run<true, 1>("PH 16 waves x 1 group, 38 KB, 256 WG", rows, out, 256, 1024, 38400);
  run<false, 1>("CH 16 waves x 1 group, 76 KB, 256 WG (4 waves/SIMD)", rows, out, 256, 1024, 76376);
  return 0;
}
