// tools/ubench_smem.hip -- how fast do the scalar caches fill from L2?  (design input, round 4: the period kernel's
// phase-pair loops consume 8 bytes of taps per packed FMA through scalar loads; is that path a ceiling?)  Every wave
// streams a 52 KB row with s_load_dwordx16, two loads (128 bytes) between waits, nothing else: bytes per second over the
// chip for rows that are (a) every wave's own, (b) shared by wave w of every workgroup -- what a launch of the
// period kernel does --, (c) shared by all waves of a workgroup, (d) by two or four of them.  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_smem.hip -o tools/ubench_smem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(1024) void k(const char *rows, uint32_t row_bytes, int reps, uint32_t *out) {
  extern __shared__ float xs[];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = blockDim.x >> 6;
  const uint32_t stream = MODE == 0 ? blockIdx.x * waves + wave : MODE == 1 ? wave : MODE == 3 ? wave / 2 : MODE == 4 ? wave / 4 : 0;
  const char *p = rows + static_cast<size_t>(stream) * row_bytes;
  for (int r = 0; r < reps; r++)
    for (uint32_t off = 0; off < row_bytes; off += 64 * INFLIGHT) {
      if (INFLIGHT == 2)
        asm volatile("s_load_dwordx16 s[36:51], %0, %1 offset:0x0\n s_load_dwordx16 s[52:67], %0, %1 offset:0x40\n s_waitcnt lgkmcnt(0)"
                     : : "s"(p), "s"(off) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
                         "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "memory");
      else
        asm volatile("s_load_dwordx16 s[36:51], %0, %1 offset:0x0\n s_waitcnt lgkmcnt(0)"
                     : : "s"(p), "s"(off) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory");
    }
  if (threadIdx.x == 0 && reps < 0) out[blockIdx.x] = xs[0];
}

template <int MODE, int INFLIGHT>
int run(const char *name, const char *rows, int blocks, int threads, uint32_t row_bytes, int reps, size_t lds, uint32_t *out) {
  CHECK(hipFuncSetAttribute((const void *)k<MODE, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(blocks), dim3(threads), lds, 0, rows, row_bytes, reps, out);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(blocks), dim3(threads), lds, 0, rows, row_bytes, reps, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = static_cast<double>(blocks) * (threads / 64) * row_bytes * reps;
  printf("%-34s blocks=%4d x %2d waves, %d load(s) between waits: %8.1f us  %7.2f TB/s over the chip  %6.2f bytes per cycle and CU (2.4 GHz, 256 CUs)\n", name,
         blocks, threads / 64, INFLIGHT, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 2.4e9 / 256);
  return 0;
}

int main() {
  const uint32_t row_bytes = 51840;  // a phase-pair group's row at 48k -> 11.025k q7: 324 trips x 160 bytes
  const size_t total = static_cast<size_t>(512) * 16 * row_bytes;
  char *rows;
  uint32_t *out;
  CHECK(hipMalloc(&rows, total));
  CHECK(hipMemset(rows, 1, total));
  CHECK(hipMalloc(&out, 1 << 20));
  const size_t lds = 80 * 1024;  // two workgroups per CU at most, like the kernels
  for (int blocks : {256, 512}) {
    if (run<0, 2>("own row per wave (HBM / L2)", rows, blocks, 1024, row_bytes, 2, lds, out)) return 1;
    if (run<1, 2>("wave w of every workgroup shares", rows, blocks, 1024, row_bytes, 4, lds, out)) return 1;
    if (run<1, 1>("wave w of every workgroup shares", rows, blocks, 1024, row_bytes, 4, lds, out)) return 1;
    if (run<2, 2>("all waves of a workgroup share", rows, blocks, 1024, row_bytes, 4, lds, out)) return 1;
  }
  // what a workgroup of (phase group, period block) waves would see: two / four waves of a workgroup on every row
  for (int blocks : {256, 512}) {
    if (run<3, 2>("pairs of waves share a row", rows, blocks, 1024, row_bytes, 4, lds, out)) return 1;
    if (run<4, 2>("four waves share a row", rows, blocks, 1024, row_bytes, 4, lds, out)) return 1;
  }
  if (run<1, 2>("wave w shares, 8 waves", rows, 256, 512, row_bytes, 4, lds, out)) return 1;
  if (run<1, 2>("wave w shares, 4 waves", rows, 256, 256, row_bytes, 4, lds, out)) return 1;
  if (run<1, 2>("wave w shares, 128 workgroups", rows, 128, 1024, row_bytes, 4, lds, out)) return 1;
  if (run<1, 2>("wave w shares, 64 workgroups", rows, 64, 1024, row_bytes, 4, lds, out)) return 1;
  return 0;
}
