// tools/probe_slots.hip -- which hardware register tells the two workgroups that share a CU apart?  (round 3:
// the period kernel gives the two resident workgroups different wave priorities so that they do not run in
// lockstep; design input, not product)   build: hipcc --offload-arch=gfx950 -O3 tools/probe_slots.hip -o tools/probe_slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(1024) void probe(unsigned long long *out, long long spin) {
  extern __shared__ float xs[];
  xs[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const long long c0 = clock64();
  float a = xs[(threadIdx.x * 7) & 1023];
  while (clock64() - c0 < spin) a = a * 1.0001f + 1.f;
  if (threadIdx.x == 0) {
    out[8 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
    out[8 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));   // LDS_ALLOC
    out[8 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // XCC_ID
    out[8 * blockIdx.x + 3] = t0;
    out[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();
    out[8 * blockIdx.x + 5] = __builtin_amdgcn_s_getreg(5 | (0 << 6) | (31 << 11));   // GPR_ALLOC
    out[8 * blockIdx.x + 6] = (unsigned long long)a;
  }
}
int main() {
  const int blocks = 2048;
  unsigned long long *d;
  CHECK(hipMalloc(&d, blocks * 8 * sizeof(unsigned long long)));
  CHECK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 76376, 0, d, 200000LL);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * 8);
  CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  // group by (xcc, se/sh/cu bits of HW_ID)
  std::map<unsigned long long, std::vector<int>> by_cu;
  for (int b = 0; b < blocks; b++) {
    const unsigned hw = (unsigned)h[8 * b];
    const unsigned long long key = ((h[8 * b + 2] & 0xf) << 32) | (hw & 0x0000ff00u) | ((hw >> 12) & 0xfu) << 16;  // cu_id[11:8], sh/se[15:12]
    by_cu[key].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu) keys for %d workgroups\n", by_cu.size(), blocks);
  int shown = 0;
  std::map<unsigned, int> lds_values;
  for (auto &kv : by_cu) {
    for (int b : kv.second) lds_values[(unsigned)h[8 * b + 1]]++;
    if (shown++ < 3) {
      printf("CU key %llx:\n", kv.first);
      for (int b : kv.second)
        printf("  wg %4d  HW_ID %08x (wave %u simd %u pipe %u cu %u sh %u se %u tg %u vm %u queue %u state %u me %u)  LDS_ALLOC %08x  GPR_ALLOC %08x  t %llu..%llu\n", b,
               (unsigned)h[8 * b], (unsigned)h[8 * b] & 15, ((unsigned)h[8 * b] >> 4) & 3, ((unsigned)h[8 * b] >> 6) & 3, ((unsigned)h[8 * b] >> 8) & 15,
               ((unsigned)h[8 * b] >> 12) & 1, ((unsigned)h[8 * b] >> 13) & 7, ((unsigned)h[8 * b] >> 16) & 15, ((unsigned)h[8 * b] >> 20) & 15,
               ((unsigned)h[8 * b] >> 24) & 7, ((unsigned)h[8 * b] >> 27) & 7, ((unsigned)h[8 * b] >> 30) & 3, (unsigned)h[8 * b + 1], (unsigned)h[8 * b + 5],
               h[8 * b + 3] % 1000000, h[8 * b + 4] % 1000000);
    }
  }
  printf("LDS_ALLOC values seen:\n");
  for (auto &kv : lds_values) printf("  %08x x %d\n", kv.first, kv.second);
  return 0;
}
