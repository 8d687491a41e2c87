// tools/check_rpi.hip -- one-off check (not part of the product): v_cvt_rpi_i32_f32 on gfx950 against
// floor(x + 0.5) evaluated in double (the reference's WORD2INT, arch.h:208-209) for every float in
// [-40000, 40000] on a 1/64 grid plus the neighbours (+-1 ulp) of every half-integer.
// build+run: hipcc --offload-arch=gfx950 -O2 tools/check_rpi.hip -o /tmp/check_rpi && /tmp/check_rpi
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, int* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r;
  asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x[i]));
  y[i] = r;
}
int main() {
  std::vector<float> xs;
  for (int i = -40000 * 64; i <= 40000 * 64; i++) xs.push_back(i / 64.0f);
  for (int h = -40000; h <= 40000; h++) {
    const float t = h + 0.5f;
    xs.push_back(std::nextafterf(t, -1e30f));
    xs.push_back(t);
    xs.push_back(std::nextafterf(t, 1e30f));
  }
  const int n = (int)xs.size();
  float* dx; int* dy;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(dx, dy, n);
  std::vector<int> ys(n);
  hipMemcpy(ys.data(), dy, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; i++) {
    const int want = (int)std::floor(0.5 + (double)xs[i]);
    if (ys[i] != want && bad++ < 10) printf("x=%.9g got %d want %d\n", xs[i], ys[i], want);
  }
  printf("checked %d values, %ld mismatches\n", n, bad);
  return bad != 0;
}
