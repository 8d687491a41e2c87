// tools/ubench_copy.hip -- how the host-buffer path can overlap its transfers (design input, not product):
// H2D of 4 MiB and D2H of 4.5 MiB between PAGEABLE host memory and HBM, (a) one after the other on one
// stream, (b) on two streams from one host thread, (c) from two host threads, (d) pinned buffers + host
// memcpy, (e) hipHostRegister on the fly.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(const char *in, char *out, size_t nin, size_t nout) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = i; j < nout / 16; j += stride) {
    uint4 v = ((const uint4 *)in)[j % (nin / 16)];
    v.x += 1;
    ((uint4 *)out)[j] = v;
  }
}
int main() {
  const size_t nin = 4u << 20, nout = 4565228;
  char *hin = (char *)malloc(nin), *hout = (char *)malloc(nout);
  memset(hin, 1, nin); memset(hout, 2, nout);
  char *din, *dout, *pin_in, *pin_out;
  CHECK(hipMalloc(&din, nin)); CHECK(hipMalloc(&dout, nout));
  CHECK(hipHostMalloc(&pin_in, nin)); CHECK(hipHostMalloc(&pin_out, nout));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto bench = [&](const char *name, auto fn) {
    for (int i = 0; i < 5; i++) fn();
    double best = 1e30, sum = 0;
    for (int i = 0; i < 20; i++) { double t0 = now(); fn(); double t = now() - t0; best = t < best ? t : best; sum += t; }
    printf("%-72s best %7.1f us  mean %7.1f us\n", name, best, sum / 20);
  };
  bench("(a) pageable H2D then D2H, one stream", [&] {
    CHECK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1));
    CHECK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s1));
    CHECK(hipStreamSynchronize(s1)); });
  bench("    pageable H2D alone", [&] { CHECK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1)); CHECK(hipStreamSynchronize(s1)); });
  bench("    pageable D2H alone", [&] { CHECK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s1)); CHECK(hipStreamSynchronize(s1)); });
  bench("(b) pageable H2D on s1, D2H on s2, one host thread", [&] {
    CHECK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1));
    CHECK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s2));
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  bench("(c) pageable H2D and D2H from two host threads", [&] {
    std::thread t([&] { CHECK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s2)); CHECK(hipStreamSynchronize(s2)); });
    CHECK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1)); CHECK(hipStreamSynchronize(s1));
    t.join(); });
  bench("(d) memcpy->pinned, H2D ; D2H->pinned, memcpy (serial)", [&] {
    memcpy(pin_in, hin, nin); CHECK(hipMemcpyAsync(din, pin_in, nin, hipMemcpyHostToDevice, s1));
    CHECK(hipMemcpyAsync(pin_out, dout, nout, hipMemcpyDeviceToHost, s1)); CHECK(hipStreamSynchronize(s1)); memcpy(hout, pin_out, nout); });
  bench("    pinned H2D ‖ pinned D2H on two streams (no host memcpy)", [&] {
    CHECK(hipMemcpyAsync(din, pin_in, nin, hipMemcpyHostToDevice, s1));
    CHECK(hipMemcpyAsync(pin_out, dout, nout, hipMemcpyDeviceToHost, s2));
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  bench("    host memcpy 4 MiB + 4.5 MiB alone (one thread)", [&] { memcpy(pin_in, hin, nin); memcpy(hout, pin_out, nout); });
  bench("(e) hipHostRegister both, copies on two streams, unregister", [&] {
    CHECK(hipHostRegister(hin, nin, hipHostRegisterDefault)); CHECK(hipHostRegister(hout, nout, hipHostRegisterDefault));
    CHECK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1));
    CHECK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s2));
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
    CHECK(hipHostUnregister(hin)); CHECK(hipHostUnregister(hout)); });
  // sliced: 4 pieces each way, pageable, two threads
  bench("(f) 4 pieces each way, pageable, two host threads", [&] {
    std::thread t([&] { for (int k = 0; k < 4; k++) CHECK(hipMemcpyAsync(hout + k * (nout / 4), dout + k * (nout / 4), nout / 4, hipMemcpyDeviceToHost, s2)); CHECK(hipStreamSynchronize(s2)); });
    for (int k = 0; k < 4; k++) CHECK(hipMemcpyAsync(din + k * (nin / 4), hin + k * (nin / 4), nin / 4, hipMemcpyHostToDevice, s1));
    CHECK(hipStreamSynchronize(s1)); t.join(); });
  bench("(g) hipHostRegister both, 4 pieces each way at offsets, two streams, one host thread", [&] {
    CHECK(hipHostRegister(hin, nin, hipHostRegisterDefault)); CHECK(hipHostRegister(hout, nout, hipHostRegisterDefault));
    for (int k = 0; k < 4; k++) {
      CHECK(hipMemcpyAsync(din + k * (nin / 4), hin + k * (nin / 4), nin / 4, hipMemcpyHostToDevice, s1));
      CHECK(hipMemcpyAsync(hout + k * (nout / 4), dout + k * (nout / 4), nout / 4, hipMemcpyDeviceToHost, s2));
    }
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
    CHECK(hipHostUnregister(hin)); CHECK(hipHostUnregister(hout)); });
  bench("(h) pinned (hipHostMalloc), 4 pieces each way at offsets, two streams", [&] {
    for (int k = 0; k < 4; k++) {
      CHECK(hipMemcpyAsync(din + k * (nin / 4), pin_in + k * (nin / 4), nin / 4, hipMemcpyHostToDevice, s1));
      CHECK(hipMemcpyAsync(pin_out + k * (nout / 4), dout + k * (nout / 4), nout / 4, hipMemcpyDeviceToHost, s2));
    }
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  hipEvent_t ev[4];
  for (int k = 0; k < 4; k++) CHECK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
  bench("(i) pinned, 4 pieces: H2D k on s1, event, D2H k on s2 after the event", [&] {
    for (int k = 0; k < 4; k++) {
      CHECK(hipMemcpyAsync(din + k * (nin / 4), pin_in + k * (nin / 4), nin / 4, hipMemcpyHostToDevice, s1));
      CHECK(hipEventRecord(ev[k], s1));
      CHECK(hipStreamWaitEvent(s2, ev[k], 0));
      CHECK(hipMemcpyAsync(pin_out + k * (nout / 4), dout + k * (nout / 4), nout / 4, hipMemcpyDeviceToHost, s2));
    }
    CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  for (int K : {1, 2, 4, 8}) {
    char name[128];
    snprintf(name, sizeof name, "(j) pinned, %d pieces alternating on two streams: H2D, kernel, D2H each", K);
    bench(name, [&] {
      for (int k = 0; k < K; k++) {
        hipStream_t st = (k & 1) ? s2 : s1;
        const size_t io = k * (nin / K) / 16 * 16, oo = k * (nout / K) / 16 * 16, ni = nin / K / 16 * 16, no = nout / K / 16 * 16;
        CHECK(hipMemcpyAsync(din + io, pin_in + io, ni, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, din + io, dout + oo, ni, no);
        CHECK(hipMemcpyAsync(pin_out + oo, dout + oo, no, hipMemcpyDeviceToHost, st));
      }
      CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2)); });
  }
  return 0;
}
