#!/usr/bin/env python3
"""tools/gen_ubench_loop.py -- writes tools/ubench_loop.hip: the period kernel's FIR inner loop as
hand-written gfx950 ISA, in the variants that were weighed against each other in round 3 (design input,
not part of the product).  Every variant: lane = period x channel pair, R accumulator pairs, taps
wave-uniform in SGPRs fed to v_pk_fma_f32, samples from LDS, two tap banks, lgkmcnt(0) before each bank.

usage: python tools/gen_ubench_loop.py && hipcc --offload-arch=gfx950 -O3 tools/ubench_loop.hip -o tools/ubench_loop
"""
import os

# A scheme: R phases per wave, steps per bank, SGPR homes of the two banks as (first register, count) pieces,
# each piece one scalar load (s32 is reserved by the backend; <= s73 under the 80-SGPR cap).
SCHEMES = {
    "r10b20": dict(R=10, steps=2, A=[(28, 4), (36, 16)], B=[(52, 16), (68, 4)]),
    "r10b30": dict(R=10, steps=3, A=[(4, 16), (20, 8), (28, 4), (36, 2)], B=[(40, 16), (56, 8), (64, 4), (38, 2)]),
    "r16b32": dict(R=16, steps=2, A=[(4, 16), (36, 16)], B=[(52, 16), (20, 8), (28, 4), (68, 4)]),
    "r16b16": dict(R=16, steps=1, A=[(36, 16)], B=[(52, 16)]),
    "r20b20": dict(R=20, steps=1, A=[(28, 4), (36, 16)], B=[(52, 16), (68, 4)]),
}


def tap_reg(bank, t):
    for first, n in bank:
        if t < n:
            return first + t
        t -= n
    raise ValueError


def clobbers(sc):
    regs = []
    for first, n in sc["A"] + sc["B"]:
        regs += list(range(first, first + n))
    return regs


def fma(i, bank, t, x):
    r = tap_reg(bank, t)
    hi = r & 1
    p = r - hi
    return "v_pk_fma_f32 %%[a%d], s[%d:%d], %%[%s], %%[a%d] op_sel:[%d,0,0] op_sel_hi:[%d,1,1]" % (i, p, p + 1, x, i, hi, hi)


def fma_bank(sc, bank, xs):
    return [fma(i, bank, u * sc["R"] + i, xs[u]) for u in range(sc["steps"]) for i in range(sc["R"])]


def loads(bank, byte_off):
    out, t = [], 0
    for f, m in bank:
        out.append("s_load_dword%s s[%d:%d], %%[p], %%[off] offset:0x%x" % ("x%d" % m if m > 1 else "", f, f + m - 1, byte_off + 4 * t))
        t += m
    return out


ROW_BYTES = 520  # period-minor window (round 4): a row = one offset inside the period for 64 (+1) periods of 8 bytes


def ds(x, off, kind):
    if kind == "same":     # every lane the same address: broadcast, no bank work (diagnostic)
        return "ds_read_b64 %%[%s], %%[zero] offset:%d" % (x, off)
    if kind == "contig":   # period-minor window: the 64 lanes read 512 consecutive bytes, a step is a row further
        return "ds_read_b64 %%[%s], %%[addrc] offset:%d" % (x, off // 8 * ROW_BYTES)
    return "ds_read_b64 %%[%s], %%[addr] offset:%d" % (x, off)


def variant(scheme, smem=True, lds="b64", waits=2):
    """asm lines.  Loop contract: %[off] runs from 2^32 - ITERS*trip_bytes up to 0 (the carry of its last
    step ends the loop); %[p] + %[off] = the row; %[addr] = LDS byte address of the lane's sample of step 0.
    waits: 2 = lgkmcnt(0) before each bank; 1 = one wait per two banks, the samples of the NEXT pair of banks
    requested right behind it into a second register set (only without scalar loads: diagnostic)."""
    sc = SCHEMES[scheme]
    S = sc["steps"]
    bank_bytes = 4 * S * sc["R"]
    xa = ["x%d" % u for u in range(S)]
    xb = ["x%d" % (S + u) for u in range(S)]
    wait = ["s_waitcnt lgkmcnt(0)"] if (smem or lds) else []
    rd = lambda xs, first_step: [ds(x, 8 * (first_step + u), lds) for u, x in enumerate(xs)] if lds else []
    pro = loads(sc["A"], 0) + ([] if smem else loads(sc["B"], bank_bytes) + ["s_waitcnt lgkmcnt(0)"])
    fa, fb = fma_bank(sc, sc["A"], xa), fma_bank(sc, sc["B"], xb)
    adv = ["v_add_u32 %%[addr], %d, %%[addr]" % (16 * S)] if lds else []
    if lds == "contig":
        adv = ["v_add_u32 %%[addrc], 0x%x, %%[addrc]" % (2 * S * ROW_BYTES)]
    if waits == 2:
        pro += rd(xa, 0)
        body = ["1:"] + wait + (loads(sc["B"], bank_bytes) if smem else []) + rd(xb, S) + fa
        body += wait + (loads(sc["A"], 2 * bank_bytes) if smem else []) + rd(xa, 2 * S) + adv + fb
        body += ["s_add_u32 %%[off], %%[off], 0x%x" % (2 * bank_bytes), "s_cbranch_scc0 1b"]
        return pro + body + wait
    assert not smem
    pro += rd(xa, 0) + rd(xb, S)
    ya = ["y%d" % u for u in range(S)]
    yb = ["y%d" % (S + u) for u in range(S)]
    body = ["1:"] + wait + rd(ya + yb, 2 * S) + fa + fb
    body += wait + rd(xa + xb, 4 * S) + ["v_add_u32 %%[addr], %d, %%[addr]" % (32 * S)]
    body += fma_bank(sc, sc["A"], ya) + fma_bank(sc, sc["B"], yb)
    body += ["s_add_u32 %%[off], %%[off], 0x%x" % (4 * bank_bytes), "s_cbranch_scc0 1b"]
    return pro + body + wait


# name, scheme, kwargs, waves per workgroup, description
VARIANTS = [
    ("ship", "r10b20", {}, 16, "R=10, banks of 20 taps (2 steps), lgkmcnt(0) x2 per 40 FMAs: the shipping structure"),
    ("nolds", "r10b20", dict(lds=None), 16, "  ... without the LDS sample reads (diagnostic)"),
    ("nosmem", "r10b20", dict(smem=False), 16, "  ... without the scalar tap loads (diagnostic)"),
    ("bare", "r10b20", dict(smem=False, lds=None), 16, "  ... FMAs + count only (diagnostic)"),
    ("same", "r10b20", dict(lds="same"), 16, "  ... every lane reads the same LDS address (diagnostic)"),
    ("contig", "r10b20", dict(lds="contig"), 16, "  ... period-minor window: a wave reads 512 consecutive bytes per step (round 4)"),
    ("contignos", "r10b20", dict(lds="contig", smem=False), 16, "  ... the same without the scalar tap loads (diagnostic)"),
    ("nosmem1w", "r10b20", dict(smem=False, waits=1), 16, "  ... no tap loads, ONE wait per 40 FMAs, samples requested 40 FMAs ahead (diagnostic)"),
    ("b30", "r10b30", {}, 16, "R=10, banks of 30 taps (3 steps): a wait per 30 FMAs"),
    ("b30nolds", "r10b30", dict(lds=None), 16, "  ... without the LDS sample reads (diagnostic)"),
    ("r16b32", "r16b32", {}, 10, "R=16, banks of 32 taps (2 steps), 10 waves per workgroup"),
    ("r16b16", "r16b16", {}, 10, "R=16, banks of 16 taps (1 step), 10 waves per workgroup"),
    ("r16b32w16", "r16b32", {}, 16, "R=16, banks of 32 taps, 16 waves per workgroup (occupancy as R=10; diagnostic)"),
    ("r20b20", "r20b20", {}, 8, "R=20, banks of 20 taps (1 step), 8 waves per workgroup"),
    ("r20b20w16", "r20b20", {}, 16, "R=20, banks of 20 taps, 16 waves per workgroup (diagnostic)"),
]

HEAD = r'''// tools/ubench_loop.hip -- GENERATED by tools/gen_ubench_loop.py; do not edit.  Design input for the period
// kernel's FIR loop (round 3): the loop as hand-written gfx950 ISA in several variants, timed in steady state
// (every wave repeats its group REPS times; 2 workgroups per CU, or 1).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 36;   // trips of the asm loop per group
'''

KERNEL = r'''
template <int V, int R, int TRIP_BYTES>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80)))
void fir(const float *__restrict__ rows, float *__restrict__ out, int reps, int num, int lds_floats,
         unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) xs[i] = (float)((i * 2654435761u) >> 17) - 16384.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  f32x2 acc[R];
#pragma unroll
  for (int i = 0; i < R; i++) acc[i] = f32x2{0.f, 0.f};
  const float *__restrict__ trow = rows + (size_t)wave * ITERS * (TRIP_BYTES / 4);
  const float *xp = xs + lane * num * 2 + wave * 18;
  for (int rep = 0; rep < reps; rep++) {
    // %[p] + %[off] walks the row: off runs from 2^32 - ITERS*TRIP_BYTES up to 0, the carry of its last step ends the loop
    const char *p = reinterpret_cast<const char *>(trow) + ITERS * TRIP_BYTES - (1ll << 32);
    uint32_t off = 0u - ITERS * (uint32_t)TRIP_BYTES;
    uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xp));
    uint32_t addrc = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xs)) + lane * 8 + (wave & 1) * 8;  // period-minor window
    f32x2 x0, x1, x2, x3, x4, x5, y0, y1, y2, y3;
    x0 = x1 = x2 = x3 = x4 = x5 = y0 = y1 = y2 = y3 = f32x2{(float)lane, 1.f};
    const uint32_t zero = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(xs)) + wave * 144;
    switch (V) {
@CASES@
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < R; i++) s += acc[i].x + acc[i].y;
  asm volatile("" :: "v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {  // shader cycles and 100 MHz ticks over the loops of wave 0
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int R, int TRIP_BYTES>
int run(const char *name, const float *rows, float *out, int blocks, int waves, int reps) {
  static unsigned long long *clk = nullptr;
  if (clk == nullptr) CHECK(hipMalloc(&clk, 2 * 4096 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto k = fir<V, R, TRIP_BYTES>;
  CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int lds = 77824;  // (2 per CU; room for the longest walk of any variant)
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), lds, 0, rows, out, reps, 147, lds / 4, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  const double fma = (double)blocks * waves * reps * ITERS * (TRIP_BYTES / 4) * 128.0;
  std::vector<unsigned long long> h(2 * blocks);
  CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> ghz, cyc;
  for (int b = 0; b < blocks; b++) { ghz.push_back(h[2 * b] / (h[2 * b + 1] * 10.0)); cyc.push_back((double)h[2 * b]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  // cycles per v_pk_fma_f32 per SIMD: wave 0's loop cycles x (SIMDs busy) / FMAs issued on its SIMD (waves share 4 SIMDs evenly)
  const double fma_per_simd = (double)reps * ITERS * (TRIP_BYTES / 4) * waves * (blocks / 256) / 4.0;
  printf("%-10s blocks=%4d x %2d waves: %8.1f us  %6.1f TFLOP/s   in-kernel clock %.2f GHz, %.2f cycles per FMA per SIMD\n", name, blocks, waves,
         best * 1e3, 2 * fma / best / 1e9, ghz[blocks / 2], cyc[blocks / 2] / fma_per_simd);
  return 0;
}
'''


def main():
    cases, calls, descs = [], [], []
    for n, (name, scheme, kw, waves, desc) in enumerate(VARIANTS):
        sc = SCHEMES[scheme]
        lines = variant(scheme, **kw)
        trip = 4 * sc["R"] * sc["steps"] * (4 if kw.get("waits") == 1 else 2)
        asm = "\n".join('          "%s\\n"' % l for l in lines)
        ops = ", ".join('[a%d] "+v"(acc[%d])' % (i, i) for i in range(sc["R"]))
        xs = ", ".join('[%s] "+v"(%s)' % (x, x) for x in ["x0", "x1", "x2", "x3", "x4", "x5"][:2 * sc["steps"]] + (["y0", "y1", "y2", "y3"] if kw.get("waits") == 1 else []))
        cases.append('''      case %d:
        if constexpr (R == %d) {
        asm volatile(
%s
          : %s, %s, [off] "+s"(off), [addr] "+v"(addr), [addrc] "+v"(addrc)
          : [p] "s"(p), [zero] "v"(zero)
          : %s, "scc", "memory");
        }
        break;''' % (n, sc["R"], asm, ops, xs, ", ".join('"s%d"' % r for r in clobbers(sc))))
        calls.append('    run<%d, %d, %d>("%s", rows, out, blocks_per_cu * 256, %d, reps);' % (n, sc["R"], trip, name, waves))
        descs.append("%-10s %s" % (name, desc))
    src = HEAD + KERNEL.replace("@CASES@", "\n".join(cases))
    src += "\nint main(int argc, char **argv) {\n  float *rows, *out;\n"
    src += r'''  CHECK(hipMalloc(&rows, 16 * ITERS * 640 + 65536));
  {
    std::vector<float> h(16 * ITERS * 160 + 16384);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 26)); }
    CHECK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMalloc(&out, (size_t)4096 * 1024 * 4));
  const int reps = argc > 1 ? atoi(argv[1]) : 64;
  for (int i = 0; i < 30; i++) { int blocks_per_cu = 2;
''' + calls[0].replace('"ship"', '"warmup"') + "\n  }\n"
    for d in descs:
        src += '  printf("# %s\\n");\n' % d.replace('"', '\\"')
    src += '  for (int blocks_per_cu : {2, 1}) {\n    printf("---- %d workgroup(s) per CU, %d x %d trips per wave\\n", blocks_per_cu, reps, ITERS);\n'
    src += "\n".join(calls) + "\n  }\n  return 0;\n}\n"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_loop.hip")
    open(path, "w").write(src)
    print("wrote", path)


if __name__ == "__main__":
    main()
