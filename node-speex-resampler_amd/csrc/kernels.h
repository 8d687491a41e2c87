// kernels.h -- host-callable launchers of the HIP kernels (product code).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "device_types.h"
#include "filter_design.h"

namespace speexhip {

// Kernel attributes (the opt-in for more than 64 KiB of dynamic LDS) are per DEVICE: set them the
// first time a kernel is launched on each device of the process.  `seen` = one bit per device id;
// two threads racing on a first use both set the (idempotent) attribute before either launches.
template <typename K>
inline void opt_in_lds_on_this_device(K kern, std::atomic<uint64_t> &seen) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (seen.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  seen.fetch_or(bit, std::memory_order_release);
}

// Warm-up: each translation unit with kernels is a code object of its own, loaded onto a device by the first launch from
// it (2-4 ms of a process's first call, tools/first_call.py).  Every such unit defines an empty kernel and
// warm_unit_<name>(stream), which launches it; speexhip_warmup runs them all off the caller's path.
#define SPEEXHIP_WARM_UNIT(name)                                             \
  __global__ void warm_kernel_##name() {}                                    \
  void warm_unit_##name(hipStream_t s) {                                     \
    hipLaunchKernelGGL(warm_kernel_##name, dim3(1), dim3(64), 0, s);          \
    (void)hipGetLastError();                                                 \
  }
void warm_unit_exact(hipStream_t s);
void warm_unit_period(hipStream_t s);
void warm_unit_slide_i16(hipStream_t s);
void warm_unit_slide_f32(hipStream_t s);
void warm_unit_slide64_i16(hipStream_t s);
void warm_unit_slide64_f32(hipStream_t s);
void warm_unit_period64(hipStream_t s);
void warm_unit_period_pp(hipStream_t s);
void warm_unit_period_odd(hipStream_t s);
void warm_unit_period_frames(hipStream_t s);
void warm_unit_period64_w16(hipStream_t s);
void warm_unit_period_w16g(hipStream_t s);

// compute units of the calling thread's current device (cached per device id)
inline uint32_t device_compute_units() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = cached[dev & 63].load(std::memory_order_relaxed);
  if (n == 0) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cached[dev & 63].store(n, std::memory_order_relaxed);
  }
  return static_cast<uint32_t>(n);
}

// ---- bit-exact kernels (kernels_exact.hip) -------------------------------------------------
struct ExactGeometry {
  int ct = 1;                  // channels per lane (2 when the channel count is even)
  uint32_t channel_groups = 1;
  uint32_t outs_per_block = 256;
  uint32_t span_cap = 0;
  size_t lds_bytes = 0;
  bool staged = true;          // false: filter/window too large for LDS, read through L2
};
ExactGeometry exact_geometry(const FilterSpec &f, uint32_t channels, size_t lds_budget);
// strides (samples between two frames of a channel) of a launch that is not a plain interleaved one;
// zero = resampler_basic_zero (resample.c:561-591)
struct ExactStrides {
  uint32_t in, out, hist;
};
hipError_t launch_exact(const FilterSpec &f, const ExactGeometry &g, const float *d_table,
                        uint32_t channels, const DescPack *pack,
                        uint32_t n_streams, uint32_t max_n_out, bool float_io, hipStream_t stream,
                        const ExactStrides *strides = nullptr, bool zero = false);

// ---- primary fast kernel: period-lane mapping, taps in SGPRs (kernels_period.hip) ----------
struct PeriodPlan {         // per filter, fixed at init
  bool usable = false;
  uint32_t r = 10, ct = 1, cgroups = 0, groups = 0;
  uint32_t row_len = 0, l4 = 0, tail_frames = 0, lane_periods = 0;
  uint32_t pad = 0;          // LDS bank padding (elements after every period), 0 = none needed
  bool w16 = false;          // the LDS window holds int16 samples (2-byte elements) instead of floats: int16
                             // calls only; twice the periods per tile where the float window limits them
  bool a64 = false;          // fp64 accumulator (round 4): the tap rows are doubles, a loop trip is half as many steps,
                             // rows_floats counts doubles
  bool pp = false;           // phase pairs (round 4; mono): a lane owns ONE period and 2r phases per group -- a tile is
                             // 64 periods, not 128 (half the window), the rows are [step][2r]
  size_t rows_floats = 0, window_bytes = 0;
  bool float_ok = true;      // false (late in round 5): not even one period of the FLOAT window fits the LDS; the plan stands for
                             // its int16-window plan to hang off (int16 calls run over that), float calls take the exact kernel
  uint32_t fold = 1;         // round 6: planned on a folded view of the filter (FilterSpec::fold, period_view): the kernel's
                             // num and den are fold times the filter's
};
// Round 6: ratios with den <= 6 that the slide kernel does not cover (7:6, 11:1, 16:3, 25:1 ...) ran the exact kernel,
// ~5x slower.  They get the period kernel on a FOLDED view: *view = f with num, den and fold multiplied so that den becomes
// a multiple of 5 that is at least 10 (whole groups of five phases); returns true and fills *view (a copy of f, table
// included) when this (filter, channel count) wants it, false (view untouched) otherwise.
bool period_view(const FilterSpec &f, uint32_t channels, FilterSpec *view);
PeriodPlan plan_period(const FilterSpec &f, uint32_t channels, size_t lds_budget, bool w16 = false, bool a64 = false,
                       bool pp = false);
PeriodPlan plan_period_r(const FilterSpec &f, uint32_t channels, size_t lds_budget, uint32_t r, bool w16 = false,
                         bool a64 = false, bool pp = false);
// Does this filter of up to three channels get phase-pair plans beside its other ones (wide windows: num >= 320), and
// should THIS launch run over `pp`?  `two` = the plan the launch would take otherwise (kernels_period.hip).
bool period_wants_pp_plans(const FilterSpec &f, uint32_t channels);
bool period_launch_prefers_pp(const FilterSpec &f, const PeriodPlan &two, const PeriodPlan &pp, const StreamDesc *h_descs,
                              uint32_t n_streams);
// Host only: tiles, phase-group splits, waves with a group, tap-range shares, lanes and whether the workgroups fetch the
// tap rows, as launch_period_plan would set them for this launch of `t` (speexhip_debug_launch_shape).
bool debug_period_shape(const FilterSpec &f, const PeriodPlan &t, uint32_t channels, const StreamDesc *h_descs, uint32_t n_streams,
                        bool float_io, uint32_t out[6]);
// The int16-window plan of a filter whose float plan is `t`, .usable only where it pays: at least 5/4 of the
// periods per tile (the loop converts every sample it reads: ~20 % more vector instructions per tile).
PeriodPlan plan_period_w16(const FilterSpec &f, uint32_t channels, size_t lds_budget, const PeriodPlan &t);
void build_period_rows(const FilterSpec &f, const PeriodPlan &t, std::vector<float> *rows);
// ... of an a64 plan: t.rows_floats doubles, then the per-group tables (uint32, bit-copied into the tail)
void build_period_rows64(const FilterSpec &f, const PeriodPlan &t, std::vector<double> *rows);
// `fine`: the same filter planned with r = 5 (or null); single-generation launches take it
hipError_t launch_period(const FilterSpec &f, const PeriodPlan &t, const float *d_rows, const PeriodPlan *fine,
                         const float *d_rows_fine, uint32_t channels, const StreamDesc *h_descs,
                         const DescPack *pack, uint32_t n_streams, bool float_io,
                         hipStream_t stream, bool fixed_shape = false);
// fixed_shape (SPEEXHIP_MODE_FAST_FIXED, round 5): no tap-range shares / parts -- the only launch-time choice that
// changes the ORDER in which an output's products are summed (partial sums of tap ranges meeting in LDS).  Everything
// else a launch chooses -- phases per wave, int16 or float window, phase pairs, phase-group splits, trimmed head and
// tail trips, tile size -- adds the same products in the same order (or skips exact zeros), so with this flag an
// output's bits depend on the stream alone: not on chunking, batch size, launch size or the GPU's CU count.

// Should this int16 launch run over the int16-window plan rather than `t` (the float-window plan, `has_fine`: with
// an r = 5 companion)?  Launches of few tiles want more and shorter pieces (kernels_period.hip).
bool period_launch_prefers_w16(const FilterSpec &f, const PeriodPlan &t, bool has_fine, const StreamDesc *h_descs,
                               uint32_t n_streams);

// ---- small-ratio fast kernel (kernels_slide.hip): den <= 6, num <= 4 -------------------------
struct SlidePlan {
  bool usable = false;
  bool pair_ch = true;      // packed FMA over channel pairs (even channel count) or phase pairs
  uint32_t p = 8, num = 1, np = 1, cgroups = 1, row_stride = 0, row_len = 0;
};
SlidePlan plan_slide(const FilterSpec &f, uint32_t channels);
size_t slide_lds_bytes(const SlidePlan &t, uint32_t waves);  // LDS of a workgroup of `waves` waves
void build_slide_rows(const FilterSpec &f, const SlidePlan &t, std::vector<float> *rows);
hipError_t launch_slide(const FilterSpec &f, const SlidePlan &t, const float *d_rows, uint32_t channels,
                        const StreamDesc *h_descs, const DescPack *pack,
                        uint32_t n_streams, bool float_io, hipStream_t stream, bool fixed_shape = false);

// ---- ... with an fp64 accumulator (kernels_slide64_impl.h, round 4): the reference's "double" kernels (quality 9
// and 10, resample.c:389-435, :501-558) on the same ratios.  The plan reuses SlidePlan: np = den (accumulators per
// period), cgroups = channels (one lane per channel of a lane block), pair_ch unused. ------------------------------
SlidePlan plan_slide64(const FilterSpec &f, uint32_t channels);
void build_slide64_rows(const FilterSpec &f, const SlidePlan &t, std::vector<double> *rows);
hipError_t launch_slide64(const FilterSpec &f, const SlidePlan &t, const double *d_rows, uint32_t channels,
                          const StreamDesc *h_descs, const DescPack *pack,
                          uint32_t n_streams, bool float_io, hipStream_t stream, bool fixed_shape = false);

}  // namespace speexhip
