// engine.h -- host engine: owns device state for a batch of independent streams that share
// one filter, plans each call on the host and launches the HIP kernels (product code).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/speexhip_resampler.h"
#include "device_types.h"
#include "filter_design.h"
#include "kernels.h"
#include "stream_plan.h"

namespace speexhip {

const char *last_device_error();  // text of the most recent HIP failure on this thread
void set_last_device_error(const std::string &text);
size_t lds_budget();  // LDS the planners may give one workgroup (of the CU's 160 KiB)
// Test hook: the n-th next device allocation of a filter install fails (resample.c:785-791 path); 0 = off.
void debug_fail_device_allocs(int n);

// Everything on the device that depends only on (filter, channel count): the reference-layout sinc
// table, the fast kernels' tap rows and the launch geometry.  Immutable once built, so states with
// the same (num, den, quality, channels) share one copy (a cache in engine.cpp keeps the most recent
// ones alive between states): a second `new SpeexResampler(2, 44100, 48000, 7)` designs nothing and
// uploads nothing.
struct DeviceTables {
  int device = 0;
  float *table = nullptr;        // reference layout (exact kernel)
  float *period_rows = nullptr;  // kernels_period.hip, R = 10
  float *fine_rows = nullptr;    // ... R = 5 (launches of one generation)
  float *w16_rows = nullptr;     // ... with an int16 LDS window (wide windows; int16 calls)
  float *slide_rows = nullptr;   // kernels_slide.hip
  double *slide64_rows = nullptr;  // kernels_slide64_impl.h: fp64 taps (filters of the reference's double kinds)
  double *period64_rows = nullptr, *fine64_rows = nullptr;  // kernels_period64.hip: the same for the period kernel
  double *period64_w16_rows = nullptr;                      // ... over an int16 LDS window (kernels_period64_w16.hip)
  float *pp_rows = nullptr, *pp_w16_rows = nullptr;         // kernels_period_pp.hip: phase pairs (mono, wide windows)
  ExactGeometry geo, geo_ch;     // exact kernel, all channels / one channel per launch
  PeriodPlan period, fine, w16, period64, fine64, pp, pp_w16, period64_w16;
  SlidePlan slide, slide64;
  size_t bytes = 0;
  DeviceTables() = default;
  DeviceTables(const DeviceTables &) = delete;
  DeviceTables &operator=(const DeviceTables &) = delete;
  ~DeviceTables();
};
// speexhip_warmup: the runtime's and the pool's one-time start-up costs on `device` (< 0: every device the placement
// rule can choose), paid now instead of by the first states.
int warmup(int device);
// Idle cache entries back to the pool (speexhip_release_cached_memory); the bytes they held.
size_t release_cached_tables();

class Batch {
 public:
  // Returns nullptr and sets *err on failure.  device: logical ordinal (devices.h), or < 0 for the process-wide
  // placement rule (SPEEXHIP_DEVICE / SPEEXHIP_DEVICES, default = the calling thread's current HIP device).
  static Batch *create(uint32_t n_streams, uint32_t channels, uint32_t in_rate, uint32_t out_rate,
                       int quality, int *err, int device = -1);
  // speex_resampler_init_frac (resample.c:799): ratio given separately from the nominal rates.
  static Batch *create_frac(uint32_t n_streams, uint32_t channels, uint32_t ratio_num, uint32_t ratio_den,
                            uint32_t in_rate, uint32_t out_rate, int quality, int *err, int device = -1);
  ~Batch();

  // Device-resident call for all streams; asynchronous on `stream`.
  // float_io selects the sample type of in/out: int16 (process_interleaved_int) or float
  // (process_interleaved_float); strides are in samples of that type.
  int process_device(const void *d_in, uint64_t in_stride, uint32_t *in_len, void *d_out,
                     uint64_t out_stride, uint32_t *out_len, bool float_io, hipStream_t stream);
  // Host-buffer call for a single-stream batch; synchronous (H2D, kernels, D2H).
  int process_host(const void *in, uint32_t *in_len, void *out, uint32_t *out_len, bool float_io);
  // The same call with the result left in a pinned block of the pool that the caller then OWNS (release_block):
  // the kernel writes the block straight through PCIe, so the samples cross memory once on their way out instead
  // of twice (device or pinned buffer -> copy -> the caller's buffer).  *block = nullptr when nothing was written.
  int process_host_take(const void *in, uint32_t *in_len, uint32_t *out_len, bool float_io, void **block);
  static void release_block(void *block);
  // Host-buffer calls of n single-stream states at once (SURVEY 8b "array of states/buffers"; the reference's model
  // is many instances in one process, src/index.ts:18-45): per GPU one transfer in, one launch per <= 32 states that
  // share a filter, one transfer out; states on different GPUs run side by side (a GPU is a PCIe link of its own).
  // Samples and counters of state i are exactly those of process_host(in[i], &in_len[i], out[i], &out_len[i]);
  // codes[i] (may be null) = that call's return code; returns the first code that is not SUCCESS.
  static int process_host_many(uint32_t n, Batch *const *states, const void *const *in, uint32_t *in_len,
                               void *const *out, uint32_t *out_len, bool float_io, int *codes);
  int device() const { return device_; }
  // n_chunks consecutive host-buffer calls of a single-stream batch as one launch; outputs are
  // written back to back into `out` (room for the sum of the capacities).
  int process_host_chunks(uint32_t n_chunks, const void *const *in, uint32_t *in_len, void *out,
                          uint32_t *out_len, bool float_io);
  // One channel of a single-stream batch through host buffers with the state's input / output
  // strides: speex_resampler_process_int / _process_float (resample.c:927-1036).  Channels
  // advance independently, as in the reference (per-channel last_sample / samp_frac_num /
  // magic_samples, resample.c:135-137).
  int process_channel_host(uint32_t channel, const void *in, uint32_t *in_len, void *out, uint32_t *out_len,
                           bool float_io);
  void set_strides(uint32_t in_stride, uint32_t out_stride, bool set_in, bool set_out) {
    if (set_in) in_stride_ = in_stride;
    if (set_out) out_stride_ = out_stride;
  }
  uint32_t in_stride() const { return in_stride_; }
  uint32_t out_stride() const { return out_stride_; }
  // position of one channel (resample.c last_sample / samp_frac_num / magic_samples)
  StreamPos channel_pos(uint32_t s, uint32_t c) const { return pos_[static_cast<size_t>(s) * channels_ + c]; }
  bool zero_mode() const { return zero_mode_; }

  // Mid-stream control (reference resample.c:1084-1220).  These wait for the
  // device, re-align every stream's history on the host (resample.c:727-782) and rebuild the
  // filter tables; they apply to all streams of the batch.
  int set_rate_frac(uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate);
  int set_quality(int quality);
  int skip_zeros();
  int reset_mem();
  int input_latency() const { return static_cast<int>(filter_.taps / 2); }
  int output_latency() const {
    return static_cast<int>(((filter_.taps / 2) * filter_.den + (filter_.num >> 1)) / filter_.num);
  }

  // Counters of the next call for stream s, state untouched.
  CallPlan peek(uint32_t s, uint32_t in_len, uint32_t out_capacity, bool float_io) const;

  int set_mode(int mode);
  // The caller is about to destroy the stream of the state's last device-pointer call: order what is still in
  // flight on it behind an event of the state's own and forget the stream.
  int release_stream();
  void info(uint32_t stream, SpeexHipInfo *out) const;
  int history(uint32_t stream, float *dst);
  const FilterSpec &filter() const { return filter_; }
  // What speex_resampler_get_rate / get_ratio report: the filter's rates, except after a set_rate_frac that
  // returned RESAMPLER_ERR_OVERFLOW -- the reference has stored the new ones by then (resample.c:1119-1127).
  struct RateView {
    uint32_t in_rate, out_rate, num, den;
  };
  RateView rates() const {
    return shown_valid_ ? shown_ : RateView{filter_.in_rate, filter_.out_rate, filter_.num, filter_.den};
  }
  uint32_t n_streams() const { return n_streams_; }
  uint32_t channels() const { return channels_; }

 private:
  Batch() = default;
  int setup();
  int install_filter(const FilterSpec &f, const std::vector<float> &hist, uint32_t hist_frames_cap);
  int adopt_filter(const FilterSpec &next);
  int change_filter(const FilterSpec &next, int design_rc, const std::vector<uint32_t> *fracs);
  void enter_zero_mode(const FilterSpec &partly_designed);
  StreamPos &P(uint32_t s, uint32_t c) { return pos_[static_cast<size_t>(s) * channels_ + c]; }
  const StreamPos &P(uint32_t s, uint32_t c) const { return pos_[static_cast<size_t>(s) * channels_ + c]; }
  bool uniform(uint32_t s) const;  // all channels of stream s at the same position
  uint32_t max_magic(uint32_t s) const;
  int run_channel(uint32_t c, const void *d_in, uint32_t in_stride, uint32_t in_frames, void *d_out,
                  uint32_t out_stride, const CallPlan &plan, bool float_io, hipStream_t stream);
  int process_split(const void *d_in, uint32_t *in_len, void *d_out, uint32_t *out_len, bool float_io,
                    hipStream_t stream, std::vector<CallPlan> *plans_out);
  int fetch_history(std::vector<float> *host);
  int quiesce();  // waits for this batch's own enqueued work (never for the whole device)
  uint32_t block_in() const { return line_ - (filter_.taps - 1); }
  int ensure_stage(size_t dev_in, size_t dev_out, size_t pin_in, size_t pin_out);
  int run_plans(const void *d_in, uint64_t in_stride, const uint32_t *in_frames, void *d_out,
                uint64_t out_stride, const CallPlan *plans, bool float_io, hipStream_t stream);
  int launch_chunk(const StreamDesc *descs, const DescPack &pack, uint32_t n, uint32_t max_out, bool float_io,
                   hipStream_t stream);
  // a large host call as pieces: input copies on a second stream, one launch per piece behind each (process_host_take)
  int take_in_pieces(const void *in, uint32_t *in_len, uint32_t *out_len, bool float_io, void *blk, uint32_t pieces);
  static int many_on_device(int device, int lane, const std::vector<uint32_t> &idx, Batch *const *st, const void *const *in,
                            uint32_t *in_len, void *const *out, uint32_t *out_len, bool float_io, int *rcs);
  bool have_copy_stream();  // copy_stream_ = a pool stream other than own_stream_ (false: none -> no piecewise call)

  FilterSpec filter_;
  uint32_t n_streams_ = 0, channels_ = 0;
  int device_ = 0;
  bool counted_ = false;  // this state is in its device's live count (devices::state_born)
  int mode_ = SPEEXHIP_MODE_FAST_FIXED;  // (round 6: the default is the mode whose bytes depend on the stream alone)
  std::vector<StreamPos> pos_;    // [stream][channel] (the reference keeps them per channel, resample.c:135-137;
                                  // interleaved calls move all channels of a stream together)
  bool zero_mode_ = false;        // resampler_ptr == resampler_basic_zero (resample.c:785-791): the last
                                  // filter change failed; outputs are zeros until one succeeds
  uint32_t in_stride_ = 1, out_stride_ = 1;  // resample.c:842-843, 1170-1188 (per-channel entry points)
  RateView shown_ = {0, 0, 0, 0};  // see rates()
  bool shown_valid_ = false;
  ExactGeometry exact_geo_ch_;    // the exact kernel's geometry for one-channel launches
  std::vector<uint8_t> started_;  // per stream: a block has run (resample.c:886), so a filter
                                  // change must re-align the history instead of clearing it
  uint32_t line_ = 0;             // frames per channel line, grow-only (resample.c
                                  // "mem_alloc_size", :709-720): block size = line_-(taps-1)

  std::shared_ptr<const DeviceTables> tables_;  // owns the four table pointers below
  float *d_table_ = nullptr;
  float *d_hist_[2] = {nullptr, nullptr};  // float, like the reference's `mem`
  size_t hist_elems_ = 0;  // per stream: (taps-1 + room for pending frames)*channels
  size_t hist_bytes_ = 0;  // allocation of each history buffer (>= the copy-engine minimum, engine.cpp kCtlCopyMin)
  int hist_cur_ = 0;

  ExactGeometry exact_geo_;
  PeriodPlan period_;      // primary fast path (kernels_period.hip)
  float *d_period_rows_ = nullptr;
  PeriodPlan period_fine_;  // the same filter with 5 phases per wave: single-generation launches
  float *d_period_fine_rows_ = nullptr;
  PeriodPlan period_w16_;   // the same filter over an int16 LDS window (usable only where it pays)
  float *d_period_w16_rows_ = nullptr;
  bool float_seen_ = false;  // a float call has put samples into the histories that an int16 window cannot hold
  void int16_call_done(const CallPlan *plans, uint32_t n);  // ... until int16 calls have replaced all of them (round 6)
  SlidePlan slide_;        // small-ratio fast path (kernels_slide.hip); neither usable -> exact
  float *d_slide_rows_ = nullptr;
  SlidePlan slide64_;      // ... with an fp64 accumulator: what FAST runs for the double kinds (quality 9, 10)
  double *d_slide64_rows_ = nullptr;
  PeriodPlan period_pp_, period_pp_w16_;  // phase-pair plans (mono, wide windows): chosen per launch
  float *d_period_pp_rows_ = nullptr, *d_period_pp_w16_rows_ = nullptr;
  PeriodPlan period64_, period64_fine_;  // the period kernel's plans with an fp64 accumulator (kernels_period64.hip)
  double *d_period64_rows_ = nullptr, *d_period64_fine_rows_ = nullptr;
  PeriodPlan period64_w16_;              // ... over an int16 LDS window (wide windows; int16 calls)
  double *d_period64_w16_rows_ = nullptr;
  bool acc64() const {     // the fast path sums in fp64 (mode FAST on a filter the reference sums in fp64)
    return (mode_ == SPEEXHIP_MODE_FAST || mode_ == SPEEXHIP_MODE_FAST_FIXED) &&
           (filter_.kind == kDirectDouble || filter_.kind == kInterpolateDouble);
  }

  // Calls on one batch are chained: a call on another stream than the previous one waits for it on the device
  // (an event recorded on the previous stream at that moment), control calls and the destructor wait for the
  // previous stream on the host.  So the stream of a device-pointer call must outlive the state's next call --
  // unless the caller hands it back with release_stream(): the event is recorded then and the stream forgotten.
  // (Round 4 tried an event of the batch's own recorded behind EVERY device-pointer launch, so that nothing would
  // ever be asked of a caller's stream after its call: +3.0 us per launch on this stack -- BASELINE configs[1]
  // 11.7 -> 14.7 us per step, profiles/r04_ab_done_event.txt.  And a stale handle cannot be recognised after the
  // fact: this runtime dereferences it -- hipStreamSynchronize / hipEventRecord on a destroyed stream segfault,
  // tools/probe_stream_gone.hip.)
  hipStream_t last_stream_ = nullptr;
  bool have_last_stream_ = false;
  hipEvent_t order_ev_ = nullptr;
  bool ev_pending_ = false;          // order_ev_ stands for the batch's last work (release_stream)
  int chain_to(hipStream_t stream);  // before a launch on `stream`

  // host-buffer path (single stream)
  hipStream_t own_stream_ = nullptr;
  hipStream_t copy_stream_ = nullptr;   // input copies of a piecewise call (another of the pool's streams)
  static const int kMaxPieces = 8;
  hipEvent_t piece_ev_[kMaxPieces] = {};
  char *d_stage_in_ = nullptr, *d_stage_out_ = nullptr;
  char *h_pin_in_ = nullptr, *h_pin_out_ = nullptr;
  uint32_t done_seq_ = 0;  // completion word of the small host-buffer calls (engine.cpp, process_host)
  size_t stage_in_cap_ = 0, stage_out_cap_ = 0, pin_in_cap_ = 0, pin_out_cap_ = 0;  // bytes
};

}  // namespace speexhip
