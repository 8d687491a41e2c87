// c_api.cpp -- the extern "C" boundary declared in include/speexhip_resampler.h.
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/speexhip_resampler.h"
#include "devices.h"
#include "engine.h"
#include "pool.h"

using speexhip::Batch;

namespace {
// No C++ exception may cross the C boundary: a host allocation that fails inside a call (descriptor
// vectors, the history image of a filter change, a multi-gigabyte sinc table) becomes the
// reference's RESAMPLER_ERR_ALLOC_FAILED, anything else the device error code with its text.
template <typename F>
int guarded(F &&f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    return SPEEXHIP_ERR_ALLOC_FAILED;
  } catch (const std::exception &e) {
    speexhip::set_last_device_error(std::string("internal error: ") + e.what());
    return SPEEXHIP_ERR_DEVICE;
  } catch (...) {
    speexhip::set_last_device_error("internal error: unknown exception");
    return SPEEXHIP_ERR_DEVICE;
  }
}
}  // namespace

struct SpeexHipResamplerState_ {
  Batch *batch;
};
struct SpeexHipBatch_ {
  Batch *batch;
};

extern "C" {

int speexhip_device_count(void) { return speexhip::devices::count(); }
int speexhip_warmup(int device) {
  return guarded([&] { return speexhip::warmup(device); });
}
int speexhip_debug_placement(int device_count, const char *env_device, const char *env_devices, uint64_t k,
                             int current_device) {
  return speexhip::devices::placement_rule(device_count, env_device, env_devices, k, current_device);
}

int speexhip_debug_placement_live(int device_count, const char *env_device, const char *env_devices, uint64_t k,
                                  int current_device, const uint32_t *live) {
  return speexhip::devices::placement_rule_live(device_count, env_device, env_devices, k, current_device, live);
}
uint32_t speexhip_debug_live_states(int device) { return speexhip::devices::live_states(device); }

SpeexHipResamplerState *speexhip_resampler_init(uint32_t nb_channels, uint32_t in_rate,
                                                uint32_t out_rate, int quality, int *err) {
  return speexhip_resampler_init_on(-1, nb_channels, in_rate, out_rate, quality, err);
}

SpeexHipResamplerState *speexhip_resampler_init_on(int device, uint32_t nb_channels, uint32_t in_rate,
                                                   uint32_t out_rate, int quality, int *err) {
  Batch *b = nullptr;
  int code = SPEEXHIP_ERR_SUCCESS;
  const int rc = guarded([&] {
    b = Batch::create(1, nb_channels, in_rate, out_rate, quality, &code, device < 0 ? -1 : device);
    return code;
  });
  if (err) *err = rc;
  if (b == nullptr) return nullptr;
  SpeexHipResamplerState *st = new (std::nothrow) SpeexHipResamplerState_{b};
  if (st == nullptr) {
    delete b;
    if (err) *err = SPEEXHIP_ERR_ALLOC_FAILED;
  }
  return st;
}

SpeexHipResamplerState *speexhip_resampler_init_frac(uint32_t nb_channels, uint32_t ratio_num,
                                                     uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate,
                                                     int quality, int *err) {
  Batch *b = nullptr;
  int code = SPEEXHIP_ERR_SUCCESS;
  const int rc = guarded([&] {
    b = Batch::create_frac(1, nb_channels, ratio_num, ratio_den, in_rate, out_rate, quality, &code);
    return code;
  });
  if (err) *err = rc;
  if (b == nullptr) return nullptr;
  SpeexHipResamplerState *st = new (std::nothrow) SpeexHipResamplerState_{b};
  if (st == nullptr) {
    delete b;
    if (err) *err = SPEEXHIP_ERR_ALLOC_FAILED;
  }
  return st;
}

int speexhip_resampler_set_rate(SpeexHipResamplerState *st, uint32_t in_rate, uint32_t out_rate) {
  return guarded([&] { return st ? st->batch->set_rate_frac(in_rate, out_rate, in_rate, out_rate) : SPEEXHIP_ERR_INVALID_ARG; });
}
int speexhip_resampler_set_rate_frac(SpeexHipResamplerState *st, uint32_t ratio_num, uint32_t ratio_den,
                                     uint32_t in_rate, uint32_t out_rate) {
  return guarded([&] { return st ? st->batch->set_rate_frac(ratio_num, ratio_den, in_rate, out_rate) : SPEEXHIP_ERR_INVALID_ARG; });
}
void speexhip_resampler_get_ratio(SpeexHipResamplerState *st, uint32_t *ratio_num, uint32_t *ratio_den) {
  *ratio_num = st->batch->rates().num;
  *ratio_den = st->batch->rates().den;
}
int speexhip_resampler_set_quality(SpeexHipResamplerState *st, int quality) {
  return guarded([&] { return st ? st->batch->set_quality(quality) : SPEEXHIP_ERR_INVALID_ARG; });
}
void speexhip_resampler_get_quality(SpeexHipResamplerState *st, int *quality) {
  *quality = st->batch->filter().quality;
}
int speexhip_resampler_get_input_latency(SpeexHipResamplerState *st) { return st->batch->input_latency(); }
int speexhip_resampler_get_output_latency(SpeexHipResamplerState *st) { return st->batch->output_latency(); }
int speexhip_resampler_skip_zeros(SpeexHipResamplerState *st) {
  return guarded([&] { return st ? st->batch->skip_zeros() : SPEEXHIP_ERR_INVALID_ARG; });
}
int speexhip_resampler_reset_mem(SpeexHipResamplerState *st) {
  return guarded([&] { return st ? st->batch->reset_mem() : SPEEXHIP_ERR_INVALID_ARG; });
}

int speexhip_batch_set_rate_frac(SpeexHipBatch *b, uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate,
                                 uint32_t out_rate) {
  return guarded([&] { return b ? b->batch->set_rate_frac(ratio_num, ratio_den, in_rate, out_rate) : SPEEXHIP_ERR_INVALID_ARG; });
}
int speexhip_batch_set_quality(SpeexHipBatch *b, int quality) {
  return guarded([&] { return b ? b->batch->set_quality(quality) : SPEEXHIP_ERR_INVALID_ARG; });
}
int speexhip_batch_skip_zeros(SpeexHipBatch *b) { return guarded([&] { return b ? b->batch->skip_zeros() : SPEEXHIP_ERR_INVALID_ARG; }); }
int speexhip_batch_reset_mem(SpeexHipBatch *b) { return guarded([&] { return b ? b->batch->reset_mem() : SPEEXHIP_ERR_INVALID_ARG; }); }
int speexhip_batch_get_history(SpeexHipBatch *b, uint32_t stream, float *dst) {
  if (b == nullptr || dst == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return b->batch->history(stream, dst); });
}

void speexhip_resampler_destroy(SpeexHipResamplerState *st) {
  if (st == nullptr) return;
  delete st->batch;
  delete st;
}

int speexhip_resampler_process_interleaved_int(SpeexHipResamplerState *st, const int16_t *in,
                                               uint32_t *in_len, int16_t *out, uint32_t *out_len) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr || (out == nullptr && *out_len != 0))
    return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_host(in, in_len, out, out_len, false); });
}

int speexhip_resampler_process_interleaved_int_device(SpeexHipResamplerState *st, const int16_t *d_in,
                                                      uint32_t *in_len, int16_t *d_out,
                                                      uint32_t *out_len, void *hip_stream) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_device(d_in, 0, in_len, d_out, 0, out_len, false,
                                   static_cast<hipStream_t>(hip_stream)); });
}

int speexhip_resampler_process_interleaved_int_take(SpeexHipResamplerState *st, const int16_t *in, uint32_t *in_len,
                                                    uint32_t *out_len, int16_t **out_block) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr || out_block == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_host_take(in, in_len, out_len, false, reinterpret_cast<void **>(out_block)); });
}

int speexhip_resampler_process_interleaved_float_take(SpeexHipResamplerState *st, const float *in, uint32_t *in_len,
                                                      uint32_t *out_len, float **out_block) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr || out_block == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_host_take(in, in_len, out_len, true, reinterpret_cast<void **>(out_block)); });
}

void speexhip_block_release(void *block) {
  if (block != nullptr) speexhip::Batch::release_block(block);
}

void *speexhip_block_acquire(uint64_t bytes) {
  void *p = nullptr;
  const int rc = guarded([&] {
    return speexhip::pool::block_get(&p, static_cast<size_t>(bytes)) ? SPEEXHIP_ERR_SUCCESS : SPEEXHIP_ERR_NO_BLOCK;
  });
  return rc == SPEEXHIP_ERR_SUCCESS ? p : nullptr;
}

int speexhip_resampler_process_interleaved_float(SpeexHipResamplerState *st, const float *in,
                                                 uint32_t *in_len, float *out, uint32_t *out_len) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr || (out == nullptr && *out_len != 0))
    return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_host(in, in_len, out, out_len, true); });
}

int speexhip_resampler_process_interleaved_float_device(SpeexHipResamplerState *st, const float *d_in,
                                                        uint32_t *in_len, float *d_out, uint32_t *out_len,
                                                        void *hip_stream) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_device(d_in, 0, in_len, d_out, 0, out_len, true,
                                   static_cast<hipStream_t>(hip_stream)); });
}

int speexhip_batch_process_interleaved_float_device(SpeexHipBatch *b, const float *d_in,
                                                    uint64_t in_stream_stride, uint32_t *in_len, float *d_out,
                                                    uint64_t out_stream_stride, uint32_t *out_len,
                                                    void *hip_stream) {
  if (b == nullptr || in_len == nullptr || out_len == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return b->batch->process_device(d_in, in_stream_stride, in_len, d_out, out_stream_stride, out_len, true,
                                  static_cast<hipStream_t>(hip_stream)); });
}

int speexhip_resampler_process_chunks_int(SpeexHipResamplerState *st, uint32_t n_chunks,
                                          const int16_t *const *in, uint32_t *in_len, int16_t *out,
                                          uint32_t *out_len) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_host_chunks(n_chunks, reinterpret_cast<const void *const *>(in), in_len, out,
                                        out_len, false); });
}

int speexhip_resampler_process_chunks_float(SpeexHipResamplerState *st, uint32_t n_chunks,
                                            const float *const *in, uint32_t *in_len, float *out,
                                            uint32_t *out_len) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_host_chunks(n_chunks, reinterpret_cast<const void *const *>(in), in_len, out,
                                        out_len, true); });
}

int speexhip_resampler_process_int(SpeexHipResamplerState *st, uint32_t channel_index, const int16_t *in,
                                   uint32_t *in_len, int16_t *out, uint32_t *out_len) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr || (out == nullptr && *out_len != 0))
    return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_channel_host(channel_index, in, in_len, out, out_len, false); });
}

int speexhip_resampler_process_float(SpeexHipResamplerState *st, uint32_t channel_index, const float *in,
                                     uint32_t *in_len, float *out, uint32_t *out_len) {
  if (st == nullptr || in_len == nullptr || out_len == nullptr || (out == nullptr && *out_len != 0))
    return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->process_channel_host(channel_index, in, in_len, out, out_len, true); });
}

void speexhip_resampler_set_input_stride(SpeexHipResamplerState *st, uint32_t stride) {
  st->batch->set_strides(stride, 0, true, false);
}
void speexhip_resampler_get_input_stride(SpeexHipResamplerState *st, uint32_t *stride) {
  *stride = st->batch->in_stride();
}
void speexhip_resampler_set_output_stride(SpeexHipResamplerState *st, uint32_t stride) {
  st->batch->set_strides(0, stride, false, true);
}
void speexhip_resampler_get_output_stride(SpeexHipResamplerState *st, uint32_t *stride) {
  *stride = st->batch->out_stride();
}

int speexhip_resampler_get_channel_position(SpeexHipResamplerState *st, uint32_t channel, int32_t *last_sample,
                                            uint32_t *samp_frac_num, uint32_t *magic_samples) {
  if (st == nullptr || channel >= st->batch->channels()) return SPEEXHIP_ERR_INVALID_ARG;
  const speexhip::StreamPos p = st->batch->channel_pos(0, channel);
  if (last_sample) *last_sample = p.last;
  if (samp_frac_num) *samp_frac_num = p.frac;
  if (magic_samples) *magic_samples = p.magic;
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_debug_plan(uint32_t ratio_num, uint32_t ratio_den, int quality, uint32_t channels, uint32_t out[8]) {
  if (out == nullptr || channels == 0) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] {
    speexhip::FilterSpec f;
    const int rc = speexhip::design_filter_frac(ratio_num, ratio_den, ratio_num, ratio_den, quality, &f, false);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    std::memset(out, 0, 8 * sizeof(uint32_t));
    // (round 6: small-denominator ratios outside the slide kernel's shapes plan the period kernel on a folded view)
    speexhip::FilterSpec view;
    const speexhip::FilterSpec &real = f;
    const speexhip::FilterSpec &pf = speexhip::period_view(real, channels, &view) ? view : real;
    const speexhip::PeriodPlan t = speexhip::plan_period(pf, channels, speexhip::lds_budget());
    const speexhip::SlidePlan sl = speexhip::plan_slide(real, channels);
    if (t.usable) {
      out[0] = 2;
      out[1] = t.r;
      out[2] = t.lane_periods;
      out[3] = t.row_len;
      out[4] = static_cast<uint32_t>(t.window_bytes);
      out[5] = t.pad;
      if (t.r == 10) {
        const speexhip::PeriodPlan fine = speexhip::plan_period_r(pf, channels, speexhip::lds_budget(), 5);
        out[6] = fine.usable && fine.float_ok && fine.lane_periods == t.lane_periods;
      }
      const speexhip::PeriodPlan w16 = speexhip::plan_period_w16(pf, channels, speexhip::lds_budget(), t);
      out[7] = w16.usable ? w16.lane_periods : 0;
    } else if (sl.usable) {
      out[0] = 3;
      out[1] = sl.p;
      out[3] = sl.row_len;
      out[4] = static_cast<uint32_t>(speexhip::slide_lds_bytes(sl, 2));
      out[7] = sl.p * sl.num;
    }
    return static_cast<int>(SPEEXHIP_ERR_SUCCESS);
  });
}
int speexhip_debug_launch_shape(uint32_t ratio_num, uint32_t ratio_den, int quality, uint32_t channels, uint32_t streams,
                                uint32_t frames, int float_io, uint32_t out[10]) {
  if (out == nullptr || channels == 0 || streams == 0 || streams > 32) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&]() -> int {
    speexhip::FilterSpec f;
    const int rc = speexhip::design_filter_frac(ratio_num, ratio_den, ratio_num, ratio_den, quality, &f, false);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    std::memset(out, 0, 10 * sizeof(uint32_t));
    if (f.kind == speexhip::kDirectDouble || f.kind == speexhip::kInterpolateDouble) return SPEEXHIP_ERR_SUCCESS;
    // the plans a stream state holds (engine.cpp, build_tables) and the choice of launch_chunk among them, for a
    // first call of `frames` frames on every stream (an r = 5 companion plan, where one exists, is not modelled)
    const size_t lds = speexhip::lds_budget();
    speexhip::FilterSpec view;
    const speexhip::FilterSpec &pf = speexhip::period_view(f, channels, &view) ? view : f;
    const speexhip::PeriodPlan base = speexhip::plan_period(pf, channels, lds);
    if (!base.usable) return SPEEXHIP_ERR_SUCCESS;
    if (float_io != 0 && !base.float_ok) return SPEEXHIP_ERR_SUCCESS;  // (a plan that stands for its int16 plan alone: float calls run the exact kernel)
    const speexhip::PeriodPlan w16 = speexhip::plan_period_w16(pf, channels, lds, base);
    speexhip::PeriodPlan pp, pp_w16;
    if (speexhip::period_wants_pp_plans(pf, channels)) {
      pp = speexhip::plan_period(pf, channels, lds, false, false, true);
      pp_w16 = speexhip::plan_period_w16(pf, channels, lds, pp);
    }
    std::vector<speexhip::StreamDesc> descs(streams);
    for (auto &d : descs) {
      std::memset(&d, 0, sizeof(d));
      d.in_frames = frames;
      d.n_out = static_cast<uint32_t>(static_cast<uint64_t>(frames) * f.den / f.num);
    }
    const bool i16 = !float_io;
    const speexhip::PeriodPlan &two = (i16 && w16.usable) ? w16 : base;
    const speexhip::PeriodPlan &pairs = (i16 && pp_w16.usable) ? pp_w16 : pp;
    const speexhip::PeriodPlan *t = &base;
    if (pp.usable && speexhip::period_launch_prefers_pp(f, two, pairs, descs.data(), streams)) {
      t = &pairs;
      out[0] = 1;
    } else if (i16 && w16.usable && speexhip::period_launch_prefers_w16(f, base, false, descs.data(), streams)) {
      t = &w16;
    }
    out[1] = t->r;
    out[2] = t->w16 ? 1u : 0u;
    out[3] = t->lane_periods;
    if (!speexhip::debug_period_shape(f, *t, channels, descs.data(), streams, float_io != 0, out + 4)) return SPEEXHIP_ERR_BAD_STATE;
    return SPEEXHIP_ERR_SUCCESS;
  });
}
int speexhip_debug_plan64(uint32_t ratio_num, uint32_t ratio_den, int quality, uint32_t channels, uint32_t out[8]) {
  if (out == nullptr || channels == 0) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] {
    speexhip::FilterSpec f;
    const int rc = speexhip::design_filter_frac(ratio_num, ratio_den, ratio_num, ratio_den, quality, &f, false);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    std::memset(out, 0, 8 * sizeof(uint32_t));
    const bool double_kind = f.kind == speexhip::kDirectDouble || f.kind == speexhip::kInterpolateDouble;
    speexhip::FilterSpec view;
    const speexhip::FilterSpec &pf = speexhip::period_view(f, channels, &view) ? view : f;
    const speexhip::PeriodPlan base = speexhip::plan_period(pf, channels, speexhip::lds_budget());
    if (double_kind && base.usable) {
      const speexhip::PeriodPlan t = speexhip::plan_period(pf, channels, speexhip::lds_budget(), false, true);
      if (t.usable) {
        out[0] = 5;
        out[1] = t.r;
        out[2] = t.lane_periods;
        out[3] = t.row_len;
        out[4] = static_cast<uint32_t>(t.window_bytes);
        out[5] = t.pad;
        out[6] = t.l4;
        const speexhip::PeriodPlan w16 = speexhip::plan_period_w16(pf, channels, speexhip::lds_budget(), t);  // (round 5)
        out[7] = w16.usable ? w16.lane_periods : 0;
      }
    } else if (double_kind && speexhip::plan_slide(f, channels).usable) {
      const speexhip::SlidePlan sl = speexhip::plan_slide64(f, channels);
      if (sl.usable) {
        out[0] = 4;
        out[1] = sl.p;
        out[3] = sl.row_len;
        out[4] = static_cast<uint32_t>(speexhip::slide_lds_bytes(sl, 2) * (sl.p * f.den >= 4 ? 2 : 1));  // (image in doubles)
        out[5] = sl.row_stride;
        out[7] = sl.p * sl.num;
      }
    }
    // phase-pair plans of mono filters with wide windows (any quality below 9)
    if (!double_kind && base.usable && speexhip::period_wants_pp_plans(pf, channels)) {
      const speexhip::PeriodPlan t = speexhip::plan_period(pf, channels, speexhip::lds_budget(), false, false, true);
      if (t.usable) {
        out[0] = 6;
        out[1] = t.r;
        out[2] = t.lane_periods;
        out[3] = t.row_len;
        out[4] = static_cast<uint32_t>(t.window_bytes);
        out[5] = t.pad;
        const speexhip::PeriodPlan w16 = speexhip::plan_period_w16(pf, channels, speexhip::lds_budget(), t);
        out[7] = w16.usable ? w16.lane_periods : 0;
      }
    }
    return static_cast<int>(SPEEXHIP_ERR_SUCCESS);
  });
}
void speexhip_debug_fail_device_allocs(int n) { speexhip::debug_fail_device_allocs(n); }
uint64_t speexhip_release_cached_memory(void) {
  const uint64_t tables = speexhip::release_cached_tables();  // first: they return their buffers to the pool
  (void)tables;
  return speexhip::pool::release_idle();
}

void speexhip_resampler_get_rate(SpeexHipResamplerState *st, uint32_t *in_rate, uint32_t *out_rate) {
  *in_rate = st->batch->rates().in_rate;
  *out_rate = st->batch->rates().out_rate;
}

const char *speexhip_resampler_strerror(int err) {
  switch (err) {  // reference resample.c:1222-1239
    case SPEEXHIP_ERR_SUCCESS: return "Success.";
    case SPEEXHIP_ERR_ALLOC_FAILED: return "Memory allocation failed.";
    case SPEEXHIP_ERR_BAD_STATE: return "Bad resampler state.";
    case SPEEXHIP_ERR_INVALID_ARG: return "Invalid argument.";
    case SPEEXHIP_ERR_PTR_OVERLAP: return "Input and output buffers overlap.";
    case SPEEXHIP_ERR_DEVICE: return speexhip::last_device_error();
    case SPEEXHIP_ERR_NO_BLOCK: return "No pinned result block available (state untouched).";
    default: return "Unknown error. Bad error code or strange version mismatch.";
  }
}

int speexhip_resampler_peek(SpeexHipResamplerState *st, uint32_t in_len, uint32_t out_capacity, int float_entry,
                            uint32_t *consumed, uint32_t *produced) {
  if (st == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  const speexhip::CallPlan plan = st->batch->peek(0, in_len, out_capacity, float_entry != 0);
  if (consumed) *consumed = plan.consumed;
  if (produced) *produced = plan.produced;
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_resampler_set_mode(SpeexHipResamplerState *st, int mode) {
  return guarded([&] { return st ? st->batch->set_mode(mode) : SPEEXHIP_ERR_INVALID_ARG; });
}

int speexhip_resampler_release_stream(SpeexHipResamplerState *st) {
  return guarded([&] { return st ? st->batch->release_stream() : SPEEXHIP_ERR_INVALID_ARG; });
}

int speexhip_resampler_get_info(SpeexHipResamplerState *st, SpeexHipInfo *info) {
  if (st == nullptr || info == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  st->batch->info(0, info);
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_resampler_get_info2(SpeexHipResamplerState *st, SpeexHipInfo *info, uint32_t struct_size) {
  if (st == nullptr || info == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  SpeexHipInfo full;
  st->batch->info(0, &full);
  std::memcpy(info, &full, struct_size < sizeof(full) ? struct_size : sizeof(full));
  return SPEEXHIP_ERR_SUCCESS;
}

namespace {
int process_many(uint32_t n, SpeexHipResamplerState *const *st, const void *const *in, uint32_t *in_len,
                 void *const *out, uint32_t *out_len, int *codes, bool float_io) {
  if (n == 0) return SPEEXHIP_ERR_SUCCESS;
  if (st == nullptr || in == nullptr || in_len == nullptr || out == nullptr || out_len == nullptr)
    return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] {
    std::vector<Batch *> b(n);
    for (uint32_t i = 0; i < n; i++) b[i] = st[i] != nullptr ? st[i]->batch : nullptr;
    return Batch::process_host_many(n, b.data(), in, in_len, out, out_len, float_io, codes);
  });
}
}  // namespace

int speexhip_resampler_process_many_int(uint32_t n, SpeexHipResamplerState *const *st, const int16_t *const *in,
                                        uint32_t *in_len, int16_t *const *out, uint32_t *out_len, int *codes) {
  return process_many(n, st, reinterpret_cast<const void *const *>(in), in_len, reinterpret_cast<void *const *>(out),
                      out_len, codes, false);
}
int speexhip_resampler_process_many_float(uint32_t n, SpeexHipResamplerState *const *st, const float *const *in,
                                          uint32_t *in_len, float *const *out, uint32_t *out_len, int *codes) {
  return process_many(n, st, reinterpret_cast<const void *const *>(in), in_len, reinterpret_cast<void *const *>(out),
                      out_len, codes, true);
}

int speexhip_resampler_get_history(SpeexHipResamplerState *st, float *dst) {
  if (st == nullptr || dst == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return st->batch->history(0, dst); });
}

SpeexHipBatch *speexhip_batch_init(uint32_t n_streams, uint32_t nb_channels, uint32_t in_rate,
                                   uint32_t out_rate, int quality, int *err) {
  return speexhip_batch_init_on(-1, n_streams, nb_channels, in_rate, out_rate, quality, err);
}

SpeexHipBatch *speexhip_batch_init_on(int device, uint32_t n_streams, uint32_t nb_channels, uint32_t in_rate,
                                      uint32_t out_rate, int quality, int *err) {
  Batch *b = nullptr;
  int code = SPEEXHIP_ERR_SUCCESS;
  const int rc = guarded([&] {
    b = Batch::create(n_streams, nb_channels, in_rate, out_rate, quality, &code, device < 0 ? -1 : device);
    return code;
  });
  if (err) *err = rc;
  if (b == nullptr) return nullptr;
  SpeexHipBatch *h = new (std::nothrow) SpeexHipBatch_{b};
  if (h == nullptr) {
    delete b;
    if (err) *err = SPEEXHIP_ERR_ALLOC_FAILED;
  }
  return h;
}

void speexhip_batch_destroy(SpeexHipBatch *b) {
  if (b == nullptr) return;
  delete b->batch;
  delete b;
}

int speexhip_batch_set_mode(SpeexHipBatch *b, int mode) {
  return guarded([&] { return b ? b->batch->set_mode(mode) : SPEEXHIP_ERR_INVALID_ARG; });
}

int speexhip_batch_release_stream(SpeexHipBatch *b) {
  return guarded([&] { return b ? b->batch->release_stream() : SPEEXHIP_ERR_INVALID_ARG; });
}

int speexhip_batch_get_info(SpeexHipBatch *b, uint32_t stream, SpeexHipInfo *info) {
  if (b == nullptr || info == nullptr || stream >= b->batch->n_streams()) return SPEEXHIP_ERR_INVALID_ARG;
  b->batch->info(stream, info);
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_batch_process_interleaved_int_device(SpeexHipBatch *b, const int16_t *d_in,
                                                  uint64_t in_stream_stride, uint32_t *in_len,
                                                  int16_t *d_out, uint64_t out_stream_stride,
                                                  uint32_t *out_len, void *hip_stream) {
  if (b == nullptr || in_len == nullptr || out_len == nullptr) return SPEEXHIP_ERR_INVALID_ARG;
  return guarded([&] { return b->batch->process_device(d_in, in_stream_stride, in_len, d_out, out_stream_stride, out_len, false,
                                  static_cast<hipStream_t>(hip_stream)); });
}

int speexhip_design_filter(uint32_t in_rate, uint32_t out_rate, int quality, SpeexHipInfo *info,
                           float *table, uint32_t table_capacity) {
  if (in_rate == 0 || out_rate == 0) return SPEEXHIP_ERR_INVALID_ARG;
  return speexhip_design_filter_frac(in_rate, out_rate, in_rate, out_rate, quality, info, table, table_capacity);
}

int speexhip_design_filter_frac(uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate,
                                int quality, SpeexHipInfo *info, float *table, uint32_t table_capacity) {
  speexhip::FilterSpec f;
  const int rc = guarded([&] {
    return speexhip::design_filter_frac(ratio_num, ratio_den, in_rate, out_rate, quality, &f, table != nullptr);
  });
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (info != nullptr) {
    std::memset(info, 0, sizeof(*info));
    info->in_rate = f.in_rate;
    info->out_rate = f.out_rate;
    info->num_rate = f.num;
    info->den_rate = f.den;
    info->quality = f.quality;
    info->filt_len = f.taps;
    info->oversample = f.oversample;
    info->sinc_table_length = f.table_len;
    info->kernel = f.kind;
    info->device = -1;
  }
  if (table != nullptr) {
    const uint32_t n = f.table_len < table_capacity ? f.table_len : table_capacity;
    std::memcpy(table, f.table.data(), sizeof(float) * n);
  }
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_plan_call(uint32_t num_rate, uint32_t den_rate, uint32_t in_len, uint32_t out_cap,
                       int32_t *last_sample, uint32_t *samp_frac_num, uint32_t *consumed,
                       uint32_t *produced) {
  if (num_rate == 0 || den_rate == 0 || last_sample == nullptr || samp_frac_num == nullptr)
    return SPEEXHIP_ERR_INVALID_ARG;
  speexhip::StreamPos p;
  p.last = *last_sample;
  p.frac = *samp_frac_num;
  const speexhip::CallPlan plan = speexhip::plan_call(num_rate, den_rate, in_len, out_cap, p);
  *last_sample = plan.end.last;
  *samp_frac_num = plan.end.frac;
  if (consumed) *consumed = plan.consumed;
  if (produced) *produced = plan.produced;
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_plan_call_ex(uint32_t num_rate, uint32_t den_rate, uint32_t in_len, uint32_t out_cap,
                          int float_entry, uint32_t block_in, int32_t *last_sample, uint32_t *samp_frac_num,
                          uint32_t *magic_samples, uint32_t *consumed, uint32_t *produced) {
  if (num_rate == 0 || den_rate == 0 || block_in == 0 || last_sample == nullptr ||
      samp_frac_num == nullptr || magic_samples == nullptr)
    return SPEEXHIP_ERR_INVALID_ARG;
  speexhip::StreamPos p;
  p.last = *last_sample;
  p.frac = *samp_frac_num;
  p.magic = *magic_samples;
  speexhip::EntryRules rules;
  rules.block_in = block_in;
  rules.float_entry = float_entry != 0;
  const speexhip::CallPlan plan = speexhip::plan_call(num_rate, den_rate, in_len, out_cap, p, rules);
  *last_sample = plan.end.last;
  *samp_frac_num = plan.end.frac;
  *magic_samples = plan.end.magic;
  if (consumed) *consumed = plan.consumed;
  if (produced) *produced = plan.produced;
  return SPEEXHIP_ERR_SUCCESS;
}

int speexhip_plan_filter_change(uint32_t old_filt_len, uint32_t new_filt_len, uint32_t magic, int64_t *shift,
                                uint32_t *new_magic, int32_t *last_delta, uint32_t *phase, uint32_t old_den,
                                uint32_t new_den) {
  if (old_filt_len == 0 || new_filt_len == 0 || shift == nullptr || new_magic == nullptr ||
      last_delta == nullptr)
    return SPEEXHIP_ERR_INVALID_ARG;
  if (phase != nullptr) {
    if (old_den == 0 || new_den == 0) return SPEEXHIP_ERR_INVALID_ARG;
    if (!speexhip::scale_phase(phase, new_den, old_den)) return SPEEXHIP_ERR_OVERFLOW;
  }
  const speexhip::Realign r = speexhip::realign_history(old_filt_len, new_filt_len, magic);
  *shift = r.shift;
  *new_magic = r.new_magic;
  *last_delta = r.last_delta;
  return SPEEXHIP_ERR_SUCCESS;
}

const char *speexhip_version(void) { return "speexhip 0.4.0 gfx950"; }

}  // extern "C"
