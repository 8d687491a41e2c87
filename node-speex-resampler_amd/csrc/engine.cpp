// engine.cpp -- see engine.h.
#include "engine.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace speexhip {
namespace {
thread_local std::string g_last_error = "no HIP error recorded";

bool hip_failed(hipError_t e, const char *what) {
  if (e == hipSuccess) return false;
  g_last_error = std::string("HIP device error: ") + what + ": " + hipGetErrorString(e);
  return true;
}
#define HIP_TRY(expr)                                            \
  do {                                                           \
    if (hip_failed((expr), #expr)) return SPEEXHIP_ERR_DEVICE;   \
  } while (0)

const size_t kLdsBudget = 150 * 1024;  // of the CU's 160 KiB
}  // namespace

const char *last_device_error() { return g_last_error.c_str(); }

Batch *Batch::create(uint32_t n_streams, uint32_t channels, uint32_t in_rate, uint32_t out_rate,
                     int quality, int *err) {
  int e = SPEEXHIP_ERR_SUCCESS;
  Batch *b = nullptr;
  // argument checks first, like the reference (resample.c:804-809)
  if (n_streams == 0 || channels == 0 || in_rate == 0 || out_rate == 0 || quality > 10 ||
      quality < 0) {
    e = SPEEXHIP_ERR_INVALID_ARG;
  } else {
    b = new (std::nothrow) Batch();
    if (b == nullptr) {
      e = SPEEXHIP_ERR_ALLOC_FAILED;
    } else {
      b->n_streams_ = n_streams;
      b->channels_ = channels;
      e = design_filter(in_rate, out_rate, quality, &b->filter_);
      if (e == SPEEXHIP_ERR_SUCCESS) e = b->setup();
      if (e != SPEEXHIP_ERR_SUCCESS) {
        delete b;
        b = nullptr;
      }
    }
  }
  if (err) *err = e;
  return b;
}

int Batch::setup() {
  int count = 0;
  if (hip_failed(hipGetDeviceCount(&count), "hipGetDeviceCount") || count <= 0) {
    if (count <= 0) g_last_error = "HIP device error: no GPU visible (libspeexhip has no CPU fallback)";
    return SPEEXHIP_ERR_DEVICE;
  }
  HIP_TRY(hipGetDevice(&device_));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("HIP device error: built for gfx950 (MI355X), found ") + prop.gcnArchName;
    return SPEEXHIP_ERR_DEVICE;
  }
  const char *m = std::getenv("SPEEXHIP_MODE");
  if (m != nullptr && std::strcmp(m, "exact") == 0) mode_ = SPEEXHIP_MODE_EXACT;

  pos_.assign(n_streams_, StreamPos());
  HIP_TRY(hipMalloc(&d_table_, sizeof(float) * filter_.table_len));
  HIP_TRY(hipMemcpy(d_table_, filter_.table.data(), sizeof(float) * filter_.table_len,
                    hipMemcpyHostToDevice));
  hist_elems_ = static_cast<size_t>(filter_.taps - 1) * channels_;
  const size_t hist_bytes = std::max<size_t>(hist_elems_ * n_streams_ * sizeof(float), 16);
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipMalloc(&d_hist_[i], hist_bytes));
    HIP_TRY(hipMemset(d_hist_[i], 0, hist_bytes));  // resample.c:721-725: history starts silent
  }
  exact_geo_ = exact_geometry(filter_, channels_, kLdsBudget);
  period_ = plan_period(filter_, channels_, kLdsBudget);
  if (period_.usable) {
    std::vector<float> rows;
    build_period_rows(filter_, period_, &rows);
    HIP_TRY(hipMalloc(&d_period_rows_, rows.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(d_period_rows_, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  slide_ = plan_slide(filter_, channels_);
  if (slide_.usable && !period_.usable) {
    std::vector<float> rows;
    build_slide_rows(filter_, slide_, &rows);
    HIP_TRY(hipMalloc(&d_slide_rows_, rows.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(d_slide_rows_, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
  } else {
    slide_.usable = false;
  }
  if (n_streams_ > static_cast<uint32_t>(kMaxPackedStreams)) {
    const size_t ring_bytes = sizeof(StreamDesc) * n_streams_ * kRing;
    HIP_TRY(hipHostMalloc(&h_ring_, ring_bytes, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&d_ring_, ring_bytes));
    for (int i = 0; i < kRing; i++) HIP_TRY(hipEventCreateWithFlags(&ring_done_[i], hipEventDisableTiming));
  }
  HIP_TRY(hipDeviceSynchronize());
  return SPEEXHIP_ERR_SUCCESS;
}

Batch::~Batch() {
  if (own_stream_) (void)hipStreamSynchronize(own_stream_);
  (void)hipFree(d_table_);
  (void)hipFree(d_hist_[0]);
  (void)hipFree(d_hist_[1]);
  (void)hipFree(d_period_rows_);
  (void)hipFree(d_slide_rows_);
  (void)hipFree(d_ring_);
  if (h_ring_) (void)hipHostFree(h_ring_);
  for (int i = 0; i < kRing; i++)
    if (ring_done_[i]) (void)hipEventDestroy(ring_done_[i]);
  (void)hipFree(d_stage_in_);
  (void)hipFree(d_stage_out_);
  if (h_pin_in_) (void)hipHostFree(h_pin_in_);
  if (h_pin_out_) (void)hipHostFree(h_pin_out_);
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

int Batch::set_mode(int mode) {
  if (mode != SPEEXHIP_MODE_FAST && mode != SPEEXHIP_MODE_EXACT) return SPEEXHIP_ERR_INVALID_ARG;
  mode_ = mode;
  return SPEEXHIP_ERR_SUCCESS;
}

void Batch::info(uint32_t s, SpeexHipInfo *o) const {
  std::memset(o, 0, sizeof(*o));
  o->in_rate = filter_.in_rate;
  o->out_rate = filter_.out_rate;
  o->num_rate = filter_.num;
  o->den_rate = filter_.den;
  o->nb_channels = channels_;
  o->quality = filter_.quality;
  o->filt_len = filter_.taps;
  o->oversample = filter_.oversample;
  o->sinc_table_length = filter_.table_len;
  o->kernel = filter_.kind;
  o->mode = mode_;
  o->fast_path = period_.usable ? 2 : (slide_.usable ? 3 : 0);
  if (s < n_streams_) {
    o->last_sample = pos_[s].last;
    o->samp_frac_num = pos_[s].frac;
  }
  o->device = device_;
}

int Batch::history(uint32_t s, float *dst) {
  if (s >= n_streams_) return SPEEXHIP_ERR_INVALID_ARG;
  HIP_TRY(hipDeviceSynchronize());
  if (hist_elems_)
    HIP_TRY(hipMemcpy(dst, d_hist_[hist_cur_] + s * hist_elems_, hist_elems_ * sizeof(float),
                      hipMemcpyDeviceToHost));
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::process_device(const void *d_in, uint64_t in_stride, uint32_t *in_len, void *d_out,
                          uint64_t out_stride, uint32_t *out_len, bool float_io, hipStream_t stream) {
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  // the int16 entry point emits at most 1024 outputs per 160-frame block (its stack buffer,
  // resample.c:982-991); the float entry point has no such cap (resample.c:943)
  const uint32_t block_out = float_io ? 0xffffffffu : kBlockOut;
  const bool packed = n_streams_ <= static_cast<uint32_t>(kMaxPackedStreams);
  DescPack pack;
  StreamDesc *descs = pack.d;
  int slot = 0;
  if (packed) {
    std::memset(&pack, 0, sizeof(pack));
  } else {
    slot = ring_next_;
    ring_next_ = (ring_next_ + 1) % kRing;
    if (ring_busy_[slot]) {  // the launch that last used this slot must have read it
      HIP_TRY(hipEventSynchronize(ring_done_[slot]));
      ring_busy_[slot] = false;
    }
    descs = h_ring_ + static_cast<size_t>(slot) * n_streams_;
  }

  uint32_t max_out = 0;
  bool any_work = false;
  std::vector<CallPlan> plans(n_streams_);
  for (uint32_t s = 0; s < n_streams_; s++) {
    const CallPlan plan = plan_call(filter_.num, filter_.den, in_len[s], out_len[s], pos_[s], block_out);
    plans[s] = plan;
    StreamDesc &d = descs[s];
    d.in = d_in ? static_cast<const char *>(d_in) + s * in_stride * es : nullptr;
    d.hist = d_hist_[hist_cur_] + s * hist_elems_;
    d.out = static_cast<char *>(d_out) + s * out_stride * es;
    d.hist_next = d_hist_[hist_cur_ ^ 1] + s * hist_elems_;
    d.in_frames = in_len[s];
    d.n_out = plan.produced;
    d.consumed = plan.consumed;
    d.last0 = plan.begin.last;
    d.frac0 = plan.begin.frac;
    d.k_shift = phase_index_of(filter_.num, filter_.den, plan.begin.frac);
    d.base_shift = plan.begin.last -
                   static_cast<int32_t>((static_cast<uint64_t>(d.k_shift) * filter_.num) / filter_.den);
    d.tile_begin = 0;
    max_out = std::max(max_out, plan.produced);
    any_work = any_work || plan.produced != 0 || plan.consumed != 0;
  }

  if (any_work) {
    const StreamDesc *d_descs = nullptr;
    if (!packed) {
      StreamDesc *dst = d_ring_ + static_cast<size_t>(slot) * n_streams_;
      HIP_TRY(hipMemcpyAsync(dst, descs, sizeof(StreamDesc) * n_streams_, hipMemcpyHostToDevice, stream));
      d_descs = dst;
    }
    hipError_t e;
    if (mode_ == SPEEXHIP_MODE_FAST && period_.usable)
      e = launch_period(filter_, period_, d_period_rows_, channels_, descs, d_descs,
                        packed ? &pack : nullptr, n_streams_, float_io, stream);
    else if (mode_ == SPEEXHIP_MODE_FAST && slide_.usable)
      e = launch_slide(filter_, slide_, d_slide_rows_, channels_, descs, d_descs,
                       packed ? &pack : nullptr, n_streams_, float_io, stream);
    else
      e = launch_exact(filter_, exact_geo_, d_table_, channels_, d_descs, packed ? &pack : nullptr,
                       n_streams_, max_out, float_io, stream);
    if (hip_failed(e, "kernel launch")) return SPEEXHIP_ERR_DEVICE;
    if (!packed) {
      HIP_TRY(hipEventRecord(ring_done_[slot], stream));
      ring_busy_[slot] = true;
    }
    hist_cur_ ^= 1;
  }
  for (uint32_t s = 0; s < n_streams_; s++) {
    pos_[s] = plans[s].end;
    in_len[s] = plans[s].consumed;
    out_len[s] = plans[s].produced;
  }
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::ensure_stage(size_t in_bytes, size_t out_bytes) {
  if (own_stream_ == nullptr) HIP_TRY(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  if (in_bytes > stage_in_cap_) {  // grow-only, like the wrapper's heap buffers (src/index.ts:71-87)
    (void)hipFree(d_stage_in_);
    if (h_pin_in_) (void)hipHostFree(h_pin_in_);
    d_stage_in_ = nullptr;
    h_pin_in_ = nullptr;
    stage_in_cap_ = 0;
    const size_t cap = std::max<size_t>(in_bytes + in_bytes / 4, 8192);
    HIP_TRY(hipMalloc(&d_stage_in_, cap));
    HIP_TRY(hipHostMalloc(&h_pin_in_, cap, hipHostMallocDefault));
    stage_in_cap_ = cap;
  }
  if (out_bytes > stage_out_cap_) {
    (void)hipFree(d_stage_out_);
    if (h_pin_out_) (void)hipHostFree(h_pin_out_);
    d_stage_out_ = nullptr;
    h_pin_out_ = nullptr;
    stage_out_cap_ = 0;
    const size_t cap = std::max<size_t>(out_bytes + out_bytes / 4, 8192);
    HIP_TRY(hipMalloc(&d_stage_out_, cap));
    HIP_TRY(hipHostMalloc(&h_pin_out_, cap, hipHostMallocDefault));
    stage_out_cap_ = cap;
  }
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::process_host(const void *in, uint32_t *in_len, void *out, uint32_t *out_len, bool float_io) {
  if (n_streams_ != 1) return SPEEXHIP_ERR_BAD_STATE;
  HIP_TRY(hipSetDevice(device_));
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  const uint32_t frames = *in_len;
  // only as many output frames as this call can produce need a device buffer
  const uint32_t will_make =
      produced_closed_form(filter_.num, filter_.den, frames, *out_len, pos_[0]);
  const size_t in_bytes = static_cast<size_t>(frames) * channels_ * es;
  const size_t out_bytes = static_cast<size_t>(will_make) * channels_ * es;
  int rc = ensure_stage(in_bytes, out_bytes);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (in != nullptr && in_bytes != 0) {
    std::memcpy(h_pin_in_, in, in_bytes);
    HIP_TRY(hipMemcpyAsync(d_stage_in_, h_pin_in_, in_bytes, hipMemcpyHostToDevice, own_stream_));
  }
  rc = process_device(in != nullptr ? d_stage_in_ : nullptr, 0, in_len, d_stage_out_, 0, out_len, float_io,
                      own_stream_);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  const size_t made = static_cast<size_t>(*out_len) * channels_ * es;
  if (made != 0)
    HIP_TRY(hipMemcpyAsync(h_pin_out_, d_stage_out_, made, hipMemcpyDeviceToHost, own_stream_));
  HIP_TRY(hipStreamSynchronize(own_stream_));
  if (made != 0) std::memcpy(out, h_pin_out_, made);
  return SPEEXHIP_ERR_SUCCESS;
}

}  // namespace speexhip
