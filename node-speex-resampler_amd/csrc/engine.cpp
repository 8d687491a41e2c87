// engine.cpp -- see engine.h.
#include "engine.h"

#include "devices.h"
#include "pool.h"
#include "unit_workers.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

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

// Every entry point runs on the batch's own device whatever the calling thread's current one is,
// and leaves the thread's current device as it found it.
class DeviceScope {
 public:
  explicit DeviceScope(int logical_device) {  // (devices.h: logical ordinals; physical = logical unless aliased)
    const int device = devices::physical(logical_device);
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = device;
    if (prev_ != device) err_ = hipSetDevice(device);
    // (a failed call leaves its code behind as the thread's "last error", and the launchers read that after their
    //  next launch: hipLaunchKernelGGL + hipGetLastError would report THIS failure for a launch that worked)
    if (err_ != hipSuccess) (void)hipGetLastError();
  }
  ~DeviceScope() {
    if (prev_ != device_now()) (void)hipSetDevice(prev_);
  }
  hipError_t error() const { return err_; }

 private:
  static int device_now() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
  }
  int prev_ = 0;
  hipError_t err_ = hipSuccess;
};
#define ON_DEVICE()                 \
  DeviceScope device_scope(device_); \
  HIP_TRY(device_scope.error())

std::atomic<int> g_fail_allocs{0};

// Device allocation of the filter installs: ALLOC_FAILED when the device is out of memory (or the
// test hook says so) -- the caller then falls back to resampler_basic_zero like the reference
// (resample.c:785-791) --, DEVICE for anything else.
int dev_alloc(int device, void **ptr, size_t bytes) {
  *ptr = nullptr;
  // test hook: a countdown -- the allocation that brings it to zero fails
  if (g_fail_allocs.load() > 0 && g_fail_allocs.fetch_sub(1) == 1) return SPEEXHIP_ERR_ALLOC_FAILED;
  hipError_t e = pool::device_get(device, ptr, bytes);
  if (e == hipErrorOutOfMemory) {
    // (pool::device_get has already given the pool's idle buffers back and tried again.)  Table sets
    // that no state references any more are reclaimable too: evict them, empty the pool, try once more.
    (void)hipGetLastError();
    if (release_cached_tables() != 0) {
      (void)pool::release_idle();
      e = pool::device_get(device, ptr, bytes);
    }
  }
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    return SPEEXHIP_ERR_ALLOC_FAILED;
  }
  if (hip_failed(e, "hipMalloc")) return SPEEXHIP_ERR_DEVICE;
  return SPEEXHIP_ERR_SUCCESS;
}

// The host-buffer calls are synchronous, and their staging buffers and the (shared) stream go on to the
// next call or to another state: on EVERY exit nothing they enqueued may still be in flight.  The normal
// path waits explicitly (and checks the result); this guard covers the early error returns.
struct DrainOnExit {
  hipStream_t *stream;  // (pointer: the stream is taken from the pool after the guard is set up)
  bool armed = true;
  explicit DrainOnExit(hipStream_t *s) : stream(s) {}
  ~DrainOnExit() {
    if (armed && *stream != nullptr) (void)hipStreamSynchronize(*stream);
  }
};

// Control-plane copies (histories of a filter change, table uploads).  The runtime performs copies of
// <= 16 KiB with a blit KERNEL, and a kernel of this stream is dispatched only once the kernels other
// states have running let go of the CUs: 2 KB beside a 5 ms launch took 5 ms, 16 385 bytes 9 us
// (tools/ubench_ctl.hip, profiles/r03_ubench_ctl.txt).  So every control copy is at least kCtlCopyMin bytes
// -- the device buffers involved are allocated at least that big -- and runs on the copy engines from / to a
// pinned image, beside whatever the device is doing.
const size_t kCtlCopyMin = 20 * 1024;
struct PinnedImage {
  char *p = nullptr;
  size_t bytes = 0;
  ~PinnedImage() { pool::pinned_put(p); }
  hipError_t get(size_t want) {
    bytes = std::max(want, kCtlCopyMin);
    return pool::pinned_get(reinterpret_cast<void **>(&p), bytes);
  }
};
// host `src` (bytes) -> device `dst` (capacity >= max(bytes, kCtlCopyMin)); waits for the copy
int ctl_upload(void *dst, const void *src, size_t bytes, hipStream_t stream) {
  PinnedImage img;
  HIP_TRY(img.get(bytes));
  std::memset(img.p, 0, img.bytes);
  if (bytes != 0) std::memcpy(img.p, src, bytes);
  HIP_TRY(hipMemcpyAsync(dst, img.p, img.bytes, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return SPEEXHIP_ERR_SUCCESS;
}
// device buffer `base` of `total` bytes (>= kCtlCopyMin): bytes [off, off+len) -> host `dst`; waits
int ctl_download(void *dst, const char *base, size_t total, size_t off, size_t len, hipStream_t stream) {
  if (len == 0) return SPEEXHIP_ERR_SUCCESS;
  const size_t span = std::min(total, std::max(len, kCtlCopyMin));
  const size_t first = std::min(off, total - span);
  PinnedImage img;
  HIP_TRY(img.get(span));
  HIP_TRY(hipMemcpyAsync(img.p, base + first, span, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  std::memcpy(dst, img.p + (off - first), len);
  return SPEEXHIP_ERR_SUCCESS;
}

const size_t kLdsBudget = 150 * 1024;  // of the CU's 160 KiB
const size_t kDirectCopyBytes = 256 * 1024;  // host buffers at least this big skip the pinned bounce buffer
// A piecewise host call (take_in_pieces) takes 2 MB of input per piece, at most four: a piece costs ~15 us (an event,
// a cross-stream wait, a launch) and buys the overlap of its launch -- the result leaving through PCIe -- with the next
// piece's copy.  2^20 stereo frames (4.2 MB in): 1 / 2 / 3 / 4 pieces 0.1935 / 0.1879 / 0.198 / 0.224 ms per call; 8
// channels (16.8 MB in), four pieces: 0.617 -> 0.517 ms.
const size_t kPieceBytes = static_cast<size_t>(2) << 20;
const size_t kZeroCopyBelow = 720 * 1024;    // ... and calls whose buffers are smaller than this run on pinned memory alone

inline bool buffers_overlap(const void *a, size_t na, const void *b, size_t nb) {
  const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
  return x < y + nb && y < x + na;
}

// Round 6 -- pinned host buffers are used in place.  Where the kernels reach a HOST buffer directly: the address the
// device sees when all of [p, p + bytes) is pinned memory -- a block of the library's slabs (speexhip_block_acquire, a
// result block of ..._take: a range check, no runtime call), or, for buffers of kDirectCopyBytes and more, memory the
// caller pinned itself (hipHostMalloc / hipHostRegister: asked of the runtime at both ends of the buffer) -- else nullptr.
// Such a buffer needs no staging: the kernel that reads it through PCIe can write its result block through PCIe at the
// same time (the link is full duplex), where copy -> launch -> copy takes the two crossings one after the other; for the
// Node wrapper this replaces the copy into the module's heap, src/index.ts:71-92.
void *pinned_view(const void *p, size_t bytes) {
  if (p == nullptr || bytes == 0) return nullptr;
  if (pool::block_owns(p, bytes)) return const_cast<void *>(p);
  if (bytes < kDirectCopyBytes) return nullptr;
  auto device_side = [](const void *q) -> char * {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) {
      (void)hipGetLastError();  // (an ordinary malloc'ed pointer is an "invalid value" to older runtimes)
      return nullptr;
    }
    return a.type == hipMemoryTypeHost ? static_cast<char *>(a.devicePointer) : nullptr;
  };
  char *first = device_side(p);
  if (first == nullptr) return nullptr;
  char *last = device_side(static_cast<const char *>(p) + bytes - 1);
  return last != nullptr && static_cast<size_t>(last - first) == bytes - 1 ? first : nullptr;
}

// The wait of a host-buffer call whose kernels read and write pinned memory: hipStreamSynchronize after a small launch
// costs 11-12 us on this stack; a 32-bit stream write behind the kernel (hipStreamWriteValue32: performed once
// everything before it on the stream has completed) into pinned memory, polled by the caller, 8.8 (tools/ubench_sync.hip).
// The pool's streams are shared by all states -- behind another state's long launch the word will not come soon -- so the
// spin is bounded (`spin_us`, `pause` between the reads) and then the thread sleeps in the runtime, which also reports
// the error if that is what happened.  *word must not equal seq when the call is made.
int wait_done(hipStream_t stream, volatile uint32_t *word, uint32_t seq, uint32_t spin_us) {
  static const bool poll_done = SPEEXHIP_DIAG_ENV("SPEEXHIP_NO_POLL") == nullptr;  // (A/B)
  bool signalled = false;
  if (poll_done && hipStreamWriteValue32(stream, const_cast<uint32_t *>(word), seq, 0) == hipSuccess) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
    for (uint32_t spins = 0; !signalled; spins++) {
      signalled = __atomic_load_n(const_cast<const uint32_t *>(word), __ATOMIC_ACQUIRE) == seq;
      if (signalled) break;
      __builtin_ia32_pause();
      if ((spins & 63u) == 63u && std::chrono::steady_clock::now() > deadline) break;
    }
  } else {
    (void)hipGetLastError();
  }
  if (!signalled) HIP_TRY(hipStreamSynchronize(stream));
  return SPEEXHIP_ERR_SUCCESS;
}
// A/B (diagnostics build): a large pinned input through the copy engines in pieces (take_in_pieces) instead of read in place
bool pinned_in_pieces() { return diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_PINNED_IN_PIECES"), 0) != 0; }
// how long such a call may spin: 300 us for the launch itself plus what `bytes` take to cross PCIe (~40 GB/s), 2 ms at most
uint32_t spin_budget_us(size_t bytes) { return static_cast<uint32_t>(std::min<size_t>(2000, 300 + bytes / 40000)); }
}  // namespace

// SPEEXHIP_INIT_TRACE=1: where a state's creation goes, step by step (stderr; tools/first_call.py)
struct InitTrace {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  static bool on() {
    static const bool v = SPEEXHIP_DIAG_ENV("SPEEXHIP_INIT_TRACE") != nullptr;
    return v;
  }
  void step(const char *what) {
    if (!on()) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "speexhip init: %-40s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

const char *last_device_error() { return g_last_error.c_str(); }
void set_last_device_error(const std::string &text) { g_last_error = text; }
size_t lds_budget() { return kLdsBudget; }

// diagnostics / tests: the int16-window plan at every launch size (normally only launches that fill the chip)
// (SPEEXHIP_W16_ALWAYS=0: never -- A/B runs)
static int w16_env() {
  static const int v = [] {
    const char *e = SPEEXHIP_DIAG_ENV("SPEEXHIP_W16_ALWAYS");
    return e == nullptr ? -1 : (e[0] == '0' && e[1] == '\0' ? 0 : 1);
  }();
  return v;
}
static bool w16_always() { return w16_env() == 1; }
static bool w16_never() { return w16_env() == 0; }
void debug_fail_device_allocs(int n) { g_fail_allocs.store(n < 0 ? 0 : n); }

bool Batch::uniform(uint32_t s) const {
  const StreamPos &a = P(s, 0);
  for (uint32_t c = 1; c < channels_; c++) {
    const StreamPos &b = P(s, c);
    if (a.last != b.last || a.frac != b.frac || a.magic != b.magic) return false;
  }
  return true;
}

uint32_t Batch::max_magic(uint32_t s) const {
  uint32_t m = 0;
  for (uint32_t c = 0; c < channels_; c++) m = std::max(m, P(s, c).magic);
  return m;
}

Batch *Batch::create(uint32_t n_streams, uint32_t channels, uint32_t in_rate, uint32_t out_rate,
                     int quality, int *err, int device) {
  return create_frac(n_streams, channels, in_rate, out_rate, in_rate, out_rate, quality, err, device);
}

Batch *Batch::create_frac(uint32_t n_streams, uint32_t channels, uint32_t ratio_num, uint32_t ratio_den,
                          uint32_t in_rate, uint32_t out_rate, int quality, int *err, int device) {
  int e = SPEEXHIP_ERR_SUCCESS;
  // (owned until it is handed out: an exception below -- a bad_alloc in a position vector, in the design
  //  or in the tap rows; c_api.cpp maps it to a code -- must not leak the batch and what it already took)
  std::unique_ptr<Batch> b;
  // argument checks first, like the reference (resample.c:804-809)
  if (n_streams == 0 || channels == 0 || ratio_num == 0 || ratio_den == 0 || quality > 10 ||
      quality < 0) {
    e = SPEEXHIP_ERR_INVALID_ARG;
  } else {
    b.reset(new (std::nothrow) Batch());
    if (b == nullptr) {
      e = SPEEXHIP_ERR_ALLOC_FAILED;
    } else {
      b->n_streams_ = n_streams;
      b->channels_ = channels;
      b->device_ = device;  // (< 0: the placement rule decides, setup())
      e = design_filter_frac(ratio_num, ratio_den, in_rate, out_rate, quality, &b->filter_, /*fill_table=*/false);
      if (e == SPEEXHIP_ERR_SUCCESS) e = b->setup();
      if (e != SPEEXHIP_ERR_SUCCESS) b.reset();
    }
  }
  if (err) *err = e;
  return b.release();
}

int Batch::setup() {
  InitTrace trace;
  const int count = devices::count();
  trace.step("device count (first HIP call)");
  if (count <= 0) {
    g_last_error = count < 0 ? "HIP device error: hipGetDeviceCount failed (libspeexhip has no CPU fallback)"
                             : "HIP device error: no GPU visible (libspeexhip has no CPU fallback)";
    return SPEEXHIP_ERR_DEVICE;
  }
  // Which GPU: the caller's choice (..._init_on), else the process-wide rule (devices.h: SPEEXHIP_DEVICE,
  // SPEEXHIP_DEVICES=all -> state k on device k mod count, default = the thread's current device).
  if (device_ < 0) {
    device_ = devices::place_next_state();
    if (device_ < 0) {
      g_last_error = "HIP device error: SPEEXHIP_DEVICE / SPEEXHIP_DEVICES names a device this node does not have";
      return SPEEXHIP_ERR_DEVICE;
    }
  } else if (device_ >= count) {
    g_last_error = "HIP device error: device " + std::to_string(device_) + " requested, the node has " + std::to_string(count);
    device_ = -1;  // (nothing lives anywhere yet: the destructor has no device to visit)
    return SPEEXHIP_ERR_DEVICE;
  }
  ON_DEVICE();  // (everything below -- pool, tables, uploads -- runs on the state's device)
  hipDeviceProp_t prop;
  trace.step("placement + hipSetDevice");
  HIP_TRY(hipGetDeviceProperties(&prop, devices::physical(device_)));
  trace.step("hipGetDeviceProperties");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("HIP device error: built for gfx950 (MI355X), found ") + prop.gcnArchName;
    return SPEEXHIP_ERR_DEVICE;
  }
  // (default: SPEEXHIP_MODE_FAST_FIXED -- a stream's bytes do not depend on chunking, batch size or GPU, like the
  //  reference's; 'fast' opts into launch-time re-association where it buys speed)
  const char *m = std::getenv("SPEEXHIP_MODE");
  if (m != nullptr && std::strcmp(m, "fast") == 0) mode_ = SPEEXHIP_MODE_FAST;
  if (m != nullptr && std::strcmp(m, "exact") == 0) mode_ = SPEEXHIP_MODE_EXACT;
  if (m != nullptr && std::strcmp(m, "fast_f32") == 0) mode_ = SPEEXHIP_MODE_FAST_F32;
  if (m != nullptr && std::strcmp(m, "fast_fixed") == 0) mode_ = SPEEXHIP_MODE_FAST_FIXED;

  pos_.assign(static_cast<size_t>(n_streams_) * channels_, StreamPos());
  started_.assign(n_streams_, 0);
  const FilterSpec designed = filter_;
  int rc = install_filter(designed, std::vector<float>(), designed.taps - 1);  // resample.c:721-725: silence
  trace.step("install_filter (total)");
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  devices::state_born(device_);  // (the live count the placement rule balances: devices.h, round 6)
  counted_ = true;
  return SPEEXHIP_ERR_SUCCESS;  // (install_filter waited for its own uploads)
}

DeviceTables::~DeviceTables() {
  // (whoever dropped the last reference has synchronised: ~Batch, install_filter)
  pool::device_put(device, table);
  pool::device_put(device, period_rows);
  pool::device_put(device, fine_rows);
  pool::device_put(device, w16_rows);
  pool::device_put(device, slide_rows);
  pool::device_put(device, slide64_rows);
  pool::device_put(device, pp_rows);
  pool::device_put(device, pp_w16_rows);
  pool::device_put(device, period64_rows);
  pool::device_put(device, fine64_rows);
  pool::device_put(device, period64_w16_rows);
}

namespace {
struct TablesKey {
  int device;
  uint32_t num, den, channels;
  int quality;
  bool operator==(const TablesKey &o) const {
    return device == o.device && num == o.num && den == o.den && channels == o.channels && quality == o.quality;
  }
};
struct TablesCache {
  std::mutex mu;
  std::list<std::pair<TablesKey, std::shared_ptr<const DeviceTables>>> lru;  // most recent first
};
TablesCache &tables_cache() {
  static TablesCache *c = new TablesCache();  // never destroyed, like the pool
  return *c;
}
const size_t kCachedTables = 24;                           // entries kept beyond the ones in use ...
const size_t kCachedTableBytes = static_cast<size_t>(256) << 20;  // ... and their bytes (most recent first)
const size_t kCacheableBytes = static_cast<size_t>(64) << 20;  // bigger table sets are built per state

// Design (host, double precision), plan and upload the tables of filter `g` (geometry only: num,
// den, quality, taps ...) for `channels` channels.
// Uploads run on `stream` (the installing state's control stream) and are waited for before the tables
// are published: never on the null stream, whose copies would wait for every blocking stream of the process.
int build_tables(int device, const FilterSpec &g, uint32_t channels, hipStream_t stream,
                 std::shared_ptr<const DeviceTables> *out) {
  InitTrace trace;
  FilterSpec f;
  int rc = design_filter_frac(g.num, g.den, g.in_rate, g.out_rate, g.quality, &f);
  trace.step("    filter design (host)");
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  auto t = std::make_shared<DeviceTables>();
  t->device = device;
  auto upload_bytes = [&](void **dst, const void *src, size_t bytes) -> int {
    const int arc = dev_alloc(device, dst, std::max(bytes, kCtlCopyMin));
    if (arc != SPEEXHIP_ERR_SUCCESS) return arc;
    if (bytes != 0) {
      const int urc = ctl_upload(*dst, src, bytes, stream);
      if (urc != SPEEXHIP_ERR_SUCCESS) return urc;
    }
    t->bytes += std::max(bytes, kCtlCopyMin);
    return SPEEXHIP_ERR_SUCCESS;
  };
  auto upload = [&](float **dst, const float *src, size_t count) -> int {
    return upload_bytes(reinterpret_cast<void **>(dst), src, count * sizeof(float));
  };
  rc = upload(&t->table, f.table.data(), f.table_len);
  trace.step("    first allocation + upload (sinc table)");
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  t->geo = exact_geometry(f, channels, kLdsBudget);
  t->geo_ch = exact_geometry(f, 1, kLdsBudget);
  // (ratios with den <= 6 outside the slide kernel's shapes -- 7:6, 11:1, 16:3 ... -- plan the period kernel on a folded
  //  view of the filter, 35:30, 110:10, 80:15: kernels.h, period_view; `pf` is what every period plan below is made on)
  FilterSpec folded;
  const bool use_fold = period_view(f, channels, &folded);
  const FilterSpec &pf = use_fold ? folded : f;
  t->period = plan_period(pf, channels, kLdsBudget);
  if (t->period.usable && t->period.float_ok) {
    std::vector<float> rows;
    build_period_rows(pf, t->period, &rows);
    rc = upload(&t->period_rows, rows.data(), rows.size());
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  if (t->period.usable && t->period.float_ok && t->period.r == 10) {
    static const bool no_fine = SPEEXHIP_DIAG_ENV("SPEEXHIP_NO_FINE") != nullptr;  // diagnostics: A/B
    t->fine = plan_period_r(pf, channels, kLdsBudget, 5);
    // (... and a float window of its own: an R = 5 plan that only stands for its int16 plan -- 100 channels of 320:147, one
    //  period per tile either way -- has nothing to launch; found by the fuzzer the day the layouts without an ISA loop
    //  got int16 plans, seed 611002591)
    if (no_fine || !t->fine.float_ok || t->fine.lane_periods != t->period.lane_periods) t->fine.usable = false;
    if (t->fine.usable) {
      std::vector<float> rows;
      build_period_rows(pf, t->fine, &rows);
      rc = upload(&t->fine_rows, rows.data(), rows.size());
      if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    }
  }
  t->w16 = plan_period_w16(pf, channels, kLdsBudget, t->period);
  if (t->w16.usable) {
    std::vector<float> rows;
    build_period_rows(pf, t->w16, &rows);
    rc = upload(&t->w16_rows, rows.data(), rows.size());
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  if (t->period.usable && period_wants_pp_plans(pf, channels)) {
    t->pp = plan_period(pf, channels, kLdsBudget, false, false, true);
    if (t->pp.usable) {
      std::vector<float> rows;
      build_period_rows(pf, t->pp, &rows);
      rc = upload(&t->pp_rows, rows.data(), rows.size());
      if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
      t->pp_w16 = plan_period_w16(pf, channels, kLdsBudget, t->pp);
      if (t->pp_w16.usable) {
        build_period_rows(pf, t->pp_w16, &rows);
        rc = upload(&t->pp_w16_rows, rows.data(), rows.size());
        if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
      }
    }
  }
  t->slide = plan_slide(f, channels);
  if (t->slide.usable && !t->period.usable) {
    std::vector<float> rows;
    build_slide_rows(f, t->slide, &rows);
    rc = upload(&t->slide_rows, rows.data(), rows.size());
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  } else {
    t->slide.usable = false;
  }
  // the reference's double kinds (quality 9, 10): fp64-accumulate twins of the fast kernels
  if (f.kind == kDirectDouble || f.kind == kInterpolateDouble) {
    if (t->period.usable) {
      t->period64 = plan_period(pf, channels, kLdsBudget, false, true);
      if (t->period64.usable) {
        std::vector<double> rows;
        build_period_rows64(pf, t->period64, &rows);
        rc = upload_bytes(reinterpret_cast<void **>(&t->period64_rows), rows.data(), rows.size() * sizeof(double));
        if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
      }
      if (t->period64.usable && t->period64.r == 10) {
        static const bool no_fine64 = SPEEXHIP_DIAG_ENV("SPEEXHIP_NO_FINE") != nullptr;  // diagnostics: A/B
        t->fine64 = plan_period_r(pf, channels, kLdsBudget, 5, false, true);
        if (no_fine64 || t->fine64.lane_periods != t->period64.lane_periods) t->fine64.usable = false;
        if (t->fine64.usable) {
          std::vector<double> rows;
          build_period_rows64(pf, t->fine64, &rows);
          rc = upload_bytes(reinterpret_cast<void **>(&t->fine64_rows), rows.data(), rows.size() * sizeof(double));
          if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
        }
      }
    }
    if (t->period64.usable) {
      t->period64_w16 = plan_period_w16(pf, channels, kLdsBudget, t->period64);
      if (t->period64_w16.usable) {
        std::vector<double> rows;
        build_period_rows64(pf, t->period64_w16, &rows);
        rc = upload_bytes(reinterpret_cast<void **>(&t->period64_w16_rows), rows.data(), rows.size() * sizeof(double));
        if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
      }
    }
    if (t->slide.usable) t->slide64 = plan_slide64(f, channels);
    if (t->slide64.usable) {
      std::vector<double> rows;
      build_slide64_rows(f, t->slide64, &rows);
      rc = upload_bytes(reinterpret_cast<void **>(&t->slide64_rows), rows.data(), rows.size() * sizeof(double));
      if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    }
  }
  trace.step("    plans, tap rows, their uploads");
  *out = t;
  return SPEEXHIP_ERR_SUCCESS;
}

int acquire_tables(int device, const FilterSpec &g, uint32_t channels, hipStream_t stream,
                   std::shared_ptr<const DeviceTables> *out) {
  static const bool enabled = [] {
    const char *e = std::getenv("SPEEXHIP_POOL_MB");
    return e == nullptr || std::atoi(e) != 0;
  }();
  // (an armed allocation-failure hook must see the allocations: no cache then)
  const bool cacheable = enabled && g_fail_allocs.load() <= 0;
  const TablesKey key{device, g.num, g.den, channels, g.quality};
  TablesCache &c = tables_cache();
  if (cacheable) {
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto it = c.lru.begin(); it != c.lru.end(); ++it)
      if (it->first == key) {
        c.lru.splice(c.lru.begin(), c.lru, it);
        *out = it->second;
        return SPEEXHIP_ERR_SUCCESS;
      }
  }
  std::shared_ptr<const DeviceTables> built;
  const int rc = build_tables(device, g, channels, stream, &built);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (cacheable && built->bytes <= kCacheableBytes) {
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto &e : c.lru)
      if (e.first == key) {  // another thread built the same tables meanwhile: share those
        *out = e.second;
        return SPEEXHIP_ERR_SUCCESS;
      }
    c.lru.emplace_front(key, built);
    size_t idle = 0, idle_bytes = 0;
    for (auto it = c.lru.begin(); it != c.lru.end();) {
      if (it->second.use_count() == 1 &&
          (++idle > kCachedTables || (idle_bytes += it->second->bytes) > kCachedTableBytes))
        it = c.lru.erase(it);  // nobody uses them: back to the pool
      else
        ++it;
    }
  }
  *out = built;
  return SPEEXHIP_ERR_SUCCESS;
}
}  // namespace

size_t release_cached_tables() {
  TablesCache &c = tables_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  size_t bytes = 0;
  for (auto it = c.lru.begin(); it != c.lru.end();) {
    if (it->second.use_count() == 1) {
      bytes += it->second->bytes;
      it = c.lru.erase(it);
    } else {
      ++it;
    }
  }
  return bytes;
}

// Put everything on the device that depends on the filter `f` (geometry: the tables are designed
// here on a cache miss) in place -- the shared tables, the history buffers (hist = all streams'
// lines, hist_frames_cap frames each, interleaved; empty = silence) -- and only then replace what
// the batch holds: a failed allocation leaves the batch exactly as it was (the caller decides what
// a failure means, resample.c:785-791).
int Batch::install_filter(const FilterSpec &f, const std::vector<float> &hist, uint32_t hist_frames_cap) {
  InitTrace trace;
  if (own_stream_ == nullptr) HIP_TRY(pool::stream_get(device_, &own_stream_));
  trace.step("  stream");
  std::shared_ptr<const DeviceTables> tables;
  int rc = acquire_tables(device_, f, channels_, own_stream_, &tables);
  trace.step("  tables (design, allocations, uploads)");
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  struct Fresh {
    int device = 0;
    float *hist[2] = {nullptr, nullptr};
    ~Fresh() {
      pool::device_put(device, hist[0]);
      pool::device_put(device, hist[1]);
    }
  } n;
  n.device = device_;
  const size_t hist_elems = static_cast<size_t>(hist_frames_cap) * channels_;
  const size_t hist_bytes = std::max(hist_elems * n_streams_ * sizeof(float), kCtlCopyMin);
  for (int i = 0; i < 2; i++) {
    rc = dev_alloc(device_, reinterpret_cast<void **>(&n.hist[i]), hist_bytes);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  {
    // Both buffers start from one pinned image of the histories (zeros = silence) through the copy
    // engines (see kCtlCopyMin; not hipMemsetAsync either: a fill kernel queues behind other states'
    // kernels just the same -- set_quality beside another state's 6 ms launch took 6 ms).
    PinnedImage image;
    DrainOnExit drain(&own_stream_);  // (declared behind the buffers it protects: an early return waits for the copies first)
    HIP_TRY(image.get(hist_bytes));
    std::memset(image.p, 0, hist_bytes);
    if (!hist.empty()) std::memcpy(image.p, hist.data(), hist_elems * n_streams_ * sizeof(float));
    for (int i = 0; i < 2; i++)
      HIP_TRY(hipMemcpyAsync(n.hist[i], image.p, hist_bytes, hipMemcpyHostToDevice, own_stream_));
    HIP_TRY(hipStreamSynchronize(own_stream_));  // the new buffers are in place for a launch on any stream
    drain.armed = false;
  }
  trace.step("  history buffers + their image");
  // What the batch held goes back to the pool / the cache below, so nothing in flight may still read it:
  // wait for this batch's own last call -- not for the device: other states' launches keep running.
  if (tables_ != nullptr) {
    rc = quiesce();
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  // commit: nothing below can fail
  hist_bytes_ = hist_bytes;
  std::swap(d_hist_[0], n.hist[0]);
  std::swap(d_hist_[1], n.hist[1]);  // (~Fresh releases the old history buffers)
  tables_ = tables;
  d_table_ = tables->table;
  d_period_rows_ = tables->period_rows;
  d_period_fine_rows_ = tables->fine_rows;
  d_period_w16_rows_ = tables->w16_rows;
  d_slide_rows_ = tables->slide_rows;
  d_slide64_rows_ = tables->slide64_rows;
  d_period64_rows_ = tables->period64_rows;
  d_period_pp_rows_ = tables->pp_rows;
  d_period_pp_w16_rows_ = tables->pp_w16_rows;
  d_period64_fine_rows_ = tables->fine64_rows;
  d_period64_w16_rows_ = tables->period64_w16_rows;
  const std::vector<float> no_table;
  filter_ = f;
  filter_.table = no_table;  // the host copy of the sinc table lives only while the tables are built
  line_ = std::max(line_, f.taps - 1 + kBlockIn);  // grow-only, resample.c:709-720
  hist_elems_ = hist_elems;
  hist_cur_ = 0;
  exact_geo_ = tables->geo;
  exact_geo_ch_ = tables->geo_ch;
  period_ = tables->period;
  period_fine_ = tables->fine;
  period_w16_ = tables->w16;
  slide_ = tables->slide;
  slide64_ = tables->slide64;
  period64_ = tables->period64;
  period_pp_ = tables->pp;
  period_pp_w16_ = tables->pp_w16;
  period64_fine_ = tables->fine64;
  period64_w16_ = tables->period64_w16;
  return SPEEXHIP_ERR_SUCCESS;
}

// Wait for what THIS batch has enqueued.  Calls on one batch are ordered (a call on another stream
// than the previous one waits for it on the device, chain_to), so the previous call's stream ends
// with the batch's last piece of work; the control stream's own copies are waited for where they are
// issued.
int Batch::quiesce() {
  if (ev_pending_) {
    HIP_TRY(hipEventSynchronize(order_ev_));
    ev_pending_ = false;
  }
  if (have_last_stream_) HIP_TRY(hipStreamSynchronize(last_stream_));
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::chain_to(hipStream_t stream) {
  if (have_last_stream_ && stream != last_stream_) {
    if (order_ev_ == nullptr) HIP_TRY(pool::event_get(device_, &order_ev_));
    HIP_TRY(hipEventRecord(order_ev_, last_stream_));
    ev_pending_ = true;
  }
  if (ev_pending_) {
    HIP_TRY(hipStreamWaitEvent(stream, order_ev_, 0));
    ev_pending_ = false;  // (from here on the tail of `stream` stands for it)
  }
  last_stream_ = stream;
  have_last_stream_ = true;
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::release_stream() {
  ON_DEVICE();
  if (!have_last_stream_ || last_stream_ == own_stream_) return SPEEXHIP_ERR_SUCCESS;  // (the pool's streams never die)
  if (order_ev_ == nullptr) HIP_TRY(pool::event_get(device_, &order_ev_));
  HIP_TRY(hipEventRecord(order_ev_, last_stream_));
  ev_pending_ = true;
  have_last_stream_ = false;
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::fetch_history(std::vector<float> *host) {
  ON_DEVICE();
  int rc = quiesce();  // every enqueued call of this batch has left its history
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  host->assign(hist_elems_ * n_streams_, 0.f);
  if (!host->empty())
    return ctl_download(host->data(), reinterpret_cast<const char *>(d_hist_[hist_cur_]), hist_bytes_, 0,
                        host->size() * sizeof(float), own_stream_);
  return SPEEXHIP_ERR_SUCCESS;
}

// Switch every stream to the filter `next` (already designed): the tail of update_filter(),
// resample.c:703-782, per channel, on the host -- this is rare control-plane work.  Positions and
// device state change only if everything succeeded.
int Batch::adopt_filter(const FilterSpec &next) {
  ON_DEVICE();
  std::vector<float> old;
  int rc = fetch_history(&old);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  const uint32_t old_taps = filter_.taps, new_taps = next.taps;
  const size_t old_stride = hist_elems_;
  std::vector<StreamPos> moved = pos_;
  std::vector<Realign> moves(pos_.size());
  uint32_t cap_frames = new_taps - 1;
  for (uint32_t s = 0; s < n_streams_; s++)
    for (uint32_t c = 0; c < channels_; c++) {
      const size_t i = static_cast<size_t>(s) * channels_ + c;
      if (started_[s]) {
        moves[i] = realign_history(old_taps, new_taps, pos_[i].magic);
        moved[i].magic = moves[i].new_magic;
        moved[i].last += moves[i].last_delta;
      }
      cap_frames = std::max(cap_frames, new_taps - 1 + moved[i].magic);
    }
  std::vector<float> fresh(static_cast<size_t>(cap_frames) * channels_ * n_streams_, 0.f);
  for (uint32_t s = 0; s < n_streams_; s++) {
    if (!started_[s]) continue;  // resample.c:721-726: nothing processed yet -> silence
    const float *src = old.data() + s * old_stride;
    float *dst = fresh.data() + static_cast<size_t>(s) * cap_frames * channels_;
    for (uint32_t c = 0; c < channels_; c++) {
      const size_t i = static_cast<size_t>(s) * channels_ + c;
      const int64_t have = static_cast<int64_t>(old_taps - 1) + pos_[i].magic;
      const uint32_t keep = new_taps - 1 + moved[i].magic;
      for (uint32_t j = 0; j < keep; j++) {
        const int64_t from = static_cast<int64_t>(j) + moves[i].shift;
        if (from < 0 || from >= have) continue;
        dst[static_cast<size_t>(j) * channels_ + c] = src[from * channels_ + c];
      }
    }
  }
  rc = install_filter(next, fresh, cap_frames);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  pos_ = moved;
  return SPEEXHIP_ERR_SUCCESS;
}

// resample.c:785-791: the new filter could not be built.  The reference keeps its old filter
// length (so its history stays where it is), has already taken the new rates / ratio / quality
// and the advances derived from them, and installs resampler_basic_zero: outputs are zeros with
// the right lengths until a later filter change succeeds.
void Batch::enter_zero_mode(const FilterSpec &n) {
  filter_.in_rate = n.in_rate;
  filter_.out_rate = n.out_rate;
  filter_.num = n.num;
  filter_.den = n.den;
  filter_.quality = n.quality;
  filter_.int_advance = n.int_advance;
  filter_.frac_advance = n.frac_advance;
  zero_mode_ = true;
}

// Common tail of set_rate_frac / set_quality: `next` was designed with result design_rc; fracs =
// every channel's phase numerator on the new denominator (set_rate_frac) or null.
int Batch::change_filter(const FilterSpec &next, int design_rc, const std::vector<uint32_t> *fracs) {
  int rc = design_rc;
  if (rc == SPEEXHIP_ERR_SUCCESS) rc = adopt_filter(next);
  if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;  // nothing was touched
  if (fracs != nullptr)
    for (size_t i = 0; i < pos_.size(); i++) pos_[i].frac = (*fracs)[i];
  if (rc == SPEEXHIP_ERR_ALLOC_FAILED) {
    enter_zero_mode(next);
    return rc;
  }
  zero_mode_ = false;
  // (a filter change that succeeded after a set_rate_frac had overflowed -- set_quality with the old ratio: what
  //  get_rate / get_ratio show is the filter in force again, not rates that never were)
  shown_valid_ = false;
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::set_rate_frac(uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate) {
  if (ratio_num == 0 || ratio_den == 0) return SPEEXHIP_ERR_INVALID_ARG;
  // resample.c:1116-1117 (compares the given ratio with the REDUCED one it stores, as there)
  const RateView now = rates();
  if (now.in_rate == in_rate && now.out_rate == out_rate && now.num == ratio_num && now.den == ratio_den)
    return SPEEXHIP_ERR_SUCCESS;
  FilterSpec next;
  const int design_rc =
      design_filter_frac(ratio_num, ratio_den, in_rate, out_rate, filter_.quality, &next, /*fill_table=*/false);
  if (design_rc != SPEEXHIP_ERR_SUCCESS && design_rc != SPEEXHIP_ERR_ALLOC_FAILED) return design_rc;
  // phase numerators move to the new denominator (resample.c:1130-1139; next.num / next.den are set
  // even when the design failed later on).  On overflow the reference returns RESAMPLER_ERR_OVERFLOW from
  // the middle of that loop: it has ALREADY stored the new rates and the reduced ratio (:1119-1127), so
  // get_rate / get_ratio report them and a repeat of the same call is a no-op (:1116), while its filter,
  // its advances and the phase numerators of the failing and later channels are still the old ones --
  // a state in which its own processing indexes the old sinc table with numerators on two denominators.
  // Mirrored here: what a caller can SEE (rates, ratio, the no-op repeat).  Not mirrored: processing goes
  // on with the old ratio and filter, consistently, until a later set_rate succeeds (named deviation,
  // pinned by test_set_rate_overflow_leaves_the_reference_s_visible_state).
  std::vector<uint32_t> frac(pos_.size());
  for (size_t i = 0; i < pos_.size(); i++) {
    frac[i] = pos_[i].frac;
    if (!scale_phase(&frac[i], next.den, filter_.den)) {
      shown_ = RateView{in_rate, out_rate, next.num, next.den};
      shown_valid_ = true;
      return SPEEXHIP_ERR_OVERFLOW;
    }
  }
  const int rc = change_filter(next, design_rc, &frac);
  if (rc == SPEEXHIP_ERR_SUCCESS || rc == SPEEXHIP_ERR_ALLOC_FAILED) shown_valid_ = false;  // filter_ holds the truth again
  return rc;
}

int Batch::set_quality(int quality) {
  if (quality > 10 || quality < 0) return SPEEXHIP_ERR_INVALID_ARG;
  if (quality == filter_.quality) return SPEEXHIP_ERR_SUCCESS;  // resample.c:1157-1158
  FilterSpec next;
  // the stored ratio is already reduced, so designing from it reproduces num/den
  const int design_rc =
      design_filter_frac(filter_.num, filter_.den, filter_.in_rate, filter_.out_rate, quality, &next, /*fill_table=*/false);
  if (design_rc != SPEEXHIP_ERR_SUCCESS && design_rc != SPEEXHIP_ERR_ALLOC_FAILED) return design_rc;
  return change_filter(next, design_rc, nullptr);
}

int Batch::skip_zeros() {  // resample.c:1200-1206
  for (StreamPos &p : pos_) p.last = static_cast<int32_t>(filter_.taps / 2);
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::reset_mem() {  // resample.c:1208-1220
  ON_DEVICE();
  std::vector<float> h;
  int rc = fetch_history(&h);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  // The reference clears the first channels*(taps-1) floats of its buffer as ONE run, but its
  // channel lines are line_ floats apart: only the lines (or parts of lines) that fall inside
  // that run are silenced, later channels keep their history.  Restated, not "fixed": the
  // outputs after a reset must be the reference's.
  const uint64_t run = static_cast<uint64_t>(channels_) * (filter_.taps - 1);
  for (uint32_t s = 0; s < n_streams_; s++) {
    float *line = h.data() + s * hist_elems_;
    for (uint32_t c = 0; c < channels_; c++) {
      const uint64_t first = static_cast<uint64_t>(c) * line_;
      for (uint32_t j = 0; j + 1 < filter_.taps && first + j < run; j++) line[static_cast<size_t>(j) * channels_ + c] = 0.f;
    }
    for (uint32_t c = 0; c < channels_; c++) P(s, c) = StreamPos();
  }
  if (!h.empty()) {
    // the whole buffer (>= kCtlCopyMin) from an image: what lies behind the histories is never read
    std::vector<float> image(hist_bytes_ / sizeof(float), 0.f);
    std::copy(h.begin(), h.end(), image.begin());
    return ctl_upload(d_hist_[hist_cur_], image.data(), hist_bytes_, own_stream_);
  }
  return SPEEXHIP_ERR_SUCCESS;
}

Batch::~Batch() {
  if (device_ < 0) return;  // setup() never got as far as a device
  if (counted_) devices::state_gone(device_);
  DeviceScope device_scope(device_);
  // everything goes back to the pool (pool.h) for the next state, once nothing in flight uses it:
  // this batch's calls are chained, so the tail of its last stream is all there is to wait for
  // (a device-wide wait here stalled a server's every other state behind one state's garbage collection)
  (void)quiesce();
  pool::event_put(device_, order_ev_);
  tables_.reset();  // shared (DeviceTables): the cache keeps them for the next state with this filter
  pool::device_put(device_, d_hist_[0]);
  pool::device_put(device_, d_hist_[1]);
  pool::device_put(device_, d_stage_in_);
  pool::device_put(device_, d_stage_out_);
  pool::pinned_put(h_pin_in_);
  pool::pinned_put(h_pin_out_);
  for (int i = 0; i < kMaxPieces; i++) pool::event_put(device_, piece_ev_[i]);
  pool::stream_put(device_, copy_stream_);
  pool::stream_put(device_, own_stream_);
}

CallPlan Batch::peek(uint32_t s, uint32_t in_len, uint32_t out_capacity, bool float_io) const {
  EntryRules rules;
  rules.block_in = block_in();
  rules.float_entry = float_io;
  // (an interleaved call reports the counters of its LAST channel, resample.c:1070-1078)
  return plan_call(filter_.num, filter_.den, in_len, out_capacity, P(s, channels_ - 1), rules);
}

int Batch::set_mode(int mode) {
  if (mode != SPEEXHIP_MODE_FAST && mode != SPEEXHIP_MODE_EXACT && mode != SPEEXHIP_MODE_FAST_F32 &&
      mode != SPEEXHIP_MODE_FAST_FIXED)
    return SPEEXHIP_ERR_INVALID_ARG;
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
  const bool double_kind = filter_.kind == kDirectDouble || filter_.kind == kInterpolateDouble;
  // (what FAST would run: in EXACT mode too; FAST_F32 reports its own fp32-chain kernels)
  if (double_kind && mode_ != SPEEXHIP_MODE_FAST_F32 && !period_.usable && slide64_.usable) o->fast_path = 4;
  if (double_kind && mode_ != SPEEXHIP_MODE_FAST_F32 && period64_.usable) o->fast_path = 5;
  o->accumulate_bits = mode_ == SPEEXHIP_MODE_EXACT || o->fast_path == 0 ? (double_kind ? 64 : 32)
                                                                          : (o->fast_path >= 4 ? 64 : 32);
  if (s < n_streams_) {
    o->last_sample = P(s, 0).last;
    o->samp_frac_num = P(s, 0).frac;
    o->magic_samples = P(s, 0).magic;
  }
  o->block_in = block_in();
  o->device = device_;
}

int Batch::history(uint32_t s, float *dst) {
  if (s >= n_streams_) return SPEEXHIP_ERR_INVALID_ARG;
  ON_DEVICE();
  int rc = quiesce();
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  const size_t n = static_cast<size_t>(filter_.taps - 1 + max_magic(s)) * channels_;
  if (n)
    return ctl_download(dst, reinterpret_cast<const char *>(d_hist_[hist_cur_]), hist_bytes_,
                        s * hist_elems_ * sizeof(float), n * sizeof(float), own_stream_);
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::process_device(const void *d_in, uint64_t in_stride, uint32_t *in_len, void *d_out,
                          uint64_t out_stride, uint32_t *out_len, bool float_io, hipStream_t stream) {
  ON_DEVICE();
  // the int16 entry point emits at most 1024 outputs per 160-frame block (its stack buffer,
  // resample.c:982-991); the float entry point has no such cap (resample.c:943)
  EntryRules rules;
  rules.block_in = block_in();
  rules.float_entry = float_io;
  for (uint32_t s = 0; s < n_streams_; s++)
    if (!uniform(s)) {
      // channels that the per-channel entry points moved apart: one channel at a time, as the
      // reference's interleaved call does (resample.c:1061-1082)
      if (n_streams_ != 1) return SPEEXHIP_ERR_BAD_STATE;
      return process_split(d_in, in_len, d_out, out_len, float_io, stream, nullptr);
    }
  std::vector<CallPlan> plans(n_streams_);
  for (uint32_t s = 0; s < n_streams_; s++) {
    plans[s] = plan_call(filter_.num, filter_.den, in_len[s], out_len[s], P(s, 0), rules);
    // resample.c:886: any block run marks the state as started
    if (in_len[s] != 0 && out_len[s] != 0) started_[s] = 1;
  }
  const int rc = run_plans(d_in, in_stride, in_len, d_out, out_stride, plans.data(), float_io, stream);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  for (uint32_t s = 0; s < n_streams_; s++) {
    in_len[s] = plans[s].consumed;
    out_len[s] = plans[s].produced;
  }
  // resample.c:1081 (and :962, :1035): with resampler_basic_zero installed every call reports the failure
  return zero_mode_ ? SPEEXHIP_ERR_ALLOC_FAILED : SPEEXHIP_ERR_SUCCESS;
}

// One channel of a single-stream batch from plan.begin to plan.end on device buffers whose frames
// are in_stride / out_stride samples apart (the exact kernel on one channel; bit-exact in both
// modes).  The history ping-pong is per stream, so the channel's new line is copied back beside
// the other channels' lines.
int Batch::run_channel(uint32_t c, const void *d_in, uint32_t in_stride, uint32_t in_frames, void *d_out,
                       uint32_t out_stride, const CallPlan &plan, bool float_io, hipStream_t stream) {
  const uint32_t walked = plan.magic_used + plan.consumed;
  if (float_io) float_seen_ = true;
  if (plan.produced == 0 && walked == 0) return SPEEXHIP_ERR_SUCCESS;
  int chain_rc = chain_to(stream);
  if (chain_rc != SPEEXHIP_ERR_SUCCESS) return chain_rc;
  DescPack pack;
  std::memset(&pack, 0, sizeof(pack));
  StreamDesc &d = pack.d[0];
  d.in = d_in;
  d.hist = d_hist_[hist_cur_] + c;
  d.out = d_out;
  d.hist_next = d_hist_[hist_cur_ ^ 1] + c;
  d.in_frames = in_frames;
  d.n_out = plan.produced;
  d.consumed = walked;
  d.hist_frames = filter_.taps - 1 + plan.begin.magic;
  d.hist_keep = filter_.taps - 1 + plan.end.magic;
  d.last0 = plan.begin.last;
  d.frac0 = plan.begin.frac;
  const ExactStrides strides = {in_stride, out_stride, channels_};
  ExactGeometry geo = exact_geo_ch_;
  if (zero_mode_) {  // the window geometry belongs to the filter that is no longer in force
    geo.staged = false;
    geo.lds_bytes = 0;
    geo.outs_per_block = 256;
  }
  const hipError_t e = launch_exact(filter_, geo, d_table_, 1, &pack, 1, plan.produced, float_io, stream, &strides, zero_mode_);
  if (hip_failed(e, "kernel launch")) return SPEEXHIP_ERR_DEVICE;
  if (d.hist_keep != 0)
    HIP_TRY(hipMemcpy2DAsync(d_hist_[hist_cur_] + c, channels_ * sizeof(float), d_hist_[hist_cur_ ^ 1] + c,
                             channels_ * sizeof(float), sizeof(float), d.hist_keep, hipMemcpyDeviceToDevice,
                             stream));
  return SPEEXHIP_ERR_SUCCESS;
}

// The interleaved call on a stream whose channels stand at different positions: channel by
// channel with the caller's lengths restored before each, the lengths of the LAST channel
// reported (resample.c:1061-1082).  d_in / d_out: interleaved device buffers.
int Batch::process_split(const void *d_in, uint32_t *in_len, void *d_out, uint32_t *out_len, bool float_io,
                         hipStream_t stream, std::vector<CallPlan> *plans_out) {
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  EntryRules rules;
  rules.block_in = block_in();
  rules.float_entry = float_io;
  const uint32_t want_in = *in_len, want_out = *out_len;
  if (want_in != 0 && want_out != 0) started_[0] = 1;
  for (uint32_t c = 0; c < channels_; c++) {
    const CallPlan plan = plan_call(filter_.num, filter_.den, want_in, want_out, P(0, c), rules);
    const int rc = run_channel(c, d_in ? static_cast<const char *>(d_in) + c * es : nullptr, channels_, want_in,
                               static_cast<char *>(d_out) + c * es, channels_, plan, float_io, stream);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    P(0, c) = plan.end;
    *in_len = plan.consumed;
    *out_len = plan.produced;
    if (plans_out) plans_out->push_back(plan);
  }
  return zero_mode_ ? SPEEXHIP_ERR_ALLOC_FAILED : SPEEXHIP_ERR_SUCCESS;
}

// The kernel for one launch of up to 32 stream descriptors (`descs` = pack.d): what the mode, the filter's plans and
// the launch's size select.
int Batch::launch_chunk(const StreamDesc *descs, const DescPack &pack, uint32_t n, uint32_t max_out, bool float_io,
                        hipStream_t stream) {
  hipError_t e;
  const bool fast = mode_ != SPEEXHIP_MODE_EXACT;
  const bool fixed = mode_ == SPEEXHIP_MODE_FAST_FIXED;  // (kernels.h: no tap-range shares -- an output's bits depend on the stream alone)
  if (zero_mode_) {
    ExactGeometry geo = exact_geo_;  // (its window geometry belongs to the filter no longer in force)
    geo.staged = false;
    geo.lds_bytes = 0;
    geo.outs_per_block = 256;
    e = launch_exact(filter_, geo, d_table_, channels_, &pack, n, max_out, float_io, stream, nullptr, true);
  } else if (fast && acc64() && period64_.usable && !float_io && !float_seen_ && period64_w16_.usable && !w16_never() &&
             (w16_always() || period_launch_prefers_w16(filter_, period64_, period64_fine_.usable, descs, n))) {
    // ... over an int16 LDS window where the float window holds a fraction of a tile (wide windows; round 5)
    e = launch_period(filter_, period64_w16_, reinterpret_cast<const float *>(d_period64_w16_rows_), nullptr, nullptr, channels_,
                      descs, &pack, n, false, stream, fixed);
  } else if (fast && acc64() && period64_.usable) {
    // the reference sums these filters in fp64 (resample.c:389-435, :501-558): v_fma_f64 kernels
    e = launch_period(filter_, period64_, reinterpret_cast<const float *>(d_period64_rows_), &period64_fine_,
                      reinterpret_cast<const float *>(d_period64_fine_rows_), channels_, descs, &pack, n, float_io, stream, fixed);
  } else if (fast && acc64() && !period_.usable && slide64_.usable) {
    e = launch_slide64(filter_, slide64_, d_slide64_rows_, channels_, descs, &pack, n, float_io, stream, fixed);
  } else if (fast && period_pp_.usable &&
             period_launch_prefers_pp(filter_, (!float_io && !float_seen_ && period_w16_.usable) ? period_w16_ : period_,
                                      (!float_io && !float_seen_ && period_pp_w16_.usable) ? period_pp_w16_ : period_pp_, descs, n)) {
    // up to three channels, wide windows: phase pairs (lane = (period, channel), half the window per tile) where this
    // launch gains
    const bool w16 = !float_io && !float_seen_ && period_pp_w16_.usable;
    e = launch_period(filter_, w16 ? period_pp_w16_ : period_pp_, w16 ? d_period_pp_w16_rows_ : d_period_pp_rows_, nullptr,
                      nullptr, channels_, descs, &pack, n, float_io, stream, fixed);
  } else if (fast && period_.usable && !float_io && !float_seen_ && period_w16_.usable &&
             (w16_always() || period_launch_prefers_w16(filter_, period_, period_fine_.usable, descs, n))) {
    // wide windows: twice the periods per tile over an int16 LDS image (the histories hold PCM values) -- unless
    // the launch is too small for that to pay (period_launch_prefers_w16)
    e = launch_period(filter_, period_w16_, d_period_w16_rows_, nullptr, nullptr, channels_, descs, &pack, n, false, stream, fixed);
  } else if (fast && period_.usable && period_.float_ok) {
    e = launch_period(filter_, period_, d_period_rows_, &period_fine_, d_period_fine_rows_, channels_, descs, &pack, n,
                      float_io, stream, fixed);
  } else if (fast && slide_.usable) {
    e = launch_slide(filter_, slide_, d_slide_rows_, channels_, descs, &pack, n, float_io, stream, fixed);
  } else {
    e = launch_exact(filter_, exact_geo_, d_table_, channels_, &pack, n, max_out, float_io, stream);
  }
  if (hip_failed(e, "kernel launch")) return SPEEXHIP_ERR_DEVICE;
  return SPEEXHIP_ERR_SUCCESS;
}

// Every stream from plans[s].begin to plans[s].end: one launch per 32 streams (kMaxPackedStreams: the descriptors
// of a launch travel in its kernel arguments -- no descriptor copy, no dependent load in the kernels).  in_frames[s]
// = frames readable at the stream's input pointer.
// (Until round 4 a batch of more than 32 streams was ONE launch whose descriptors went through a pinned -> device
//  ring, and every kernel existed twice -- reading its descriptor from the arguments or from the ring: half of the
//  library's code objects, half of its build time and of what a process loads at its first call, for launches that
//  are many generations long anyway; BASELINE configs[4] is 32 streams per GPU.  A launch's ragged end costs ~14 us
//  of a 32-stream launch's ~200: what a 64-stream batch pays for being two launches.)
int Batch::run_plans(const void *d_in, uint64_t in_stride, const uint32_t *in_frames, void *d_out,
                     uint64_t out_stride, const CallPlan *plans, bool float_io, hipStream_t stream) {
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  const uint32_t kChunk = static_cast<uint32_t>(kMaxPackedStreams);
  if (float_io) float_seen_ = true;  // (from here on the histories may hold non-integer samples)
  bool chained = false;
  for (uint32_t s0 = 0; s0 < n_streams_; s0 += kChunk) {
    const uint32_t n = std::min(kChunk, n_streams_ - s0);
    DescPack pack;
    std::memset(&pack, 0, sizeof(pack));
    StreamDesc *descs = pack.d;
    uint32_t max_out = 0;
    bool any_work = false;
    for (uint32_t j = 0; j < n; j++) {
      const uint32_t s = s0 + j;
      const CallPlan &plan = plans[s];
      StreamDesc &d = descs[j];
      d.in = d_in ? static_cast<const char *>(d_in) + s * in_stride * es : nullptr;
      d.hist = d_hist_[hist_cur_] + s * hist_elems_;
      d.out = static_cast<char *>(d_out) + s * out_stride * es;
      d.hist_next = d_hist_[hist_cur_ ^ 1] + s * hist_elems_;
      d.in_frames = in_frames[s];
      d.n_out = plan.produced;
      d.consumed = plan.magic_used + plan.consumed;  // frames of V past the history
      d.hist_frames = filter_.taps - 1 + plan.begin.magic;
      d.hist_keep = filter_.taps - 1 + plan.end.magic;
      d.last0 = plan.begin.last;
      d.frac0 = plan.begin.frac;
      d.k_shift = phase_index_of(filter_.num, filter_.den, plan.begin.frac);
      d.base_shift = plan.begin.last -
                     static_cast<int32_t>((static_cast<uint64_t>(d.k_shift) * filter_.num) / filter_.den);
      d.tile_begin = 0;
      d.m_total = static_cast<uint32_t>((static_cast<uint64_t>(d.k_shift) + d.n_out + filter_.den - 1) / filter_.den);
      max_out = std::max(max_out, plan.produced);
      any_work = any_work || plan.produced != 0 || d.consumed != 0;
    }
    if (!any_work) {
      // (nothing to run for these streams -- but the ping-pong below flips for the whole batch: their histories
      //  must move to the other buffer with everybody else's, which the kernels do even for an idle stream)
      bool others = false;
      for (uint32_t s = 0; s < n_streams_ && !others; s++)
        others = plans[s].produced != 0 || plans[s].magic_used + plans[s].consumed != 0;
      if (!others) continue;
    }
    if (!chained) {
      // Calls on one batch are ordered (each reads the history the previous one left and the
      // ping-pong buffers alternate): a call enqueued on another stream than the previous one
      // waits for it on the device.
      const int chain_rc = chain_to(stream);
      if (chain_rc != SPEEXHIP_ERR_SUCCESS) return chain_rc;
      chained = true;
    }
    const int lrc = launch_chunk(descs, pack, n, max_out, float_io, stream);
    if (lrc != SPEEXHIP_ERR_SUCCESS) return lrc;
  }
  if (chained) hist_cur_ ^= 1;
  for (uint32_t s = 0; s < n_streams_; s++)
    for (uint32_t c = 0; c < channels_; c++) P(s, c) = plans[s].end;
  if (!float_io) int16_call_done(plans, n_streams_);
  return SPEEXHIP_ERR_SUCCESS;
}

// Round 6 (VERDICT r5 #7b): ONE float call must not keep a state off its int16 window for good.  float_seen_ says the
// histories may hold samples an int16 image cannot (a float call put them there).  An int16 call that consumes at least
// as many frames as the history it leaves behind (taps - 1 + the pending frames) replaces every one of them with its own
// int16 input: from the next call on the int16-window plans serve again.  (THIS call still ran over the float window: it
// read the old history.)
void Batch::int16_call_done(const CallPlan *plans, uint32_t n) {
  if (!float_seen_) return;
  for (uint32_t s = 0; s < n; s++)
    if (plans[s].consumed < filter_.taps - 1 + plans[s].end.magic) return;
  float_seen_ = false;
}

// Staging buffers of the host-buffer calls, each grow-only (like the wrapper's heap buffers,
// src/index.ts:71-87) and taken only when a call needs it: small calls run on the pinned pair alone,
// large ones on the device pair alone (the runtime copies straight from / to the caller's memory).
// The calls are synchronous, so nothing in flight uses a buffer that goes back to the pool here.
int Batch::ensure_stage(size_t dev_in, size_t dev_out, size_t pin_in, size_t pin_out) {
  if (own_stream_ == nullptr) HIP_TRY(pool::stream_get(device_, &own_stream_));
  auto grow = [&](char **buf, size_t *cap_now, size_t want, bool pinned) -> int {
    if (want <= *cap_now) return SPEEXHIP_ERR_SUCCESS;
    if (pinned)
      pool::pinned_put(*buf);
    else
      pool::device_put(device_, *buf);
    *buf = nullptr;
    *cap_now = 0;
    const size_t cap = pool::size_class(std::max<size_t>(want, 8192));
    if (pinned)
      HIP_TRY(pool::pinned_get(reinterpret_cast<void **>(buf), cap));
    else
      HIP_TRY(pool::device_get(device_, reinterpret_cast<void **>(buf), cap));
    *cap_now = cap;
    return SPEEXHIP_ERR_SUCCESS;
  };
  int rc = grow(&d_stage_in_, &stage_in_cap_, dev_in, false);
  if (rc == SPEEXHIP_ERR_SUCCESS) rc = grow(&d_stage_out_, &stage_out_cap_, dev_out, false);
  if (rc == SPEEXHIP_ERR_SUCCESS) rc = grow(&h_pin_in_, &pin_in_cap_, pin_in, true);
  if (rc == SPEEXHIP_ERR_SUCCESS) rc = grow(&h_pin_out_, &pin_out_cap_, pin_out, true);
  return rc;
}

// The second stream of a piecewise call: one of the pool's shared streams that is NOT this state's own one (on the
// same stream the copies and the launches would simply take turns: correct, but nothing overlaps).
bool Batch::have_copy_stream() {
  if (copy_stream_ != nullptr) return copy_stream_ != own_stream_;
  if (own_stream_ == nullptr && pool::stream_get(device_, &own_stream_) != hipSuccess) return false;
  for (int tries = 0; tries < 4; tries++) {
    hipStream_t s = nullptr;
    if (pool::stream_get(device_, &s) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    if (s != own_stream_) {
      copy_stream_ = s;
      return true;
    }
  }
  return false;
}

// A large owned-block call in `pieces` pieces (process_host_take).  Piece i: input frames [f_i, f_{i+1}) copied on
// copy_stream_, an event behind the copy, and on own_stream_ -- waiting for that event -- one launch for the outputs
// whose windows end inside the frames copied so far.  Output k of the call reads V-frames
// [last0 + (frac0 + k*num) div den, ... + taps) (stream_plan.h), so the count is closed-form; a piece's descriptor is
// the call's with the position advanced by its first output.  A tile's window may reach past the frames copied so far:
// what it stages from there only feeds outputs of later pieces, which this launch does not store.
int Batch::take_in_pieces(const void *in, uint32_t *in_len, uint32_t *out_len, bool float_io, void *blk, uint32_t pieces) {
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  const uint32_t frames = *in_len;
  EntryRules rules;
  rules.block_in = block_in();
  rules.float_entry = float_io;
  const CallPlan plan = plan_call(filter_.num, filter_.den, frames, *out_len, P(0, 0), rules);
  if (frames != 0 && *out_len != 0) started_[0] = 1;
  const size_t in_bytes = static_cast<size_t>(frames) * channels_ * es;
  DrainOnExit drain(&own_stream_);
  int rc = ensure_stage(in_bytes, 0, 0, 0);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  DrainOnExit drain_copy(&copy_stream_);
  for (uint32_t i = 0; i < pieces; i++)
    if (piece_ev_[i] == nullptr) HIP_TRY(pool::event_get(device_, &piece_ev_[i]));
  rc = chain_to(own_stream_);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (float_io) float_seen_ = true;
  const uint32_t hist_frames = filter_.taps - 1 + plan.begin.magic;
  const int64_t num = filter_.num, den = filter_.den;
  // outputs (of the call's plan.produced) whose windows lie inside history + the first f input frames
  auto outputs_within = [&](uint64_t f) -> uint32_t {
    const int64_t x = static_cast<int64_t>(f) + hist_frames - filter_.taps - plan.begin.last;
    if (x < 0) return 0;
    const int64_t k = ((x + 1) * den - 1 - static_cast<int64_t>(plan.begin.frac)) / num;  // last such output
    return static_cast<uint32_t>(std::min<int64_t>(k + 1, plan.produced));
  };
  // A fast kernel runs every phase of a group over the group's whole zero-padded row (row_len >= taps steps), so an
  // output's accumulator also meets samples up to row_len - taps frames BEHIND its own window -- times zero taps.
  // Behind the frames copied so far the staging buffer holds stale pool memory (or the next piece arriving): as float
  // samples that may be Inf / NaN, and 0 * NaN = NaN.  A piece therefore only stores the outputs whose
  // whole ROW lies inside the frames copied so far: the bound moves back by the longest row any plan of this filter
  // may run (+ one loop iteration of prefetch).
  uint32_t guard = 0;
  for (const PeriodPlan *t : {&period_, &period_fine_, &period_w16_, &period_pp_, &period_pp_w16_, &period64_, &period64_fine_})
    if (t->usable && t->row_len > filter_.taps) guard = std::max(guard, t->row_len - filter_.taps);
  for (const SlidePlan *t : {&slide_, &slide64_})
    if (t->usable && t->row_len > filter_.taps) guard = std::max(guard, t->row_len - filter_.taps + t->p * t->num);
  guard += 16;
  uint32_t done_out = 0;
  for (uint32_t i = 0; i < pieces; i++) {
    const uint64_t f0 = static_cast<uint64_t>(frames) * i / pieces, f1 = static_cast<uint64_t>(frames) * (i + 1) / pieces;
    const size_t off = static_cast<size_t>(f0) * channels_ * es, bytes = static_cast<size_t>(f1 - f0) * channels_ * es;
    if (bytes != 0) HIP_TRY(hipMemcpyAsync(d_stage_in_ + off, static_cast<const char *>(in) + off, bytes, hipMemcpyHostToDevice, copy_stream_));
    HIP_TRY(hipEventRecord(piece_ev_[i], copy_stream_));
    HIP_TRY(hipStreamWaitEvent(own_stream_, piece_ev_[i], 0));
    const bool last = i + 1 == pieces;
    const uint32_t upto = last ? plan.produced : std::max(done_out, outputs_within(f1 > guard ? f1 - guard : 0));
    const uint32_t n_out = upto > done_out ? upto - done_out : 0;
    if (n_out == 0 && !last) continue;
    // the piece's first output: the call's position advanced by done_out outputs
    const uint64_t adv = static_cast<uint64_t>(plan.begin.frac) + static_cast<uint64_t>(done_out) * filter_.num;
    StreamPos at = plan.begin;
    at.last = plan.begin.last + static_cast<int32_t>(adv / filter_.den);
    at.frac = static_cast<uint32_t>(adv % filter_.den);
    DescPack pack;
    std::memset(&pack, 0, sizeof(pack));
    StreamDesc &d = pack.d[0];
    d.in = d_stage_in_;
    d.hist = d_hist_[hist_cur_];
    d.out = static_cast<char *>(blk) + static_cast<size_t>(done_out) * channels_ * es;
    d.hist_next = d_hist_[hist_cur_ ^ 1];
    d.in_frames = frames;
    d.n_out = n_out;
    d.consumed = plan.magic_used + plan.consumed;
    d.hist_frames = hist_frames;
    d.hist_keep = last ? filter_.taps - 1 + plan.end.magic : 0;  // (the history moves once, with the last piece)
    d.last0 = at.last;
    d.frac0 = at.frac;
    d.k_shift = phase_index_of(filter_.num, filter_.den, at.frac);
    d.base_shift = at.last - static_cast<int32_t>((static_cast<uint64_t>(d.k_shift) * filter_.num) / filter_.den);
    d.tile_begin = 0;
    d.m_total = static_cast<uint32_t>((static_cast<uint64_t>(d.k_shift) + d.n_out + filter_.den - 1) / filter_.den);
    rc = launch_chunk(pack.d, pack, 1, n_out, float_io, own_stream_);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    done_out = upto;
  }
  hist_cur_ ^= 1;
  HIP_TRY(hipStreamSynchronize(own_stream_));  // (behind the last launch, which waited for the last copy)
  drain.armed = false;
  drain_copy.armed = false;
  for (uint32_t c = 0; c < channels_; c++) P(0, c) = plan.end;
  if (!float_io) int16_call_done(&plan, 1);
  *in_len = plan.consumed;
  *out_len = plan.produced;
  return SPEEXHIP_ERR_SUCCESS;
}

// Host buffer in, result in a pinned block the caller owns afterwards (the N-API addon wraps it in an external
// Buffer: src/index.ts:111-115 returns a fresh, caller-owned Buffer, and so does this -- without the copy into it).
// The kernel writes the block directly: small calls as before (they already wrote pinned memory, then the result
// was copied out of it: 0.18 us per KB), large ones instead of a device buffer + a copy back.  States whose
// channels stand apart and the zero fallback take the ordinary path into the block.
int Batch::process_host_take(const void *in, uint32_t *in_len, uint32_t *out_len, bool float_io, void **block) {
  *block = nullptr;
  if (n_streams_ != 1) return SPEEXHIP_ERR_BAD_STATE;
  ON_DEVICE();
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  const uint32_t frames = *in_len;
  const bool split = !uniform(0);
  uint32_t will_make = 0;
  for (uint32_t c = 0; c < (split ? channels_ : 1u); c++)
    will_make = std::max(will_make, produced_closed_form(filter_.num, filter_.den, frames, *out_len, P(0, c)));
  const size_t in_bytes = static_cast<size_t>(frames) * channels_ * es;
  const size_t out_bytes = static_cast<size_t>(will_make) * channels_ * es;
  // (+ 64 bytes behind the samples: the completion word)  Nothing has touched the state yet: without a block the
  // caller makes the copying call instead.
  void *blk = nullptr;
  if (!pool::block_get(&blk, out_bytes + 128)) return SPEEXHIP_ERR_NO_BLOCK;
  struct Guard {
    void *p;
    ~Guard() {
      if (p != nullptr) (void)pool::block_put(p);
    }
  } guard{blk};
  int rc;
  // Large calls in pieces (round 4): PCIe is full duplex, and with the kernel writing the result block itself the
  // two directions belong to different engines -- the input copies run on a second stream, and behind each an event
  // lets a launch for the outputs that piece completes start while the next piece is still arriving.  No kernel
  // change: a piece is a StreamDesc of the same call that begins o_i outputs later (positions advanced in integers)
  // and only the last one rolls the history.  SPEEXHIP_PIECES=1 turns it off, =n forces n (A/B, tests).
  const uint32_t env_pieces = static_cast<uint32_t>(std::max(0, diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_PIECES"), 0)));
  uint32_t pieces = env_pieces > 0 ? env_pieces : static_cast<uint32_t>(in_bytes / kPieceBytes);
  pieces = std::min<uint32_t>(pieces, env_pieces > 0 ? kMaxPieces : 4);  // (the rule: at most four; up to kMaxPieces when forced)
  // Round 6: an input the caller left in pinned memory (speexhip_block_acquire, or memory it pinned itself) is read where
  // it lies -- ONE launch whose loads and stores cross PCIe in opposite directions at the same time, no staging copy, no
  // second stream (the pieces above overlap the two directions only partly and pay ~15 us per piece for it).
  const void *pin_in = (split || zero_mode_ || in == nullptr) ? nullptr : pinned_view(in, in_bytes);
  if (split || zero_mode_) {
    rc = process_host(in, in_len, blk, out_len, float_io);
    if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;
  } else if ((pin_in == nullptr || pinned_in_pieces()) && pieces >= 2 && in != nullptr && in_bytes >= kZeroCopyBelow && frames < 0x40000000u && have_copy_stream()) {
    rc = take_in_pieces(in, in_len, out_len, float_io, blk, pieces);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  } else {
    DrainOnExit drain(&own_stream_);
    const bool direct_in = pin_in == nullptr && in_bytes >= kZeroCopyBelow;
    rc = ensure_stage(direct_in ? in_bytes : 0, 0, (direct_in || pin_in != nullptr) ? 0 : in_bytes, 0);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    const void *src = nullptr;
    if (pin_in != nullptr) {
      src = pin_in;
    } else if (in != nullptr && in_bytes != 0) {
      if (direct_in) {
        HIP_TRY(hipMemcpyAsync(d_stage_in_, in, in_bytes, hipMemcpyHostToDevice, own_stream_));
        src = d_stage_in_;
      } else {
        std::memcpy(h_pin_in_, in, in_bytes);
        src = h_pin_in_;
      }
    }
    volatile uint32_t *done = reinterpret_cast<volatile uint32_t *>(static_cast<char *>(blk) + ((out_bytes + 63) & ~static_cast<size_t>(63)));
    *done = 0;
    rc = process_device(src, 0, in_len, blk, 0, out_len, float_io, own_stream_);
    if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;
    if (direct_in) {
      HIP_TRY(hipStreamSynchronize(own_stream_));
    } else {
      const int wrc = wait_done(own_stream_, done, 1u, spin_budget_us(pin_in != nullptr ? in_bytes + out_bytes : 0));
      if (wrc != SPEEXHIP_ERR_SUCCESS) return wrc;
    }
    drain.armed = false;
  }
  if (*out_len == 0) return rc;  // (the guard returns the block)
  guard.p = nullptr;
  *block = blk;
  return rc;
}

void Batch::release_block(void *block) { (void)pool::block_put(block); }

int Batch::process_host(const void *in, uint32_t *in_len, void *out, uint32_t *out_len, bool float_io) {
  if (n_streams_ != 1) return SPEEXHIP_ERR_BAD_STATE;
  ON_DEVICE();
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  const uint32_t frames = *in_len;
  const bool split = !uniform(0);
  // only as many output frames as this call can produce need a device buffer
  uint32_t will_make = 0;
  for (uint32_t c = 0; c < (split ? channels_ : 1u); c++)
    will_make = std::max(will_make, produced_closed_form(filter_.num, filter_.den, frames, *out_len, P(0, c)));
  const size_t in_bytes = static_cast<size_t>(frames) * channels_ * es;
  const size_t out_bytes = static_cast<size_t>(will_make) * channels_ * es;
  int rc = SPEEXHIP_ERR_SUCCESS;
  DrainOnExit drain(&own_stream_);
  // Large buffers go straight from / to the caller's pageable memory: the HIP runtime stages such
  // copies itself and does it 2.2-2.5x faster than memcpy -> pinned -> DMA in one thread (2^20
  // stereo frames: 0.46 -> 0.21 ms per call, 8 channels 1.57 -> 0.63 ms).  Small ones go through
  // the pinned bounce buffers (below).
  // What was tried in round 2 to get below this (2^20 stereo frames, 207 us per call; tools/ubench_copy.hip):
  // PCIe is full duplex -- both copies at once from pinned memory take 103 us instead of 175 -- but
  //   * H2D / kernel / D2H of 2-8 pieces on three streams chained by events: +30 us per piece (a
  //     cross-stream wait costs ~14 us on this stack and nothing overlapped);
  //   * pieces alternating on two independent in-order streams: 189 us at 2 pieces, more beyond (a
  //     copy-engine <-> kernel hand-over costs ~10 us, and kernels of two streams never ran side by side);
  //   * the caller's buffers pinned for the call (hipHostRegister, ~5 us) and read / written by the
  //     kernels straight through PCIe, one launch: 166 us -- and, one run in three, a stretch of stale
  //     zeros near the end of the output when buffers at recycled addresses were pinned again.  Not
  //     shippable; removed.
  // So the call stays three in-order steps on one stream.
  const bool direct_in = in_bytes >= kDirectCopyBytes, direct_out = out_bytes >= kDirectCopyBytes && !split;
  // Small calls -- a Transform's 64 KiB chunks, a realtime caller's 10-20 ms frames -- are all latency:
  // the kernels read the pinned bounce buffer and write the pinned result buffer straight through PCIe
  // (pool-owned hipHostMalloc memory, coherent; one launch and one wait instead of copy / launch / copy
  // / wait): 480-960 stereo frames 26.5 -> 22.7 us per call, 16384 frames 41.7 -> 30.1, 65536 frames
  // 70.4 -> 53.7 (tools/small_call_latency.py).  SPEEXHIP_ZERO_COPY_BELOW=0 turns it off (A/B).
  // Where it stops paying (profiles/r03_zero_copy_sweep.txt, per call, pinned alone vs copies): 256 KB of input
  // 49 vs 65 us (stereo), 49 vs 66 (mono), 46 vs 65 (8 channels); 512 KB 76 vs 95, 76 vs 95, 72 vs 90; 1 MB
  // 167 vs 145, 169 vs 144, 132 vs 150: the single-threaded memcpy into and out of the bounce buffers grows at
  // 0.18 us per KB against 0.10 for the runtime's own staged copies -- they cross near 740 KB.  (Until late in
  // round 3 the limit was 256 KB, which sent a 65536-frame stereo chunk down the slower way.)
  static const size_t zero_copy_below = [] {
    const char *e = SPEEXHIP_DIAG_ENV("SPEEXHIP_ZERO_COPY_BELOW");
    return e != nullptr ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : kZeroCopyBelow;
  }();
  // Round 6: buffers the caller keeps in pinned memory (speexhip_block_acquire, hipHostMalloc, hipHostRegister) are used
  // where they lie (pinned_view): a pinned input is read by the kernel through PCIe, a pinned output written by it -- with
  // both pinned the call is one launch and one wait, the two crossings side by side.  The other side, if pageable, keeps its
  // own rule: small through the bounce buffer, large by the runtime's staged copy.
  if (!split) {
    const void *pin_in = in != nullptr ? pinned_view(in, in_bytes) : nullptr;
    void *pin_out = pinned_view(out, out_bytes);
    // (a result that overlaps its chunk: the chunk is taken in whole before anything is written, as on pageable buffers)
    if (pin_in != nullptr && pin_out != nullptr && buffers_overlap(in, in_bytes, out, out_bytes)) pin_in = nullptr;
    if (pin_in != nullptr || pin_out != nullptr) {
      const bool have_in = in != nullptr && in_bytes != 0;
      const bool bounce_in = have_in && pin_in == nullptr && in_bytes < zero_copy_below;
      const bool copy_in = have_in && pin_in == nullptr && !bounce_in;
      const bool bounce_out = pin_out == nullptr && out_bytes < zero_copy_below;
      const bool copy_out = pin_out == nullptr && !bounce_out;
      rc = ensure_stage(copy_in ? in_bytes : 0, copy_out ? out_bytes : 0, bounce_in ? in_bytes : 0, (bounce_out ? out_bytes : 0) + 64);
      if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
      const void *src = nullptr;
      if (pin_in != nullptr) {
        src = pin_in;
      } else if (bounce_in) {
        std::memcpy(h_pin_in_, in, in_bytes);
        src = h_pin_in_;
      } else if (copy_in) {
        HIP_TRY(hipMemcpyAsync(d_stage_in_, in, in_bytes, hipMemcpyHostToDevice, own_stream_));
        src = d_stage_in_;
      } else if (in != nullptr) {
        src = h_pin_out_;  // (an empty chunk, not silence: no frame is read, any non-null address serves)
      }
      void *dst = pin_out != nullptr ? pin_out : bounce_out ? static_cast<void *>(h_pin_out_) : static_cast<void *>(d_stage_out_);
      volatile uint32_t *done = reinterpret_cast<volatile uint32_t *>(h_pin_out_ + ((pin_out_cap_ - 64) & ~static_cast<size_t>(63)));
      const uint32_t seq = ++done_seq_;
      *done = seq - 1;
      rc = process_device(src, 0, in_len, dst, 0, out_len, float_io, own_stream_);
      if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;
      const size_t made = static_cast<size_t>(*out_len) * channels_ * es;
      if (copy_in || copy_out) {
        if (copy_out && made != 0) HIP_TRY(hipMemcpyAsync(out, d_stage_out_, made, hipMemcpyDeviceToHost, own_stream_));
        HIP_TRY(hipStreamSynchronize(own_stream_));
      } else {
        const int wrc = wait_done(own_stream_, done, seq, spin_budget_us((pin_in != nullptr ? in_bytes : 0) + (pin_out != nullptr ? out_bytes : 0)));
        if (wrc != SPEEXHIP_ERR_SUCCESS) return wrc;
      }
      drain.armed = false;
      if (bounce_out && made != 0) std::memcpy(out, h_pin_out_, made);
      return rc;
    }
  }
  if (!split && in_bytes < zero_copy_below && out_bytes < zero_copy_below) {
    // (+ 64 bytes: the completion word below lives behind the samples, in the same pinned block)
    rc = ensure_stage(0, 0, in_bytes, out_bytes + 64);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
    if (in != nullptr && in_bytes != 0) std::memcpy(h_pin_in_, in, in_bytes);
    // The wait: hipStreamSynchronize after a tiny launch costs 11-12 us on this stack; a 32-bit stream write behind
    // the kernel (hipStreamWriteValue32: performed once everything before it on the stream has completed) into
    // pinned memory, polled by the caller, 8.8 (tools/ubench_sync.hip).  A launch that has not signalled after
    // 300 us is waited for -- and its error, if that is what happened, reported -- the ordinary way.
    volatile uint32_t *done = reinterpret_cast<volatile uint32_t *>(h_pin_out_ + ((pin_out_cap_ - 64) & ~static_cast<size_t>(63)));
    const uint32_t seq = ++done_seq_;
    *done = seq - 1;
    rc = process_device(in != nullptr ? h_pin_in_ : nullptr, 0, in_len, h_pin_out_, 0, out_len, float_io, own_stream_);
    if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;
    const int wrc = wait_done(own_stream_, done, seq, 300);  // (the word: see wait_done)
    if (wrc != SPEEXHIP_ERR_SUCCESS) return wrc;
    drain.armed = false;
    const size_t made = static_cast<size_t>(*out_len) * channels_ * es;
    if (made != 0) std::memcpy(out, h_pin_out_, made);
    return rc;
  }
  rc = ensure_stage(in_bytes, out_bytes, direct_in ? 0 : in_bytes, direct_out ? 0 : out_bytes);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (in != nullptr && in_bytes != 0) {
    if (direct_in) {
      HIP_TRY(hipMemcpyAsync(d_stage_in_, in, in_bytes, hipMemcpyHostToDevice, own_stream_));
    } else {
      std::memcpy(h_pin_in_, in, in_bytes);
      HIP_TRY(hipMemcpyAsync(d_stage_in_, h_pin_in_, in_bytes, hipMemcpyHostToDevice, own_stream_));
    }
  }
  if (split) {
    // channels at different positions write different numbers of frames: fetch the whole block
    // and hand the caller only the samples each channel really wrote
    std::vector<CallPlan> plans;
    rc = process_split(in != nullptr ? d_stage_in_ : nullptr, in_len, d_stage_out_, out_len, float_io, own_stream_,
                       &plans);
    if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;
    uint32_t most = 0;
    for (const CallPlan &pl : plans) most = std::max(most, pl.produced);
    const size_t bytes = static_cast<size_t>(most) * channels_ * es;
    if (bytes != 0) HIP_TRY(hipMemcpyAsync(h_pin_out_, d_stage_out_, bytes, hipMemcpyDeviceToHost, own_stream_));
    HIP_TRY(hipStreamSynchronize(own_stream_));
    drain.armed = false;
    for (uint32_t c = 0; c < channels_; c++)
      for (uint32_t j = 0; j < plans[c].produced; j++)
        std::memcpy(static_cast<char *>(out) + (static_cast<size_t>(j) * channels_ + c) * es,
                    h_pin_out_ + (static_cast<size_t>(j) * channels_ + c) * es, es);
    return rc;
  }
  rc = process_device(in != nullptr ? d_stage_in_ : nullptr, 0, in_len, d_stage_out_, 0, out_len, float_io,
                      own_stream_);
  if (rc != SPEEXHIP_ERR_SUCCESS && rc != SPEEXHIP_ERR_ALLOC_FAILED) return rc;
  const size_t made = static_cast<size_t>(*out_len) * channels_ * es;
  if (made != 0)
    HIP_TRY(hipMemcpyAsync(direct_out ? out : h_pin_out_, d_stage_out_, made, hipMemcpyDeviceToHost, own_stream_));
  HIP_TRY(hipStreamSynchronize(own_stream_));
  drain.armed = false;
  if (made != 0 && !direct_out) std::memcpy(out, h_pin_out_, made);
  return rc;
}

// speex_resampler_process_int / _process_float (resample.c:927-1036): ONE channel, host buffers
// whose samples are in_stride_ / out_stride_ apart (resample.c:1170-1188).  The samples travel as a
// dense line; the state's other channels are not touched.
int Batch::process_channel_host(uint32_t c, const void *in, uint32_t *in_len, void *out, uint32_t *out_len,
                                bool float_io) {
  if (n_streams_ != 1) return SPEEXHIP_ERR_BAD_STATE;
  if (c >= channels_) return SPEEXHIP_ERR_INVALID_ARG;
  ON_DEVICE();
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  EntryRules rules;
  rules.block_in = block_in();
  rules.float_entry = float_io;
  const uint32_t frames = *in_len;
  const CallPlan plan = plan_call(filter_.num, filter_.den, frames, *out_len, P(0, c), rules);
  if (frames != 0 && *out_len != 0) started_[0] = 1;
  const size_t line_in = static_cast<size_t>(frames) * es, line_out = static_cast<size_t>(plan.produced) * es;
  DrainOnExit drain(&own_stream_);
  int rc = ensure_stage(line_in, line_out, line_in, line_out);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (in != nullptr && frames != 0) {
    for (uint32_t j = 0; j < frames; j++)
      std::memcpy(h_pin_in_ + static_cast<size_t>(j) * es,
                  static_cast<const char *>(in) + static_cast<size_t>(j) * in_stride_ * es, es);
    HIP_TRY(hipMemcpyAsync(d_stage_in_, h_pin_in_, static_cast<size_t>(frames) * es, hipMemcpyHostToDevice,
                           own_stream_));
  }
  rc = run_channel(c, in != nullptr ? d_stage_in_ : nullptr, 1, frames, d_stage_out_, 1, plan, float_io, own_stream_);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  if (plan.produced != 0)
    HIP_TRY(hipMemcpyAsync(h_pin_out_, d_stage_out_, static_cast<size_t>(plan.produced) * es, hipMemcpyDeviceToHost,
                           own_stream_));
  HIP_TRY(hipStreamSynchronize(own_stream_));
  drain.armed = false;
  for (uint32_t j = 0; j < plan.produced; j++)
    std::memcpy(static_cast<char *>(out) + static_cast<size_t>(j) * out_stride_ * es,
                h_pin_out_ + static_cast<size_t>(j) * es, es);
  P(0, c) = plan.end;
  *in_len = plan.consumed;
  *out_len = plan.produced;
  return zero_mode_ ? SPEEXHIP_ERR_ALLOC_FAILED : SPEEXHIP_ERR_SUCCESS;
}

// A sequence of calls on one stream as one transfer and (normally) one launch.  Every output
// is a function of (history ++ the frames consumed so far) and its own index; call boundaries
// only decide the counters.  So: plan the calls one after the other on the host (integers),
// pack the frames each call really consumes back to back, and run the stream from the first
// call's begin state to the last call's end state.  One exception: a capacity-bound call that
// leaves input unconsumed -- its last outputs may still have read the first frame it then drops
// (up-sampling: several outputs share one newest frame), while the next call sees its own first
// frame in that slot.  Such a call closes a launch group: it keeps that one extra frame
// readable, and the calls after it start a new launch on the same stream.
int Batch::process_host_chunks(uint32_t n_chunks, const void *const *in, uint32_t *in_len, void *out,
                               uint32_t *out_len, bool float_io) {
  if (n_streams_ != 1) return SPEEXHIP_ERR_BAD_STATE;
  ON_DEVICE();
  const size_t fb = (float_io ? sizeof(float) : sizeof(int16_t)) * channels_;  // bytes per frame
  if (!uniform(0) || zero_mode_) {  // rare states: the separate calls, one after the other
    int last = SPEEXHIP_ERR_SUCCESS;
    char *o = static_cast<char *>(out);
    for (uint32_t i = 0; i < n_chunks; i++) {
      last = process_host(in != nullptr ? in[i] : nullptr, &in_len[i], o, &out_len[i], float_io);
      if (last != SPEEXHIP_ERR_SUCCESS && last != SPEEXHIP_ERR_ALLOC_FAILED) return last;
      o += static_cast<size_t>(out_len[i]) * fb;
    }
    return last;
  }
  EntryRules rules;
  rules.block_in = block_in();
  rules.float_entry = float_io;
  struct Group {
    CallPlan fused;
    uint64_t in_off = 0, out_off = 0;  // frames into the staging buffers
    uint32_t readable = 0;             // frames at in_off the launch may read
  };
  std::vector<CallPlan> plans(n_chunks);
  std::vector<Group> groups;
  StreamPos pos = P(0, 0);
  uint64_t frames = 0, made = 0;
  bool open = false;
  for (uint32_t i = 0; i < n_chunks; i++) {
    plans[i] = plan_call(filter_.num, filter_.den, in_len[i], out_len[i], pos, rules);
    pos = plans[i].end;
    if (in_len[i] != 0 && out_len[i] != 0) started_[0] = 1;
    if (!open) {
      Group g;
      g.fused.begin = plans[i].begin;
      g.in_off = frames;
      g.out_off = made;
      groups.push_back(g);
      open = true;
    }
    Group &g = groups.back();
    g.fused.end = plans[i].end;
    g.fused.magic_used += plans[i].magic_used;
    g.fused.consumed += plans[i].consumed;
    g.fused.produced += plans[i].produced;
    g.readable += plans[i].consumed;
    frames += plans[i].consumed;
    made += plans[i].produced;
    if (plans[i].consumed < in_len[i]) {  // dropped input: one more frame stays readable
      g.readable += 1;
      frames += 1;
      open = false;
    }
    if (frames > 0x7fffffffull || made > 0x7fffffffull) return SPEEXHIP_ERR_OVERFLOW;
  }
  const bool direct_out = made * fb >= kDirectCopyBytes;
  // Small runs -- what a Transform holds back, eight 64 KiB chunks say -- are all latency, like small single calls
  // (process_host): the kernels read the pinned bounce buffer and write the pinned result buffer straight through
  // PCIe, one wait, no copy-engine hand-overs.  (Until round 5 every coalesced run went through the device staging
  // buffers: H2D copy, launch, D2H copy -- three in-order steps of ~10 us hand-over each -- and `coalesceChunks: 8` was
  // SLOWER than the plain pipe on four of the reference's seven test tuples, profiles/r04_node_bench.json.)
  const bool zero_copy = frames * fb < kZeroCopyBelow && made * fb < kZeroCopyBelow;
  DrainOnExit drain(&own_stream_);
  int rc = zero_copy ? ensure_stage(0, 0, frames * fb, made * fb + 64)
                     : ensure_stage(frames * fb, made * fb, frames * fb, direct_out ? 0 : made * fb);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  size_t off = 0;
  for (uint32_t i = 0; i < n_chunks; i++) {  // frames a call drops never reach the GPU
    const size_t bytes = (plans[i].consumed + (plans[i].consumed < in_len[i] ? 1u : 0u)) * fb;
    if (in != nullptr && in[i] != nullptr)
      std::memcpy(h_pin_in_ + off, in[i], bytes);
    else
      std::memset(h_pin_in_ + off, 0, bytes);
    off += bytes;
  }
  if (off != 0 && !zero_copy) HIP_TRY(hipMemcpyAsync(d_stage_in_, h_pin_in_, off, hipMemcpyHostToDevice, own_stream_));
  char *src = zero_copy ? h_pin_in_ : d_stage_in_, *dst = zero_copy ? h_pin_out_ : d_stage_out_;
  for (const Group &g : groups) {
    rc = run_plans(src + g.in_off * fb, 0, &g.readable, dst + g.out_off * fb, 0, &g.fused, float_io, own_stream_);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  if (made != 0 && !zero_copy)
    HIP_TRY(hipMemcpyAsync(direct_out ? out : h_pin_out_, d_stage_out_, made * fb, hipMemcpyDeviceToHost, own_stream_));
  HIP_TRY(hipStreamSynchronize(own_stream_));
  drain.armed = false;
  if (made != 0 && (zero_copy || !direct_out)) std::memcpy(out, h_pin_out_, made * fb);
  for (uint32_t i = 0; i < n_chunks; i++) {
    in_len[i] = plans[i].consumed;
    out_len[i] = plans[i].produced;
  }
  return SPEEXHIP_ERR_SUCCESS;
}

// ---- warm-up (speexhip_warmup) ------------------------------------------------------------------------------------
// What the FIRST state of a process pays that has nothing to do with the state: the runtime's own start (the first HIP
// call, 90-180 ms), the first stream (20 ms; 160 ms when it is also the runtime's first), each further shared stream
// (~8 ms for the second to fourth state), the first copy (~8 ms: the copy engines' queues) -- against 0.04 ms of filter
// design and 0.2 ms of uploads (profiles/r05_first_call_trace.txt).  The reference pays its counterpart -- compiling
// the WASM module -- when the module is imported, behind SpeexResampler.initPromise (src/index.ts:18-19, :31); so does
// the drop-in: the addon runs this on a pool thread at import and initPromise resolves behind it.
int warm_device(int device) {
  DeviceScope scope(device);
  HIP_TRY(scope.error());
  HIP_TRY(pool::streams_prewarm(device));
  hipStream_t s = nullptr;
  HIP_TRY(pool::stream_get(device, &s));
  void *d = nullptr;
  HIP_TRY(pool::device_get(device, &d, kCtlCopyMin));
  const std::vector<char> zeros(kCtlCopyMin, 0);
  const int rc = ctl_upload(d, zeros.data(), zeros.size(), s);  // (pinned image -> device through the copy engines)
  pool::device_put(device, d);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  // the code objects (kernels.h, SPEEXHIP_WARM_UNIT): one empty launch per translation unit
  warm_unit_period(s);
  warm_unit_slide_i16(s);
  warm_unit_exact(s);
  warm_unit_period64(s);
  warm_unit_slide64_i16(s);
  warm_unit_period_pp(s);
  warm_unit_period_odd(s);
  warm_unit_period_frames(s);
  warm_unit_period64_w16(s);
  warm_unit_period_w16g(s);
  warm_unit_slide_f32(s);
  warm_unit_slide64_f32(s);
  HIP_TRY(hipStreamSynchronize(s));
  return SPEEXHIP_ERR_SUCCESS;
}

int warmup(int device) {
  const int count = devices::count();
  if (count <= 0) {
    g_last_error = "HIP device error: no GPU visible (libspeexhip has no CPU fallback)";
    return SPEEXHIP_ERR_DEVICE;
  }
  if (device >= count) return SPEEXHIP_ERR_INVALID_ARG;
  if (device >= 0) return warm_device(device);
  int list[64], n = 0;
  devices::placement_candidates(list, &n, 64);
  if (n == 0) {
    g_last_error = "HIP device error: SPEEXHIP_DEVICE / SPEEXHIP_DEVICES names a device this node does not have";
    return SPEEXHIP_ERR_DEVICE;
  }
  for (int k = 0; k < n; k++) {
    const int rc = warm_device(list[k]);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  return SPEEXHIP_ERR_SUCCESS;
}

// ---- host-buffer calls of many single-stream states at once (engine.h) ---------------------------------------------
namespace {
// Staging of the many-states call: one per logical device, grow-only, taken from the pool and kept for the life of the
// process like the shared streams.  A call holds the lock of every device it touches from its first copy to its last.
struct ManyStage {
  std::mutex mu;
  hipStream_t stream = nullptr, copy_stream = nullptr;  // (copy_stream + events: the pipelined large path)
  std::vector<hipEvent_t> events;
  char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
  size_t d_in_cap = 0, d_out_cap = 0, h_in_cap = 0, h_out_cap = 0;
  uint32_t seq = 0;
};
// (device, lane): a large call on one device runs as two halves side by side, each on a stage of its own (below)
ManyStage &many_stage(int device, int lane) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, ManyStage *> *all = new std::map<std::pair<int, int>, ManyStage *>();  // never destroyed, like the pool
  std::lock_guard<std::mutex> lock(mu);
  ManyStage *&m = (*all)[std::make_pair(device, lane)];
  if (m == nullptr) m = new ManyStage();
  return *m;
}
int grow_stage(int device, char **buf, size_t *cap_now, size_t want, bool pinned) {
  if (want <= *cap_now) return SPEEXHIP_ERR_SUCCESS;
  if (pinned)
    pool::pinned_put(*buf);
  else
    pool::device_put(device, *buf);
  *buf = nullptr;
  *cap_now = 0;
  const size_t cap = pool::size_class(std::max<size_t>(want, 8192));
  if (pinned)
    HIP_TRY(pool::pinned_get(reinterpret_cast<void **>(buf), cap));
  else
    HIP_TRY(pool::device_get(device, reinterpret_cast<void **>(buf), cap));
  *cap_now = cap;
  return SPEEXHIP_ERR_SUCCESS;
}
inline size_t align64(size_t v) { return (v + 63) & ~static_cast<size_t>(63); }

// The stage's copy stream: a stream of its own that carries host -> device copies and NOTHING else, made and primed here.
// Why (late in round 6; profiles/r06_engine_log.txt, r06_pinned_in_leg.txt): the runtime picks a copy engine per stream --
// for a copy in, the lowest engine free at the moment it asks; for a copy out, the engine ROCr recommends for that
// direction (0x2 on this box) -- and then KEEPS it for that stream whatever the direction of the stream's later copies.
// The pool's streams are shared with every state's own calls, which copy both ways: a copy stream that had last carried
// a state's results kept engine 0x2 for the inputs of the next many-states call, the 32 queued copies in of a call over
// pinned chunks went ahead of every copy out on that one engine, and the call took 6.0-6.3 ms instead of 4.0 (pageable
// chunks: 5.5 instead of 3.8) -- or not, depending on which calls the process had made before.  A stream that only ever
// copies in asks once, here, one stage at a time and with its copy waited for, so that the lowest engine is free when
// the next stage asks: every stage's inputs travel on engine 0x1, the results on 0x2 / 0x4.
int prime_copy_stream(int device, ManyStage &ms) {
  static std::mutex one_at_a_time;
  std::lock_guard<std::mutex> lock(one_at_a_time);
  if (ms.copy_stream != nullptr) return SPEEXHIP_ERR_SUCCESS;
  DeviceScope scope(device);
  HIP_TRY(scope.error());
  hipStream_t s = nullptr;
  HIP_TRY(pool::stream_own(device, &s));
  const size_t bytes = 64 * 1024;  // (above the 16 KiB the runtime copies with a kernel: a copy ENGINE must be asked for)
  void *h = nullptr, *d = nullptr;
  hipEvent_t e = nullptr;
  int rc = SPEEXHIP_ERR_SUCCESS;
  if (hip_failed(pool::pinned_get(&h, bytes), "hipHostMalloc") || hip_failed(pool::device_get(device, &d, bytes), "hipMalloc") ||
      hip_failed(pool::event_get(device, &e), "hipEventCreate") ||
      hip_failed(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync (priming)") ||
      hip_failed(hipEventRecord(e, s), "hipEventRecord") ||
      // (the EVENT is waited for, not the stream: the stream is never synchronised, the engine it was given stays its own)
      hip_failed(hipEventSynchronize(e), "hipEventSynchronize"))
    rc = SPEEXHIP_ERR_DEVICE;
  if (e != nullptr) pool::event_put(device, e);
  if (d != nullptr) pool::device_put(device, d);
  if (h != nullptr) pool::pinned_put(h);
  if (rc != SPEEXHIP_ERR_SUCCESS) {
    (void)hipStreamDestroy(s);
    return rc;
  }
  ms.copy_stream = s;
  return SPEEXHIP_ERR_SUCCESS;
}
}  // namespace

// The fused states of ONE device: idx = their positions in the caller's arrays.
int Batch::many_on_device(int device, int lane, const std::vector<uint32_t> &idx, Batch *const *st, const void *const *in,
                          uint32_t *in_len, void *const *out, uint32_t *out_len, bool float_io, int *rcs) {
  ManyStage &ms = many_stage(device, lane);
  std::lock_guard<std::mutex> lock(ms.mu);
  DeviceScope device_scope(device);
  HIP_TRY(device_scope.error());
  if (ms.stream == nullptr) HIP_TRY(pool::stream_get(device, &ms.stream));
  DrainOnExit drain(&ms.stream);
  const size_t es = float_io ? sizeof(float) : sizeof(int16_t);
  struct Item {
    uint32_t i;
    Batch *b;
    CallPlan plan;
    size_t in_bytes, out_bytes, in_off, out_off;  // bytes that travel through the stage (0: a pinned buffer, used in place)
    const void *pin_in;                           // round 6: the caller's own buffers where they are pinned memory
    void *pin_out;                                // (pinned_view): the kernels read / write them directly
    bool work;
  };
  std::vector<Item> items(idx.size());
  size_t total_in = 0, total_out = 0, pinned_bytes = 0;
  bool all_big = true;
  for (size_t k = 0; k < idx.size(); k++) {
    Item &it = items[k];
    it.i = idx[k];
    it.b = st[it.i];
    EntryRules rules;
    rules.block_in = it.b->block_in();
    rules.float_entry = float_io;
    it.plan = plan_call(it.b->filter_.num, it.b->filter_.den, in_len[it.i], out_len[it.i], it.b->P(0, 0), rules);
    it.in_bytes = in[it.i] != nullptr ? static_cast<size_t>(in_len[it.i]) * it.b->channels_ * es : 0;
    it.out_bytes = static_cast<size_t>(it.plan.produced) * it.b->channels_ * es;
    it.pin_in = pinned_view(in[it.i], it.in_bytes);
    it.pin_out = pinned_view(out[it.i], it.out_bytes);
    if (it.pin_in != nullptr && it.pin_out != nullptr && buffers_overlap(in[it.i], it.in_bytes, out[it.i], it.out_bytes)) it.pin_in = nullptr;
    // (a pinned input whose result goes to a LARGE pageable buffer is copied like any other -- from pinned memory the copy
    //  is a plain DMA -- so that the call can take the pipelined path, inputs arriving while results leave: read in place,
    //  all the reads come first and all the pageable copies out after them, 32 x 2^20 stereo frames 5.4 ms against 4.0,
    //  profiles/r06_bench_driver_form_v2.json)
    if (it.pin_in != nullptr && it.pin_out == nullptr && it.out_bytes >= kDirectCopyBytes &&
        diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_PIN_IN_COPY"), 1) != 0)  // (A/B: 0 = read in place even then)
      it.pin_in = nullptr;
    // (experiment, diagnostics: large chunks AND results in pinned memory by the copy engines as well, pipelined)
    if (it.pin_in != nullptr && it.pin_out != nullptr && it.in_bytes >= kDirectCopyBytes && it.out_bytes >= kDirectCopyBytes &&
        diag_int(SPEEXHIP_DIAG_ENV("SPEEXHIP_PIN_BOTH_COPY"), 0) != 0)
      it.pin_in = nullptr, it.pin_out = nullptr;
    if (it.pin_in != nullptr) pinned_bytes += it.in_bytes, it.in_bytes = 0;
    if (it.pin_out != nullptr) pinned_bytes += it.out_bytes, it.out_bytes = 0;
    it.work = it.plan.produced != 0 || it.plan.magic_used + it.plan.consumed != 0;
    total_in += align64(it.in_bytes);
    total_out += align64(it.out_bytes);
    if (it.work && (it.in_bytes < kDirectCopyBytes || it.out_bytes < kDirectCopyBytes)) all_big = false;
  }
  // Small calls (a server's 10-20 ms frames, a Transform's 64 KiB chunks) are all latency: the kernels read and write
  // pinned memory straight through PCIe, one wait (process_host's small-call path).  Larger ones: inputs of >= 256 KB
  // go from the caller's pageable memory by the runtime's own staged copies, smaller ones are gathered in the pinned
  // buffer and travel as ONE copy; the same on the way out.
  const bool zero_copy = total_in < kZeroCopyBelow && total_out < kZeroCopyBelow;
  // layout: the small buffers first (one contiguous range = one copy), then the large ones
  size_t small_in = 0, small_out = 0, off_in = 0, off_out = 0;
  for (int pass = 0; pass < 2; pass++)
    for (Item &it : items) {
      const bool big_in = !zero_copy && it.in_bytes >= kDirectCopyBytes, big_out = !zero_copy && it.out_bytes >= kDirectCopyBytes;
      if (big_in == (pass == 1)) {
        it.in_off = off_in;
        off_in += align64(it.in_bytes);
        if (pass == 0) small_in = off_in;
      }
      if (big_out == (pass == 1)) {
        it.out_off = off_out;
        off_out += align64(it.out_bytes);
        if (pass == 0) small_out = off_out;
      }
    }
  int rc = grow_stage(device, &ms.h_in, &ms.h_in_cap, zero_copy ? total_in : small_in, true);
  if (rc == SPEEXHIP_ERR_SUCCESS) rc = grow_stage(device, &ms.h_out, &ms.h_out_cap, (zero_copy ? total_out : small_out) + 128, true);
  if (rc == SPEEXHIP_ERR_SUCCESS && !zero_copy) rc = grow_stage(device, &ms.d_in, &ms.d_in_cap, total_in, false);
  if (rc == SPEEXHIP_ERR_SUCCESS && !zero_copy) rc = grow_stage(device, &ms.d_out, &ms.d_out_cap, total_out, false);
  if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  char *src_base = zero_copy ? ms.h_in : ms.d_in, *dst_base = zero_copy ? ms.h_out : ms.d_out;

  // launches: the states that share (tables, mode, window format) go together, at most 32 per launch
  std::vector<std::vector<Item *>> launches;
  {
    std::map<std::tuple<const void *, int, bool>, std::vector<Item *>> groups;
    for (Item &it : items) {
      if (float_io) it.b->float_seen_ = true;
      if (in_len[it.i] != 0 && out_len[it.i] != 0) it.b->started_[0] = 1;  // resample.c:886
      if (!it.work) continue;  // nothing to run: the state stays where it is
      groups[std::make_tuple(static_cast<const void *>(it.b->tables_.get()), it.b->mode_, it.b->float_seen_)].push_back(&it);
    }
    // Large calls in pieces (below): a launch then carries about 16 MB of input, so that the transfer of the next
    // piece and the results of the previous one have something to overlap with.
    static const int env_pipe = SPEEXHIP_DIAG_ENV("SPEEXHIP_MANY_PIPELINE") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_MANY_PIPELINE")) : -1;  // A/B: 0 off
    const bool pipelined = !zero_copy && all_big && env_pipe != 0 && total_in >= (static_cast<size_t>(32) << 20);
    for (auto &kv : groups) {
      std::vector<Item *> &g = kv.second;
      size_t per = kMaxPackedStreams;
      if (pipelined) {
        size_t bytes = 0;
        for (Item *it : g) bytes += it->in_bytes;
        const size_t pieces = std::max<size_t>(1, bytes / (static_cast<size_t>(16) << 20));
        per = std::min<size_t>(kMaxPackedStreams, std::max<size_t>(1, (g.size() + pieces - 1) / pieces));
      }
      for (size_t g0 = 0; g0 < g.size(); g0 += per)
        launches.emplace_back(g.begin() + g0, g.begin() + std::min(g.size(), g0 + per));
    }
  }
  auto commit_item = [&](Item &it) {  // counters and position of one state
    for (uint32_t c = 0; c < it.b->channels_; c++) it.b->P(0, c) = it.plan.end;
    if (!float_io && it.work) it.b->int16_call_done(&it.plan, 1);
    in_len[it.i] = it.plan.consumed;
    out_len[it.i] = it.plan.produced;
    rcs[it.i] = SPEEXHIP_ERR_SUCCESS;
  };
  // one launch of `launches[k]` on `stream`
  auto launch = [&](const std::vector<Item *> &g, hipStream_t stream) -> int {
    const uint32_t cnt = static_cast<uint32_t>(g.size());
    DescPack pack;
    std::memset(&pack, 0, sizeof(pack));
    uint32_t max_out = 0;
    for (uint32_t j = 0; j < cnt; j++) {
      Item &it = *g[j];
      Batch *b = it.b;
      const int crc = b->chain_to(stream);  // (each state's calls stay ordered, whatever stream its last one ran on)
      if (crc != SPEEXHIP_ERR_SUCCESS) return crc;
      StreamDesc &d = pack.d[j];
      const FilterSpec &f = b->filter_;
      d.in = in[it.i] == nullptr ? nullptr : it.pin_in != nullptr ? it.pin_in : src_base + it.in_off;
      d.hist = b->d_hist_[b->hist_cur_];
      d.out = it.pin_out != nullptr ? it.pin_out : dst_base + it.out_off;
      d.hist_next = b->d_hist_[b->hist_cur_ ^ 1];
      d.in_frames = in_len[it.i];
      d.n_out = it.plan.produced;
      d.consumed = it.plan.magic_used + it.plan.consumed;
      d.hist_frames = f.taps - 1 + it.plan.begin.magic;
      d.hist_keep = f.taps - 1 + it.plan.end.magic;
      d.last0 = it.plan.begin.last;
      d.frac0 = it.plan.begin.frac;
      d.k_shift = phase_index_of(f.num, f.den, it.plan.begin.frac);
      d.base_shift = it.plan.begin.last - static_cast<int32_t>((static_cast<uint64_t>(d.k_shift) * f.num) / f.den);
      d.tile_begin = 0;
      d.m_total = static_cast<uint32_t>((static_cast<uint64_t>(d.k_shift) + d.n_out + f.den - 1) / f.den);
      max_out = std::max(max_out, it.plan.produced);
    }
    const int lrc = g[0]->b->launch_chunk(pack.d, pack, cnt, max_out, float_io, stream);
    if (lrc != SPEEXHIP_ERR_SUCCESS) return lrc;
    // A state moves as ONE step, the moment its launch is queued: the history ping-pong, the position and the counters
    // together.  (Until round 6 the flip happened here and the positions behind the last launch of the call: a later
    // group's failure left the earlier groups' states with a flipped history and their OLD position -- silently corrupt
    // for every later call, ADVICE r5.  Now a failure further on costs such a state this call's audio -- its code says
    // so -- and nothing else; states whose launch was never queued have not moved at all.)
    for (uint32_t j = 0; j < cnt; j++) {
      g[j]->b->hist_cur_ ^= 1;
      commit_item(*g[j]);
    }
    return SPEEXHIP_ERR_SUCCESS;
  };
  auto commit = [&]() {  // ... and the states with nothing to run: counters of an empty call, position unchanged
    for (Item &it : items)
      if (!it.work) commit_item(it);
  };

  static const int env_pipe2 = SPEEXHIP_DIAG_ENV("SPEEXHIP_MANY_PIPELINE") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_MANY_PIPELINE")) : -1;
  if (!zero_copy && all_big && env_pipe2 != 0 && total_in >= (static_cast<size_t>(32) << 20) && launches.size() >= 2) {
    // Large calls, pipelined (round 5).  PCIe is full duplex, but the runtime's pageable copies keep the thread that
    // issues them busy until they are staged, so one thread alone moves inputs, computes, and moves results strictly
    // one after the other: 32 streams x 2^20 stereo frames 5.8 ms, of which 0.2 are the kernel.  Here the calling
    // thread copies the inputs of launch after launch on the stage's copy stream (an event behind each launch's
    // inputs), while a second thread waits for the events, runs the launches on the stage's stream and copies each
    // launch's results out behind it: results leave while the next inputs arrive.
    if (ms.copy_stream == nullptr) {
      const int prc = prime_copy_stream(device, ms);  // (process_host_many has done this for every large unit, one after the other)
      if (prc != SPEEXHIP_ERR_SUCCESS) return prc;
    }
    while (ms.events.size() < launches.size()) {
      hipEvent_t e = nullptr;
      HIP_TRY(pool::event_get(device, &e));
      ms.events.push_back(e);
    }
    DrainOnExit drain_copy(&ms.copy_stream);
    std::mutex mu;
    std::condition_variable cv;
    size_t ready = 0;       // launches whose inputs are enqueued with their event recorded
    bool copy_failed = false;
    int worker_rc = SPEEXHIP_ERR_SUCCESS;
    std::string worker_err;
    auto helper = [&] {
      DeviceScope scope(device);
      try {
      for (size_t k = 0; k < launches.size() && worker_rc == SPEEXHIP_ERR_SUCCESS; k++) {
        {
          std::unique_lock<std::mutex> l(mu);
          cv.wait(l, [&] { return ready > k || copy_failed; });
          if (copy_failed) return;
        }
        auto fail = [&](hipError_t e, const char *what) {
          if (e == hipSuccess) return false;
          worker_err = std::string("HIP device error: ") + what + ": " + hipGetErrorString(e);
          worker_rc = SPEEXHIP_ERR_DEVICE;
          return true;
        };
        if (fail(hipStreamWaitEvent(ms.stream, ms.events[k], 0), "hipStreamWaitEvent")) return;
        const int lrc = launch(launches[k], ms.stream);
        if (lrc != SPEEXHIP_ERR_SUCCESS) {
          worker_rc = lrc;
          worker_err = g_last_error;
          return;
        }
        for (Item *it : launches[k])
          if (it->out_bytes != 0 &&
              fail(hipMemcpyAsync(out[it->i], ms.d_out + it->out_off, it->out_bytes, hipMemcpyDeviceToHost, ms.stream), "hipMemcpyAsync (results)"))
            return;
      }
      if (worker_rc == SPEEXHIP_ERR_SUCCESS) {
        const hipError_t e = hipStreamSynchronize(ms.stream);
        if (e != hipSuccess) {
          worker_err = std::string("HIP device error: hipStreamSynchronize: ") + hipGetErrorString(e);
          worker_rc = SPEEXHIP_ERR_DEVICE;
        }
      }
      } catch (...) {  // (a host allocation inside a launcher: no exception leaves a thread)
        worker_rc = SPEEXHIP_ERR_ALLOC_FAILED;
      }
    };
    // (the launching half runs on this (device, lane)'s persistent helper thread -- unit_workers.h; until round 6 a
    //  std::thread made and joined inside every such call.  No thread to be had: the call falls back to one thread,
    //  copies first, launches after -- the helper's loop then finds every event recorded already.)
    workers::Ticket helper_job = workers::submit(workers::key_of(device, lane, 1), helper);
    hipError_t copy_err = hipSuccess;
    for (size_t k = 0; k < launches.size() && copy_err == hipSuccess; k++) {
      for (Item *it : launches[k])
        if (it->in_bytes != 0 && copy_err == hipSuccess)
          copy_err = hipMemcpyAsync(ms.d_in + it->in_off, in[it->i], it->in_bytes, hipMemcpyHostToDevice, ms.copy_stream);
      if (copy_err == hipSuccess) copy_err = hipEventRecord(ms.events[k], ms.copy_stream);
      {
        std::lock_guard<std::mutex> l(mu);
        if (copy_err == hipSuccess)
          ready = k + 1;
        else
          copy_failed = true;
      }
      cv.notify_all();
    }
    if (helper_job != nullptr) {
      workers::wait(helper_job);
      if (workers::failed(helper_job)) worker_rc = SPEEXHIP_ERR_ALLOC_FAILED;
    } else {
      helper();
    }
    if (hip_failed(copy_err, "hipMemcpyAsync (inputs)")) return SPEEXHIP_ERR_DEVICE;
    if (worker_rc != SPEEXHIP_ERR_SUCCESS) {
      g_last_error = worker_err;
      return worker_rc;
    }
    drain.armed = false;
    drain_copy.armed = false;
    commit();
    return SPEEXHIP_ERR_SUCCESS;
  }

  for (const Item &it : items) {
    if (it.in_bytes == 0) continue;
    if (zero_copy || it.in_bytes < kDirectCopyBytes)
      std::memcpy(ms.h_in + it.in_off, in[it.i], it.in_bytes);
    else
      HIP_TRY(hipMemcpyAsync(ms.d_in + it.in_off, in[it.i], it.in_bytes, hipMemcpyHostToDevice, ms.stream));
  }
  if (!zero_copy && small_in != 0) HIP_TRY(hipMemcpyAsync(ms.d_in, ms.h_in, small_in, hipMemcpyHostToDevice, ms.stream));
  for (const std::vector<Item *> &g : launches) {
    rc = launch(g, ms.stream);
    if (rc != SPEEXHIP_ERR_SUCCESS) return rc;
  }
  commit();
  // results back
  if (zero_copy) {
    volatile uint32_t *done = reinterpret_cast<volatile uint32_t *>(ms.h_out + ((ms.h_out_cap - 64) & ~static_cast<size_t>(63)));
    const uint32_t seq = ++ms.seq;
    *done = seq - 1;
    if (!launches.empty()) {
      const int wrc = wait_done(ms.stream, done, seq, spin_budget_us(pinned_bytes));
      if (wrc != SPEEXHIP_ERR_SUCCESS) return wrc;
    }
    drain.armed = false;
    for (const Item &it : items)
      if (it.out_bytes != 0) std::memcpy(out[it.i], ms.h_out + it.out_off, it.out_bytes);
    return SPEEXHIP_ERR_SUCCESS;
  }
  if (small_out != 0) HIP_TRY(hipMemcpyAsync(ms.h_out, ms.d_out, small_out, hipMemcpyDeviceToHost, ms.stream));
  for (const Item &it : items)
    if (it.out_bytes >= kDirectCopyBytes)
      HIP_TRY(hipMemcpyAsync(out[it.i], ms.d_out + it.out_off, it.out_bytes, hipMemcpyDeviceToHost, ms.stream));
  HIP_TRY(hipStreamSynchronize(ms.stream));
  drain.armed = false;
  for (const Item &it : items)
    if (it.out_bytes != 0 && it.out_bytes < kDirectCopyBytes) std::memcpy(out[it.i], ms.h_out + it.out_off, it.out_bytes);
  return SPEEXHIP_ERR_SUCCESS;
}

int Batch::process_host_many(uint32_t n, Batch *const *st, const void *const *in, uint32_t *in_len, void *const *out,
                             uint32_t *out_len, bool float_io, int *codes) {
  std::vector<int> rcs(n, SPEEXHIP_ERR_SUCCESS);
  std::map<int, std::vector<uint32_t>> by_device;  // (ordered: two concurrent calls lock their devices in one order)
  std::vector<uint32_t> apart;                     // states that take the single call (in the caller's order)
  for (uint32_t i = 0; i < n; i++) {
    Batch *b = st[i];
    if (b == nullptr || (out[i] == nullptr && out_len[i] != 0)) {
      rcs[i] = SPEEXHIP_ERR_INVALID_ARG;
      continue;
    }
    bool earlier = false;  // a state named twice: its second call must see the first one's end state
    for (uint32_t j = 0; j < i && !earlier; j++) earlier = st[j] == b;
    // (rare states -- channels moved apart by the per-channel calls, the zero fallback, batches of several streams --
    //  keep their own call's rules)
    if (earlier || b->n_streams_ != 1 || !b->uniform(0) || b->zero_mode_)
      apart.push_back(i);
    else
      by_device[b->device_].push_back(i);
  }
  // Units of work: a device's states -- or, for a large call, two halves of them ("lanes"): one thread's pageable copies
  // do not fill a PCIe link (64 x 2^20 stereo frames on one GPU: 11.0 ms through one stage, 7.6 ms as two logical
  // devices of 32 states each, profiles/r05_host_many.txt), two stages side by side nearly do.
  struct Unit {
    int device, lane;
    std::vector<uint32_t> idx;
  };
  std::vector<Unit> units;
  static const int env_lanes = SPEEXHIP_DIAG_ENV("SPEEXHIP_MANY_LANES") ? std::atoi(SPEEXHIP_DIAG_ENV("SPEEXHIP_MANY_LANES")) : -1;  // A/B: 1 = never split
  for (auto &kv : by_device) {
    // What a lane is for: the runtime's PAGEABLE copies keep the thread that issues them busy, so the bytes that count are
    // the pageable ones of the busier direction.  Buffers in the library's pinned blocks do not count (a range check; read or
    // written in place -- or, chunks beside large pageable results, copied by a plain DMA that keeps no thread).
    // 32 x 2^20 stereo frames, one box (profiles/r06_pin_both_copy.txt, r06_pinned_in_leg.txt): pageable both ways one lane
    // 4.11, two 3.80 ms; pinned chunks + pageable results (146 MB out) 4.04 -> 3.83; both pinned 3.61 / 3.65 (nothing to
    // split).  (Until the stages had copy streams of their own -- prime_copy_stream -- the second case measured 4.02 / 4.54:
    // that was the copy engines' lottery, not the lanes.)
    uint64_t bytes_in = 0, bytes_out = 0;
    for (uint32_t i : kv.second) {
      const uint64_t es = float_io ? 4 : 2;
      const uint64_t b = static_cast<uint64_t>(in_len[i]) * st[i]->channels_ * es;
      if (in[i] != nullptr && !pool::block_owns(in[i], b)) bytes_in += b;
      // (results: about den / num of the input's frames, capped by the caller's capacity)
      const uint64_t frames_out = std::min<uint64_t>(out_len[i], static_cast<uint64_t>(in_len[i]) * st[i]->filter_.den / std::max<uint32_t>(st[i]->filter_.num, 1) + 1);
      const uint64_t o = frames_out * st[i]->channels_ * es;
      if (out[i] != nullptr && !pool::block_owns(out[i], o)) bytes_out += o;
    }
    const uint64_t bytes = std::max(bytes_in, bytes_out);
    // (from 128 MB: 32 x 2^20 stereo frames 4.11 -> 4.00 ms, 64 states 7.90 -> 7.14; at 67 MB nothing, 2.26 / 2.44)
    const bool split = env_lanes != 1 && kv.second.size() >= 8 && (env_lanes == 2 || bytes >= (static_cast<uint64_t>(128) << 20));  // (A/B: 2 = always)
    if (!split) {
      units.push_back(Unit{kv.first, 0, kv.second});
    } else {
      const size_t half = kv.second.size() / 2;
      units.push_back(Unit{kv.first, 0, std::vector<uint32_t>(kv.second.begin(), kv.second.begin() + half)});
      units.push_back(Unit{kv.first, 1, std::vector<uint32_t>(kv.second.begin() + half, kv.second.end())});
    }
  }
  // (the copy streams of the units that will take the pipelined path: made and primed one after the other, before any
  //  unit copies anything -- prime_copy_stream)
  for (const Unit &u : units) {
    uint64_t bytes = 0;
    for (uint32_t i : u.idx) bytes += static_cast<uint64_t>(in_len[i]) * st[i]->channels_ * (float_io ? 4 : 2);
    if (bytes >= (static_cast<uint64_t>(32) << 20)) {
      ManyStage &ms = many_stage(u.device, u.lane);
      std::lock_guard<std::mutex> lock(ms.mu);  // (the order many_on_device takes them in: the stage, then the priming)
      (void)prime_copy_stream(u.device, ms);   // (a failure shows again, with its code, in the unit's own call)
    }
  }
  std::vector<int> dev_rc(units.size(), SPEEXHIP_ERR_SUCCESS);
  std::vector<std::string> dev_err(units.size());
  auto run_device = [&](size_t slot) {
    const Unit &u = units[slot];
    try {
      dev_rc[slot] = many_on_device(u.device, u.lane, u.idx, st, in, in_len, out, out_len, float_io, rcs.data());
    } catch (const std::bad_alloc &) {
      dev_rc[slot] = SPEEXHIP_ERR_ALLOC_FAILED;
    } catch (const std::exception &e) {  // (anything else: a code and its text, never std::terminate in a worker)
      g_last_error = std::string("internal error: ") + e.what();
      dev_rc[slot] = SPEEXHIP_ERR_DEVICE;
    } catch (...) {
      g_last_error = "internal error: unknown exception";
      dev_rc[slot] = SPEEXHIP_ERR_DEVICE;
    }
    if (dev_rc[slot] != SPEEXHIP_ERR_SUCCESS) {
      dev_err[slot] = g_last_error;  // (the text lives per thread)
      for (uint32_t i : u.idx) rcs[i] = dev_rc[slot];
    }
  };
  {
    // GPUs side by side: every further unit of the call runs on the persistent thread of its (device, lane) -- a GPU
    // is a PCIe link of its own, and the runtime's pageable copies keep the thread that issues them busy -- while the
    // calling thread runs the first one (unit_workers.h: one thread per (device, lane) for the life of the process;
    // until round 6 a std::thread per unit per CALL, 7-15 thread creations per step on an 8-GPU node).  A unit whose
    // job cannot be queued runs here, after the first.
    std::vector<workers::Ticket> tickets(units.size());
    for (size_t slot = 1; slot < units.size(); slot++)
      tickets[slot] = workers::submit(workers::key_of(units[slot].device, units[slot].lane, 0), [&run_device, slot] { run_device(slot); });
    if (!units.empty()) run_device(0);
    for (size_t slot = 1; slot < units.size(); slot++) {
      if (tickets[slot] != nullptr)
        workers::wait(tickets[slot]);
      else
        run_device(slot);
    }
    for (size_t k = 0; k < dev_rc.size(); k++)
      if (dev_rc[k] != SPEEXHIP_ERR_SUCCESS) {
        g_last_error = dev_err[k];
        break;
      }
  }
  for (uint32_t i : apart) rcs[i] = st[i]->process_host(in[i], &in_len[i], out[i], &out_len[i], float_io);
  int first = SPEEXHIP_ERR_SUCCESS;
  for (uint32_t i = 0; i < n; i++) {
    if (codes != nullptr) codes[i] = rcs[i];
    if (first == SPEEXHIP_ERR_SUCCESS && rcs[i] != SPEEXHIP_ERR_SUCCESS) first = rcs[i];
  }
  return first;
}

}  // namespace speexhip
