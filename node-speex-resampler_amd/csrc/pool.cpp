// pool.cpp -- see pool.h.
#include "pool.h"

#include "diag.h"

#include <algorithm>
#include <chrono>
#include <iterator>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace speexhip {
namespace pool {
namespace {

// Size classes: 4 KiB, then powers of two with a half step in between (4, 6, 8, 12, 16 ... KiB): a
// request is served with at most 1.5x its size.  Requests above kMaxPooled bypass the pool.
const size_t kMinClass = 4096;
// Pinned host memory the kernels of ANY device of the process may read and write where it lies (round 6: chunks and result
// blocks are used in place, and a state may live on a GPU other than the one that was current when the slab was made):
// portable = allocated for every context, mapped = in the devices' address space.
const unsigned int kPinnedFlags = hipHostMallocPortable | hipHostMallocMapped;
const size_t kMaxPooled = static_cast<size_t>(256) << 20;

struct Shelf {
  std::map<size_t, std::vector<void *>> idle;  // size class -> buffers
  size_t idle_bytes = 0;
};

struct State {
  std::mutex mu;
  std::map<int, Shelf> device;  // per device
  Shelf pinned;
  std::unordered_map<void *, size_t> live;  // every buffer handed out or idle -> its size class (0: not pooled)
  std::map<int, std::vector<hipStream_t>> streams;  // shared, see stream_get
  std::map<int, size_t> next_stream;
  std::map<int, std::vector<hipEvent_t>> events;
  size_t device_cap, pinned_cap;
  State() {
    size_t mb = 1024;
    if (const char *e = std::getenv("SPEEXHIP_POOL_MB")) mb = static_cast<size_t>(std::strtoull(e, nullptr, 10));
    device_cap = mb << 20;
    pinned_cap = (mb / 4) << 20;
  }
};

// never destroyed: HIP may already be gone when static destructors run
State &st() {
  static State *s = new State();
  return *s;
}

// SPEEXHIP_POOL_TRACE=1: every request the pool could not serve, with the time the driver took (stderr)
struct MissTimer {
  const char *what;
  size_t bytes;
  std::chrono::steady_clock::time_point t0;
  static bool on() {
    static const bool v = SPEEXHIP_DIAG_ENV("SPEEXHIP_POOL_TRACE") != nullptr;
    return v;
  }
  MissTimer(const char *w, size_t b) : what(w), bytes(b), t0(std::chrono::steady_clock::now()) {}
  ~MissTimer() {
    if (!on()) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "speexhip pool miss: %s %zu bytes, %.1f us\n", what, bytes, us);
  }
};

void *take(Shelf &sh, size_t cls) {
  auto it = sh.idle.find(cls);
  if (it == sh.idle.end() || it->second.empty()) return nullptr;
  void *p = it->second.back();
  it->second.pop_back();
  sh.idle_bytes -= cls;
  return p;
}

}  // namespace

// ---- the slabs of caller-owned result blocks (pool.h) -- first fit over a sorted free list, 4 KiB granules ----
// One slab of SPEEXHIP_TAKE_MB (64 MiB) is made by the first request, further ones when every slab is full, up to
// SPEEXHIP_TAKE_MAX_MB (256 MiB) in all.  Why more than one (round 5, profiles/r05_steady.txt): the N-API addon's
// external Buffers give their blocks back from a finalizer, and V8 runs finalizers from the event loop, a while after
// the collection that found the Buffer dead.  A caller producing 3.5 MB results held 18 of them when the first
// collection came (V8 starts collecting external memory at ~64 MB), the single slab was full, and every call until the
// finalizers had run took the copying path at 0.30-0.55 ms instead of 0.14.  With room beyond V8's own threshold the
// collector gets there first.
namespace {
struct Slab {
  char *base = nullptr;
  size_t bytes = 0;
  std::map<size_t, size_t> free_at;    // offset -> length
  std::map<size_t, size_t> taken;      // offset -> length
};
struct Slabs {
  std::mutex mu;
  std::vector<Slab> all;
  size_t slab_bytes = 0, max_bytes = 0, total = 0;
  bool configured = false;
  // A slab the driver refused (the pinned-memory limit of the container, say) is not asked for again at once: every
  // ..._take call would pay a failing hipHostMalloc before it falls back to the copying path.  The next
  // attempt waits until a block has come back or the idle memory has been released, or for kRetryCalls requests.
  uint32_t refused_for = 0;
};
const uint32_t kRetryCalls = 256;
Slabs &slabs() {
  static Slabs *s = new Slabs();  // never destroyed, like the pool
  return *s;
}
bool carve(Slab &sl, size_t want, void **ptr) {
  for (auto it = sl.free_at.begin(); it != sl.free_at.end(); ++it) {
    if (it->second < want) continue;
    const size_t off = it->first, len = it->second;
    sl.free_at.erase(it);
    if (len > want) sl.free_at[off + want] = len - want;
    sl.taken[off] = want;
    *ptr = sl.base + off;
    return true;
  }
  return false;
}
}  // namespace

bool block_get(void **ptr, size_t bytes) {
  *ptr = nullptr;
  Slabs &ss = slabs();
  std::lock_guard<std::mutex> lock(ss.mu);
  if (!ss.configured) {
    ss.configured = true;
    size_t mb = 64, max_mb = 256;
    if (const char *e = std::getenv("SPEEXHIP_TAKE_MB")) mb = static_cast<size_t>(std::strtoull(e, nullptr, 10));
    if (const char *e = std::getenv("SPEEXHIP_TAKE_MAX_MB")) max_mb = static_cast<size_t>(std::strtoull(e, nullptr, 10));
    ss.slab_bytes = mb << 20;
    ss.max_bytes = std::max(max_mb, mb) << 20;
  }
  if (ss.slab_bytes == 0) return false;
  const size_t want = (std::max<size_t>(bytes, 1) + 4095) & ~static_cast<size_t>(4095);
  for (Slab &sl : ss.all)
    if (carve(sl, want, ptr)) return true;
  // every slab is full (or there is none yet): one more, while the total stays under the cap
  const size_t size = std::max(ss.slab_bytes, want);
  if (ss.total + size > ss.max_bytes && !(ss.all.empty() && size <= ss.max_bytes)) return false;
  if (ss.refused_for != 0) {
    ss.refused_for--;
    return false;
  }
  void *p = nullptr;
  {
    MissTimer timer("hipHostMalloc (slab of result blocks)", size);
    if (hipHostMalloc(&p, size, kPinnedFlags) != hipSuccess) {
      (void)hipGetLastError();
      ss.refused_for = kRetryCalls;
      return false;
    }
  }
  Slab sl;
  sl.base = static_cast<char *>(p);
  sl.bytes = size;
  sl.free_at[0] = size;
  ss.all.push_back(std::move(sl));
  ss.total += size;
  return carve(ss.all.back(), want, ptr);
}

bool block_put(void *ptr) {
  Slabs &ss = slabs();
  std::lock_guard<std::mutex> lock(ss.mu);
  for (Slab &sl : ss.all) {
    if (ptr < static_cast<void *>(sl.base) || ptr >= static_cast<void *>(sl.base + sl.bytes)) continue;
    const size_t off = static_cast<size_t>(static_cast<char *>(ptr) - sl.base);
    auto it = sl.taken.find(off);
    if (it == sl.taken.end()) return false;
    size_t start = off, len = it->second;
    sl.taken.erase(it);
    auto next = sl.free_at.lower_bound(start);
    if (next != sl.free_at.end() && next->first == start + len) {  // merge with the free range behind
      len += next->second;
      next = sl.free_at.erase(next);
    }
    if (next != sl.free_at.begin()) {  // ... and with the one in front
      auto prev = std::prev(next);
      if (prev->first + prev->second == start) {
        start = prev->first;
        len += prev->second;
        sl.free_at.erase(prev);
      }
    }
    sl.free_at[start] = len;
    ss.refused_for = 0;  // (memory came back: a refused slab may be asked for again)
    return true;
  }
  return false;
}

bool block_owns(const void *ptr, size_t bytes) {
  if (ptr == nullptr) return false;
  Slabs &ss = slabs();
  std::lock_guard<std::mutex> lock(ss.mu);
  const char *p = static_cast<const char *>(ptr);
  for (const Slab &sl : ss.all)
    if (p >= sl.base && p < sl.base + sl.bytes) return bytes <= static_cast<size_t>(sl.base + sl.bytes - p);
  return false;
}

size_t size_class(size_t bytes) {
  if (bytes <= kMinClass) return kMinClass;
  size_t c = kMinClass;
  while (c < bytes) {
    if (c + c / 2 >= bytes) return c + c / 2;
    c *= 2;
  }
  return c;
}

hipError_t device_get(int device, void **ptr, size_t bytes) {
  State &s = st();
  *ptr = nullptr;
  const bool pooled = s.device_cap != 0 && bytes <= kMaxPooled;
  const size_t cls = pooled ? size_class(bytes) : bytes;
  if (pooled) {
    std::lock_guard<std::mutex> lock(s.mu);
    if (void *p = take(s.device[device], cls)) {
      *ptr = p;
      return hipSuccess;
    }
  }
  MissTimer timer("hipMalloc", cls);
  hipError_t e = hipMalloc(ptr, cls);
  if (e == hipErrorOutOfMemory && release_idle() != 0) {  // the pool itself may be what fills the device
    (void)hipGetLastError();
    e = hipMalloc(ptr, cls);
  }
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(s.mu);
  s.live[*ptr] = pooled ? cls : 0;
  return hipSuccess;
}

void device_put(int device, void *ptr) {
  if (ptr == nullptr) return;
  State &s = st();
  {
    std::lock_guard<std::mutex> lock(s.mu);
    auto it = s.live.find(ptr);
    const size_t cls = it == s.live.end() ? 0 : it->second;
    Shelf &sh = s.device[device];
    if (cls != 0 && sh.idle_bytes + cls <= s.device_cap) {
      sh.idle[cls].push_back(ptr);
      sh.idle_bytes += cls;
      return;
    }
    if (it != s.live.end()) s.live.erase(it);
  }
  (void)hipFree(ptr);
}

hipError_t pinned_get(void **ptr, size_t bytes) {
  State &s = st();
  *ptr = nullptr;
  const bool pooled = s.pinned_cap != 0 && bytes <= kMaxPooled;
  const size_t cls = pooled ? size_class(bytes) : bytes;
  if (pooled) {
    std::lock_guard<std::mutex> lock(s.mu);
    if (void *p = take(s.pinned, cls)) {
      *ptr = p;
      return hipSuccess;
    }
  }
  MissTimer timer("hipHostMalloc", cls);
  const hipError_t e = hipHostMalloc(ptr, cls, kPinnedFlags);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(s.mu);
  s.live[*ptr] = pooled ? cls : 0;
  return hipSuccess;
}

void pinned_put(void *ptr) {
  if (ptr == nullptr) return;
  State &s = st();
  {
    std::lock_guard<std::mutex> lock(s.mu);
    auto it = s.live.find(ptr);
    const size_t cls = it == s.live.end() ? 0 : it->second;
    if (cls != 0 && s.pinned.idle_bytes + cls <= s.pinned_cap) {
      s.pinned.idle[cls].push_back(ptr);
      s.pinned.idle_bytes += cls;
      return;
    }
    if (it != s.live.end()) s.live.erase(it);
  }
  (void)hipHostFree(ptr);
}

// Streams are SHARED, not lent: creating one costs 2.6-8 ms in a process that has made few (a hardware
// queue each; measured from Node, SPEEXHIP_POOL_TRACE=1) -- more than a hundred 64 KiB calls.  A state's
// host-buffer calls are synchronous (each ends with a wait on its stream), so states can take turns on a
// handful of streams: state i of a device runs on stream i % kSharedStreams, made on first use and kept
// for the life of the process.  Four, so that calls of different states on different threads
// (processChunkAsync on the libuv pool) still overlap.
const size_t kSharedStreams = 4;

hipError_t stream_get(int device, hipStream_t *out) {
  State &s = st();
  std::lock_guard<std::mutex> lock(s.mu);  // (held across the creation: at most kSharedStreams times per device)
  auto &v = s.streams[device];
  size_t &next = s.next_stream[device];
  const size_t slot = next++ % kSharedStreams;
  if (slot >= v.size()) {
    MissTimer timer("hipStreamCreate", 0);
    hipStream_t h = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&h, hipStreamNonBlocking);
    if (e != hipSuccess) {
      next--;
      return e;
    }
    v.push_back(h);
    *out = h;
    return hipSuccess;
  }
  *out = v[slot];
  return hipSuccess;
}

void stream_put(int, hipStream_t) {}  // shared: nothing to give back

// A stream of the caller's own (not one of the shared ones; never given back): the many-states stages' copy streams
hipError_t stream_own(int, hipStream_t *out) {
  MissTimer timer("hipStreamCreate (a stage's copy stream)", 0);
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// Warm-up (engine.cpp, warm_device): all of a device's shared streams made ahead of the first state -- the first
// stream of a process costs 20-160 ms (the runtime sets up its hardware queues), each further one ~8 ms
// (profiles/r05_first_call_trace.txt); made here, off the caller's path, a state's creation never meets that.
hipError_t streams_prewarm(int device) {
  State &s = st();
  for (size_t k = 0; k < kSharedStreams; k++) {
    std::lock_guard<std::mutex> lock(s.mu);
    auto &v = s.streams[device];
    if (v.size() >= kSharedStreams) break;
    MissTimer timer("hipStreamCreate (warm-up)", 0);
    hipStream_t h = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&h, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    v.push_back(h);
  }
  return hipSuccess;
}

hipError_t event_get(int device, hipEvent_t *out) {
  State &s = st();
  {
    std::lock_guard<std::mutex> lock(s.mu);
    auto &v = s.events[device];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  MissTimer timer("hipEventCreate", 0);
  return hipEventCreateWithFlags(out, hipEventDisableTiming);
}

void event_put(int device, hipEvent_t h) {
  if (h == nullptr) return;
  State &s = st();
  {
    std::lock_guard<std::mutex> lock(s.mu);
    auto &v = s.events[device];
    if (s.device_cap != 0 && v.size() < 2048) {
      v.push_back(h);
      return;
    }
  }
  (void)hipEventDestroy(h);
}

size_t release_idle() {
  State &s = st();
  std::vector<void *> dev, pin;
  std::vector<hipEvent_t> events;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lock(s.mu);
    for (auto &d : s.device) {
      for (auto &c : d.second.idle)
        for (void *p : c.second) {
          dev.push_back(p);
          s.live.erase(p);
        }
      bytes += d.second.idle_bytes;
      d.second.idle.clear();
      d.second.idle_bytes = 0;
    }
    for (auto &c : s.pinned.idle)
      for (void *p : c.second) {
        pin.push_back(p);
        s.live.erase(p);
      }
    bytes += s.pinned.idle_bytes;
    s.pinned.idle.clear();
    s.pinned.idle_bytes = 0;
    for (auto &v : s.events) {
      events.insert(events.end(), v.second.begin(), v.second.end());
      v.second.clear();
    }
  }
  {  // the slabs of result blocks nobody holds a block of (the next ..._take call makes one again)
    Slabs &ss = slabs();
    std::lock_guard<std::mutex> lock(ss.mu);
    ss.refused_for = 0;
    for (auto it = ss.all.begin(); it != ss.all.end();) {
      if (it->taken.empty()) {
        pin.push_back(it->base);
        bytes += it->bytes;
        ss.total -= it->bytes;
        it = ss.all.erase(it);
      } else {
        ++it;
      }
    }
  }
  for (void *p : dev) (void)hipFree(p);
  for (void *p : pin) (void)hipHostFree(p);
  for (hipEvent_t h : events) (void)hipEventDestroy(h);
  return bytes;
}

}  // namespace pool
}  // namespace speexhip
