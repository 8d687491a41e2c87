// pool.h -- process-wide recycling of what a state holds on the device (product code).
//
// The reference's state is a few hundred bytes of heap: callers create one per file or per
// connection and drop it (src/test.ts:27, one `new SpeexResampler` per file).  Here a state owns
// device buffers, pinned staging buffers, a stream and events, and the HIP calls that make and
// release those -- hipStreamCreate, hipHostMalloc / hipHostFree above all -- cost more than resampling
// a whole 10-second file: a fresh state's first 1.7 MB call took 0.5-1.0 ms and its destruction 1.0-1.5 ms
// against 0.2 ms for the call itself (tools/init_cost.py).  So nothing goes back to the driver when
// a state dies: buffers, streams and events return here, rounded to size classes, and the next state
// takes them.  Idle memory is bounded (SPEEXHIP_POOL_MB, default 1024 device + 256 pinned; 0 turns
// the pool off) and speexhip_release_cached_memory() hands everything idle back.
//
// A buffer may only be put back once nothing in flight touches it: callers synchronise first
// (~Batch and the filter changes do).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace speexhip {
namespace pool {

// Device memory on `device` (the calling thread's current device must be `device`).
hipError_t device_get(int device, void **ptr, size_t bytes);
void device_put(int device, void *ptr);  // nullptr is fine

// Pinned host memory (hipHostMallocDefault).
hipError_t pinned_get(void **ptr, size_t bytes);
void pinned_put(void *ptr);

// A non-blocking stream of `device` for a state's synchronous host-buffer calls.  Shared with other
// states (one of four per device, kept for the life of the process: see pool.cpp), so anything queued
// on it must be waited for by the call that queued it.
hipError_t stream_get(int device, hipStream_t *s);
void stream_put(int device, hipStream_t s);
// ... and a stream that is NOT shared (made now, on the calling thread's current device = `device`; kept by the caller)
hipError_t stream_own(int device, hipStream_t *s);
hipError_t streams_prewarm(int device);  // make the device's shared streams now (warm-up)
// A timing-disabled event of `device` (lent, returned with event_put).
hipError_t event_get(int device, hipEvent_t *e);
void event_put(int device, hipEvent_t e);

// Result blocks that a caller OWNS after a host-buffer call (Batch::process_host_take; the N-API addon's external
// Buffers): carved out of pinned slabs of SPEEXHIP_TAKE_MB (default 64 MiB, 0 = none) -- the first made by the first
// request, another whenever all are full, up to SPEEXHIP_TAKE_MAX_MB (256 MiB) in all -- so that a call pays
// hipHostMalloc for its result at most four times in the life of a process.  A caller that keeps its blocks
// (JavaScript Buffers wait for the garbage collector AND for the event loop to run their finalizers) exhausts the
// slabs instead of growing pinned memory without bound, and block_get says so: false = nothing free right now (the
// caller falls back to the copying call).  block_put: false = not one of these blocks.
bool block_get(void **ptr, size_t bytes);
bool block_put(void *ptr);
// Round 6: the same blocks serve as INPUT buffers a caller fills (speexhip_block_acquire), and the host-buffer calls
// recognise them: true when [ptr, ptr + bytes) lies inside one slab -- pinned memory the kernels read and write
// straight through PCIe, so no staging copy is made for it.  A range check under the slabs' lock, no runtime call.
bool block_owns(const void *ptr, size_t bytes);

// Returns everything idle to the driver; the number of bytes released.
size_t release_idle();

// Rounded size a request of `bytes` is served with (tests).
size_t size_class(size_t bytes);

}  // namespace pool
}  // namespace speexhip
