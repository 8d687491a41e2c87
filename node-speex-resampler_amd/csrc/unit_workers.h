// unit_workers.h -- the persistent worker threads behind the many-states call (product code; host only, no HIP: the
// sanitizer harness tools/san_workers.cpp builds it with g++ -fsanitize=thread).
//
// Batch::process_host_many runs the GPUs of a node side by side -- a GPU is a PCIe link of its own and the runtime's
// pageable copies keep the issuing thread busy -- and, for large calls, a second "lane" per GPU and a helper that launches
// while the caller copies.  Until round 5 each of those was a std::thread created and joined inside EVERY call: on 8 GPUs
// 7-15 thread creations per step of a server whose steps take 0.2 ms.  Here every (device,
// lane, role) has ONE thread, made when it is first needed and kept for the life of the process; a call hands it a job
// and waits for the job's ticket.  Jobs of one key run in the order they were submitted; different keys run side by side.
// The threads are joined when the library is unloaded (a static object's destructor; they are idle then -- every call
// waits for its own jobs before it returns).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>

namespace speexhip {
namespace workers {

struct Job;  // a submitted job: wait() for it
typedef std::shared_ptr<Job> Ticket;

inline uint64_t key_of(int device, int lane, int role) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(device)) << 16) | (static_cast<uint64_t>(lane & 0xff) << 8) |
         static_cast<uint64_t>(role & 0xff);
}

// Queues `fn` for the thread of `key`.  Never throws: when the job cannot be queued (no memory, no thread to be had) the
// returned ticket is null and the CALLER runs fn itself -- serial, but correct.  fn must not let an exception escape (one
// that does is swallowed and reported by failed()).
Ticket submit(uint64_t key, std::function<void()> fn) noexcept;
// Blocks until the job has run.  A null ticket returns at once.
void wait(const Ticket &t) noexcept;
bool failed(const Ticket &t) noexcept;  // an exception left fn

size_t thread_count() noexcept;  // threads alive (tests)
void shutdown() noexcept;        // joins every thread (library unload; tests).  submit() afterwards makes new ones.

}  // namespace workers
}  // namespace speexhip
