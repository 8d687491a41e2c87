// unit_workers.cpp -- see unit_workers.h.
#include "unit_workers.h"

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

namespace speexhip {
namespace workers {

struct Job {
  std::function<void()> fn;
  std::mutex mu;
  std::condition_variable cv;
  bool done = false, threw = false;
};

namespace {
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Ticket> queue;
  bool stop = false;

  void loop() {
    for (;;) {
      Ticket job;
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;  // (stop, and nothing left to run)
        job = std::move(queue.front());
        queue.pop_front();
      }
      bool threw = false;
      try {
        job->fn();
      } catch (...) {  // (no exception leaves a thread: the submitter reads failed())
        threw = true;
      }
      {
        std::lock_guard<std::mutex> l(job->mu);
        job->done = true;
        job->threw = threw;
      }
      job->cv.notify_all();
    }
  }
};

struct Registry {
  std::mutex mu;
  std::map<uint64_t, std::unique_ptr<Worker>> all;
  ~Registry() { stop_all(); }
  void stop_all() noexcept {
    std::map<uint64_t, std::unique_ptr<Worker>> gone;
    {
      std::lock_guard<std::mutex> l(mu);
      gone.swap(all);
    }
    for (auto &kv : gone) {
      {
        std::lock_guard<std::mutex> l(kv.second->mu);
        kv.second->stop = true;
      }
      kv.second->cv.notify_all();
      if (kv.second->th.joinable()) kv.second->th.join();
    }
  }
};
Registry &registry() {
  static Registry r;  // (destroyed at library unload: joins the -- idle -- threads)
  return r;
}
}  // namespace

Ticket submit(uint64_t key, std::function<void()> fn) noexcept {
  try {
    Ticket job = std::make_shared<Job>();
    job->fn = std::move(fn);
    Registry &r = registry();
    std::lock_guard<std::mutex> l(r.mu);
    std::unique_ptr<Worker> &w = r.all[key];
    if (w == nullptr) {
      std::unique_ptr<Worker> fresh(new Worker());
      Worker *raw = fresh.get();
      fresh->th = std::thread([raw] { raw->loop(); });  // (std::system_error: caught below, nothing was queued)
      w = std::move(fresh);
    }
    {
      std::lock_guard<std::mutex> lw(w->mu);
      w->queue.push_back(job);
    }
    w->cv.notify_one();
    return job;
  } catch (...) {
    // (an entry made by r.all[key] above may be left holding a null pointer: the next submit fills it)
    return Ticket();
  }
}

void wait(const Ticket &t) noexcept {
  if (t == nullptr) return;
  std::unique_lock<std::mutex> l(t->mu);
  t->cv.wait(l, [&] { return t->done; });
}

bool failed(const Ticket &t) noexcept {
  if (t == nullptr) return false;
  std::lock_guard<std::mutex> l(t->mu);
  return t->threw;
}

size_t thread_count() noexcept {
  Registry &r = registry();
  std::lock_guard<std::mutex> l(r.mu);
  size_t n = 0;
  for (auto &kv : r.all) n += kv.second != nullptr ? 1 : 0;
  return n;
}

void shutdown() noexcept { registry().stop_all(); }

}  // namespace workers
}  // namespace speexhip
