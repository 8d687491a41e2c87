// tools/san_workers.cpp -- the persistent worker threads of the many-states call (csrc/unit_workers.cpp) and the live-count
// placement rule (csrc/devices_rule.cpp) under ThreadSanitizer (CPU build; the GPU pool offers no sanitizers).  The shape of
// Batch::process_host_many without the GPU: several caller threads at once, each call = units on several (device, lane)
// keys -- the first unit on the caller's own thread, the others as jobs -- every unit under its stage's lock, large units
// with a nested helper job on (device, lane, 1) that waits on the caller's condition variable like the pipelined path;
// then shutdown() and a second life.  Built and run by tests/test_cpu_sanitizers.py.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include "devices.h"
#include "unit_workers.h"
using namespace speexhip;

static std::mutex g_stage_mu[8][2];
static long g_stage_work[8][2];  // touched only under the stage's lock: TSan sees any unit that runs outside it

static void unit(int device, int lane, bool pipelined, std::atomic<long> *done) {
  std::lock_guard<std::mutex> lock(g_stage_mu[device][lane]);
  g_stage_work[device][lane]++;
  if (pipelined) {
    std::mutex mu;
    std::condition_variable cv;
    size_t ready = 0;
    long launched = 0;
    auto helper = [&] {
      for (size_t k = 0; k < 4; k++) {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return ready > k; });
        launched++;
      }
    };
    workers::Ticket t = workers::submit(workers::key_of(device, lane, 1), helper);
    for (size_t k = 0; k < 4; k++) {
      {
        std::lock_guard<std::mutex> l(mu);
        ready = k + 1;
      }
      cv.notify_all();
    }
    if (t != nullptr) workers::wait(t); else helper();
    if (launched != 4) { std::printf("BAD helper\n"); std::exit(1); }
  }
  done->fetch_add(1);
}

static void many_call(std::mt19937 &rng, std::atomic<long> *done) {
  struct U { int device, lane; bool pipelined; };
  std::vector<U> units;
  const int n_dev = 1 + rng() % 8;
  for (int d = 0; d < n_dev; d++) {
    const bool two = rng() % 4 == 0;
    units.push_back(U{d, 0, rng() % 3 == 0});
    if (two) units.push_back(U{d, 1, rng() % 3 == 0});
  }
  std::vector<workers::Ticket> tickets(units.size());
  for (size_t s = 1; s < units.size(); s++) {
    const U u = units[s];
    tickets[s] = workers::submit(workers::key_of(u.device, u.lane, 0), [u, done] { unit(u.device, u.lane, u.pipelined, done); });
  }
  unit(units[0].device, units[0].lane, units[0].pipelined, done);
  for (size_t s = 1; s < units.size(); s++) {
    if (tickets[s] != nullptr) workers::wait(tickets[s]);
    else unit(units[s].device, units[s].lane, units[s].pipelined, done);
  }
}

int main() {
  long expected = 0;
  std::atomic<long> done{0};
  for (int life = 0; life < 2; life++) {
    std::vector<std::thread> callers;
    std::atomic<long> units_asked{0};
    for (int c = 0; c < 6; c++)
      callers.emplace_back([c, life, &done] {
        std::mt19937 rng(100 * life + c);
        for (int i = 0; i < 300; i++) many_call(rng, &done);
      });
    for (auto &t : callers) t.join();
    (void)units_asked;
    if (workers::thread_count() == 0 || workers::thread_count() > 8 * 2 * 2) { std::printf("BAD thread count %zu\n", workers::thread_count()); return 1; }
    // jobs of one key run in submission order
    std::vector<int> order;
    std::mutex omu;
    std::vector<workers::Ticket> ts;
    for (int i = 0; i < 200; i++) ts.push_back(workers::submit(workers::key_of(3, 0, 0), [i, &order, &omu] { std::lock_guard<std::mutex> l(omu); order.push_back(i); }));
    for (auto &t : ts) workers::wait(t);
    for (int i = 0; i < 200; i++) if (order[i] != i) { std::printf("BAD order\n"); return 1; }
    // an exception does not leave the thread; the next job of the key still runs
    workers::Ticket bad = workers::submit(workers::key_of(3, 0, 0), [] { throw 5; });
    workers::wait(bad);
    workers::Ticket good = workers::submit(workers::key_of(3, 0, 0), [] {});
    workers::wait(good);
    if (!workers::failed(bad) || workers::failed(good)) { std::printf("BAD failed()\n"); return 1; }
    workers::shutdown();
    if (workers::thread_count() != 0) { std::printf("BAD shutdown\n"); return 1; }
  }
  long total = 0;
  for (int d = 0; d < 8; d++) for (int l = 0; l < 2; l++) total += g_stage_work[d][l];
  expected = done.load();
  if (total != expected || total == 0) { std::printf("BAD totals %ld %ld\n", total, expected); return 1; }
  // the live-count placement rule from several threads over shared counters (what Batch::setup / ~Batch do)
  std::atomic<uint32_t> live[8];
  for (auto &v : live) v = 0;
  std::vector<std::thread> makers;
  for (int c = 0; c < 4; c++)
    makers.emplace_back([c, &live] {
      std::mt19937 rng(7 + c);
      std::vector<int> mine;
      for (int i = 0; i < 2000; i++) {
        if (!mine.empty() && rng() % 5 < 2) { live[mine.back()].fetch_sub(1); mine.pop_back(); continue; }
        uint32_t snap[8];
        for (int d = 0; d < 8; d++) snap[d] = live[d].load();
        const int d = devices::placement_rule_live(8, nullptr, "all", static_cast<uint64_t>(i), 0, snap);
        if (d < 0 || d >= 8) { std::printf("BAD placement\n"); std::exit(1); }
        live[d].fetch_add(1);
        mine.push_back(d);
      }
      for (int d : mine) live[d].fetch_sub(1);
    });
  for (auto &t : makers) t.join();
  for (auto &v : live) if (v.load() != 0) { std::printf("BAD live counts\n"); return 1; }
  std::printf("workers ok: %ld units\nsanitizer run ok\n", total);
  return 0;
}
