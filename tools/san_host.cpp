// tools/san_host.cpp -- the host-only half of the product (filter_design.cpp, stream_plan.cpp) under ASan + UBSan:
// every rate pair of a grid incl. absurd ones, random positions / pending frames / capacities up to 2^32.
// Built and run by tests/test_cpu_sanitizers.py (GPU sanitizers are not available on the pool).
#include <cstdio>
#include <cstdlib>
#include <random>
#include "devices.h"
#include "filter_design.h"
#include "stream_plan.h"
using namespace speexhip;
int main() {
  std::mt19937 rng(1);
  const uint32_t rates[] = {8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 192000, 1, 7, 4000000};
  long plans = 0, filters = 0;
  for (uint32_t a : rates) for (uint32_t b : rates) for (int q = 0; q <= 10; q += 5) {
    FilterSpec f;
    int rc = design_filter(a, b, q, &f, false);  // geometry first ...
    if (rc != 0) continue;
    if (f.table_len <= 20000) rc = design_filter(a, b, q, &f, true);  // ... tables where they are small
    if (rc != 0) continue;
    filters++;
    if (!f.table.empty() && f.den < 2000) {
      std::vector<double> row(f.taps);
      for (uint32_t ph = 0; ph < f.den; ph += (f.den / 7 + 1)) phase_taps(f, ph, row.data());
    }
    for (int k = 0; k < 40; k++) {
      StreamPos p; p.last = (int32_t)(rng() % (f.taps + 5)); p.frac = rng() % f.den; p.magic = (rng() % 4 == 0) ? rng() % 300 : 0;
      EntryRules r; r.float_entry = rng() & 1; r.block_in = 160 + (rng() % 3 == 0 ? rng() % 500 : 0);
      // (full-range lengths only where whole blocks fast-forward in closed form; extreme up-sampling
      //  walks 160-frame blocks one by one and would dominate the run time)
      const uint32_t big = f.num >= f.den ? rng() : rng() % (1u << 20);
      uint32_t in = rng() % 3 ? rng() % 5000 : big; uint32_t cap = rng() % 3 ? rng() % 6000 : rng();
      CallPlan c = plan_call(f.num, f.den, in, cap, p, r);
      if (c.consumed > in || c.produced > cap || c.end.frac >= f.den) { printf("BAD plan\n"); return 1; }
      (void)phase_index_of(f.num, f.den, p.frac);
      (void)produced_closed_form(f.num, f.den, in, cap, p);
      plans++;
    }
    for (uint32_t m = 0; m < 400; m += 37) { Realign g = realign_history(f.taps, f.taps + 8 * (rng() % 40), m); (void)g; g = realign_history(f.taps + 8 * (rng() % 40), f.taps, m); (void)g; }
    uint32_t fr = rng() % f.den; (void)scale_phase(&fr, 1 + rng() % 100000, f.den);
  }
  // the placement rule (devices_rule.cpp) on well-formed, malformed and random environment strings
  long placements = 0;
  const char *envs[] = {nullptr, "", "all", "0", "7", "8", "-1", "0,1,2", "0,,1", ",", "3, 1 ,2", "all,1", "99999", "1,99999",
                        "0,1,2,3,4,5,6,7,0,1,2,3,4,5,6,7,0,1,2,3,4,5,6,7", "x", " ", "1 2", "0x1"};
  for (int count : {-1, 0, 1, 2, 8, 64})
    for (const char *one : envs)
      for (const char *many : envs)
        for (uint64_t k : {0ull, 1ull, 7ull, 255ull, ~0ull}) {
          const int d = devices::placement_rule(count, one, many, k, static_cast<int>(k % 9) - 1);
          if (d < -1 || d >= (count > 0 ? count : 1)) { printf("BAD placement %d of %d\n", d, count); return 1; }
          placements++;
        }
  for (int n = 0; n < 20000; n++) {
    char buf[24];
    const int len = rng() % 23;
    for (int i = 0; i < len; i++) buf[i] = "0123456789, al-x"[rng() % 16];
    buf[len] = 0;
    const int d = devices::placement_rule(8, (rng() & 1) ? buf : nullptr, buf, rng(), rng() % 8);
    if (d < -1 || d >= 8) { printf("BAD placement\n"); return 1; }
    placements++;
  }
  printf("sanitizer run ok: %ld filters, %ld plans, %ld placements\n", filters, plans, placements);
  return 0;
}
