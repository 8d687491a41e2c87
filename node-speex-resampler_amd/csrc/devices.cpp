// devices.cpp -- see devices.h.
#include "devices.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace speexhip {
namespace devices {
namespace {

int real_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return n;
}

int alias_count() {  // SPEEXHIP_ALIAS_DEVICES (diagnostics / tests), 0 = off
  static const int n = [] {
    const char *e = std::getenv("SPEEXHIP_ALIAS_DEVICES");
    const int v = e != nullptr ? std::atoi(e) : 0;
    return v > 0 && v <= 64 ? v : 0;
  }();
  return n;
}

std::atomic<uint64_t> g_states{0};
const int kMaxDevices = 64;
std::atomic<uint32_t> g_live[kMaxDevices];  // states alive per logical device (zero-initialised: static storage)

}  // namespace

int count() {
  const int real = real_count();
  if (real <= 0) return real;
  const int alias = alias_count();
  return alias != 0 ? alias : real;
}

int physical(int logical) {
  if (alias_count() == 0) return logical;
  const int real = real_count();
  return real > 0 ? logical % real : logical;
}

int current() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return d;  // (with aliases: physical d is logical d, its lowest alias)
}

void placement_candidates(int *out, int *n, int cap) {
  *n = 0;
  const int cnt = count();
  if (cnt <= 0) return;
  const char *one = std::getenv("SPEEXHIP_DEVICE"), *many = std::getenv("SPEEXHIP_DEVICES");
  const int cur = current();
  // (the rule itself enumerates them: states 0 .. cnt-1 of a process)
  for (int k = 0; k < cnt && *n < cap; k++) {
    const int d = placement_rule(cnt, one, many, static_cast<uint64_t>(k), cur);
    if (d < 0) return;
    bool seen = false;
    for (int j = 0; j < *n; j++) seen = seen || out[j] == d;
    if (!seen) out[(*n)++] = d;
  }
}

int place_next_state() {
  const int n = count();
  if (n <= 0) return -1;
  const char *one = std::getenv("SPEEXHIP_DEVICE"), *many = std::getenv("SPEEXHIP_DEVICES");
  const bool spread = (one == nullptr || one[0] == '\0') && many != nullptr && many[0] != '\0';
  // (the counter only moves when the rule uses it: a process that never sets SPEEXHIP_DEVICES keeps no history)
  const uint64_t k = spread ? g_states.fetch_add(1) : 0;
  uint32_t live[kMaxDevices];
  for (int d = 0; d < kMaxDevices; d++) live[d] = g_live[d].load(std::memory_order_relaxed);
  return placement_rule_live(std::min(n, kMaxDevices), one, many, k, current(), live);
}

void state_born(int device) {
  if (device >= 0 && device < kMaxDevices) g_live[device].fetch_add(1, std::memory_order_relaxed);
}
void state_gone(int device) {
  if (device >= 0 && device < kMaxDevices) g_live[device].fetch_sub(1, std::memory_order_relaxed);
}
uint32_t live_states(int device) {
  return device >= 0 && device < kMaxDevices ? g_live[device].load(std::memory_order_relaxed) : 0u;
}

}  // namespace devices
}  // namespace speexhip
