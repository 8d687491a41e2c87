// devices_rule.cpp -- the placement rule of devices.h as a pure function (no HIP: built into the sanitizer harness too).
#include "devices.h"

#include <cstring>
#include <string>
#include <vector>

namespace speexhip {
namespace devices {
namespace {
// "3" -> 3; anything else (empty, signs, trailing text) -> -1
int parse_ordinal(const std::string &s) {
  if (s.empty() || s.size() > 4) return -1;
  int v = 0;
  for (char c : s) {
    if (c < '0' || c > '9') return -1;
    v = v * 10 + (c - '0');
  }
  return v;
}

}  // namespace

int placement_rule(int device_count, const char *env_device, const char *env_devices, uint64_t k, int current_device) {
  if (device_count <= 0) return -1;
  if (env_device != nullptr && env_device[0] != '\0') {
    const int d = parse_ordinal(env_device);
    return d >= 0 && d < device_count ? d : -1;
  }
  if (env_devices != nullptr && env_devices[0] != '\0') {
    if (std::strcmp(env_devices, "all") == 0) return static_cast<int>(k % static_cast<uint64_t>(device_count));
    std::vector<int> list;
    std::string item;
    for (const char *p = env_devices;; p++) {
      if (*p == ',' || *p == '\0') {
        const int d = parse_ordinal(item);
        if (d < 0 || d >= device_count) return -1;
        list.push_back(d);
        item.clear();
        if (*p == '\0') break;
      } else if (*p != ' ') {
        item.push_back(*p);
      }
    }
    return list[k % list.size()];
  }
  return current_device >= 0 && current_device < device_count ? current_device : -1;
}

int placement_rule_live(int device_count, const char *env_device, const char *env_devices, uint64_t k, int current_device,
                        const uint32_t *live) {
  const bool one = env_device != nullptr && env_device[0] != '\0';
  if (device_count > 0 && !one && env_devices != nullptr && std::strcmp(env_devices, "all") == 0 && live != nullptr) {
    const uint64_t n = static_cast<uint64_t>(device_count);
    int best = -1;
    for (uint64_t j = 0; j < n; j++) {
      const int d = static_cast<int>((k + j) % n);
      if (best < 0 || live[d] < live[best]) best = d;
    }
    return best;
  }
  return placement_rule(device_count, env_device, env_devices, k, current_device);
}

}  // namespace devices
}  // namespace speexhip
