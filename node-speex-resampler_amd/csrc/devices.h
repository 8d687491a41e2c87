// devices.h -- which GPU a new stream state lives on (product code, host only).
//
// The reference's model is "many SpeexResampler instances in one process" (src/index.ts:18-45: one
// shared module, one state per instance).  On a node with 8 MI355X the instances of ONE process
// spread over the GPUs here -- each GPU is also a PCIe link of its own, and the host-buffer calls are
// PCIe-bound -- by a rule the environment sets:
//   SPEEXHIP_DEVICE=k          every state on device k
//   SPEEXHIP_DEVICES=all       state number k of the process on device k mod (number of devices)
//   SPEEXHIP_DEVICES=0,2,5     ... on the k mod 3-th of the listed devices
//   (neither)                  the calling thread's current HIP device, as before round 5
// and that the ..._init_on entry points override per state.
//
// "Device" everywhere in the engine is a LOGICAL ordinal.  Normally logical == physical.  The
// diagnostics switch SPEEXHIP_ALIAS_DEVICES=n makes the library see n logical devices, logical d on
// physical d mod (real count): pools, table caches, shared streams and the placement rule all key on
// the logical ordinal, so a 1-GPU box exercises every multi-device code path (tests; never a
// measurement).
#pragma once
#include <cstdint>

namespace speexhip {
namespace devices {

// Number of logical devices (0 when no GPU is visible); -1 = the HIP call itself failed.
int count();
// Physical HIP ordinal of a logical device.
int physical(int logical);
// Logical ordinal of the calling thread's current HIP device (the lowest alias of it); -1 on failure.
int current();

// The pure rule (host only; speexhip_debug_placement): device of state number k given the device
// count and the two environment strings (null = unset).  `current` is the thread's current device.
// Returns the logical ordinal, or -1 when the environment names a device that does not exist / is
// malformed.
int placement_rule(int device_count, const char *env_device, const char *env_devices, uint64_t k, int current);

// The devices the rule can place a state on, given the environment (warm-up): the listed ones, or the one fixed
// device, or the thread's current device; empty when the environment is invalid or no GPU is visible.
void placement_candidates(int *out, int *n, int cap);

// Round 6 -- the rule with the load taken into account (pure, host only; speexhip_debug_placement_live): as
// placement_rule, except that SPEEXHIP_DEVICES=all picks the device with the FEWEST LIVE STATES (live[d], d <
// device_count; null = all zero), ties going to the first such device in the order k mod n, k+1 mod n, ... -- a fresh
// process therefore still deals its states round-robin, and a long-running one that destroys states (connections that
// close) fills the holes instead of piling new states onto whatever the counter points at.  An explicit LIST keeps the
// counter rule: there the caller asked for a reproducible assignment.
int placement_rule_live(int device_count, const char *env_device, const char *env_devices, uint64_t k, int current,
                        const uint32_t *live);

// Device for the next new state of this process by the rule above (advances the state counter);
// -1 = the environment is invalid for this node (init then fails with SPEEXHIP_ERR_DEVICE).
int place_next_state();
// A state now lives on / has left logical device d (Batch::setup, ~Batch): the live counts placement_rule_live reads.
void state_born(int device);
void state_gone(int device);
uint32_t live_states(int device);

}  // namespace devices
}  // namespace speexhip
