// diag.h -- experiment switches are not part of the product.
//
// Rounds 2-5 grew some forty environment switches that force a launch shape, skip a kernel phase or turn a rule off
// for an A/B run (SPEEXHIP_R, _KSPLIT, _SKIP, _PIECES, ...).  Several of them change the bytes a stream produces, and a
// drop-in's output must not depend on a stray variable in the caller's environment -- the reference has no run-time
// switch at all (scripts/build_emscripten.sh:18-19: its options are compile-time).  So they exist only in the
// diagnostics build, `make diag` -> ab/libspeexhip_diag.so (-DSPEEXHIP_DIAG), which tools/ab.sh and the tests that
// force a variant load through SPEEXHIP_LIB_PATH.  In libspeexhip.so every such switch reads as "unset" at compile
// time: no getenv call, no string, no branch -- and the kernels' phase-skipping masks (PeriodParams::skip,
// SlideParams::skip) do not exist.
//
// What the product reads from the environment, all of it placement, memory limits or start-up -- nothing that changes
// a sample: SPEEXHIP_DEVICE, SPEEXHIP_DEVICES, SPEEXHIP_ALIAS_DEVICES (tests), SPEEXHIP_MODE, SPEEXHIP_POOL_MB,
// SPEEXHIP_TAKE_MB, SPEEXHIP_TAKE_MAX_MB (library); SPEEXHIP_NAPI_COPY (addon); SPEEXHIP_NO_WARMUP (index.js).
// tests/test_cpu_host_logic.py asserts that list against the strings of the shipped library.
#pragma once
#include <cstdlib>

#ifdef SPEEXHIP_DIAG
#define SPEEXHIP_DIAG_ENV(name) std::getenv(name)
#define SPEEXHIP_DIAG_SKIP(p, bits) (((p).skip & (bits)) != 0u)
#else
#define SPEEXHIP_DIAG_ENV(name) (static_cast<const char *>(nullptr))
#define SPEEXHIP_DIAG_SKIP(p, bits) false
#endif

namespace speexhip {
// integer value of a diagnostics switch, `unset` when it is not set (always, in the product build)
inline int diag_int(const char *value, int unset) { return value != nullptr ? std::atoi(value) : unset; }
}  // namespace speexhip
