// stream_plan.h -- integer bookkeeping of one processing call (product code, host only).
//
// The reference walks a call in blocks of <=160 input / <=1024 output frames
// (deps/speex/resample.c:988-1030 around process_native :878-902).  None of that changes
// the VALUE of an output sample: sample k of a call is always the FIR of phase
//   phase_k = (frac + k*num) mod den   at window start   pos_k = last + (frac + k*num) div den
// over (history ++ input).  The blocking only decides how many input frames count as
// consumed when the call is capacity-bound.  plan_call() reproduces exactly that, in 64-bit
// integers, so the GPU can compute all outputs of a call in one launch.
#pragma once
#include <cstdint>

namespace speexhip {

struct StreamPos {
  int32_t last = 0;   // resample.c "last_sample": window start of the next output, in frames
  uint32_t frac = 0;  // resample.c "samp_frac_num": phase numerator in [0, den)
};

struct CallPlan {
  uint32_t produced = 0;  // output frames written
  uint32_t consumed = 0;  // input frames that enter the history (the rest is dropped by the
                          // JS wrapper, reference src/index.ts:108 never reads in_len back)
  StreamPos begin, end;
};

static const uint32_t kBlockIn = 160;    // st->buffer_size, resample.c:835
static const uint32_t kBlockOut = 1024;  // FIXED_STACK_ALLOC, resample.c:111

// block_out: outputs one block may emit -- kBlockOut for the int16 entry point (its stack
// buffer, resample.c:982-991), unlimited for the float entry point (resample.c:943).
CallPlan plan_call(uint32_t num, uint32_t den, uint32_t in_frames, uint32_t out_capacity,
                   StreamPos pos, uint32_t block_out = kBlockOut);

// Closed form for `produced` alone (used as a cross-check): outputs whose window starts
// inside the call's input, capped by the capacity.
uint32_t produced_closed_form(uint32_t num, uint32_t den, uint32_t in_frames,
                              uint32_t out_capacity, StreamPos pos);

// k0 in [0,den) with (k0*num) mod den == frac (num, den coprime): shifting the output index
// by k0 makes every stream's phase sequence the canonical r -> (r*num) mod den.
uint32_t phase_index_of(uint32_t num, uint32_t den, uint32_t frac);

}  // namespace speexhip
