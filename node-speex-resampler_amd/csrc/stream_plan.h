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
  int32_t last = 0;    // resample.c "last_sample": window start of the next output, in frames
  uint32_t frac = 0;   // resample.c "samp_frac_num": phase numerator in [0, den)
  uint32_t magic = 0;  // resample.c "magic_samples": buffered input frames (stored right after
                       // the history) left over from a filter-length change, consumed first
};

struct CallPlan {
  uint32_t produced = 0;    // output frames written
  uint32_t consumed = 0;    // input frames that enter the history (the rest is dropped by the
                            // JS wrapper, reference src/index.ts:108 never reads in_len back)
  uint32_t magic_used = 0;  // pending frames consumed ahead of the input
  StreamPos begin, end;
};

static const uint32_t kBlockIn = 160;    // st->buffer_size, resample.c:835
static const uint32_t kBlockOut = 1024;  // FIXED_STACK_ALLOC, resample.c:111

// How the entry point walks a call.  block_in = frames staged per block: mem_alloc_size -
// (filt_len-1), i.e. 160 until a filter has been shortened mid-stream (the buffer is
// grow-only, resample.c:709-720).  The int16 entry point emits at most block_out outputs per
// block (its stack buffer, resample.c:982-991) and drains pending frames inside its block loop
// (:994-998); the float entry point has no output cap (:943) and drains them once, up front,
// even when the call brings no input (:938-939).
struct EntryRules {
  uint32_t block_in = kBlockIn;
  uint32_t block_out = kBlockOut;
  bool float_entry = false;
};

CallPlan plan_call(uint32_t num, uint32_t den, uint32_t in_frames, uint32_t out_capacity,
                   StreamPos pos, const EntryRules &rules = EntryRules());

// Upper bound of `produced` in closed form: outputs whose window starts inside the pending
// frames and the call's input, capped by the capacity (buffer sizing, cross-checks).
uint32_t produced_closed_form(uint32_t num, uint32_t den, uint32_t in_frames,
                              uint32_t out_capacity, StreamPos pos);

// What a change of the filter length does to one stream (resample.c:727-782).  The stream
// holds `old_taps-1+magic` frames (history ++ pending); afterwards it holds
// `new_taps-1+new_magic` frames with  new[j] = old[j+shift]  where that index exists, silence
// elsewhere, and its position moves by last_delta.
struct Realign {
  int64_t shift = 0;
  uint32_t new_magic = 0;
  int32_t last_delta = 0;
};
Realign realign_history(uint32_t old_taps, uint32_t new_taps, uint32_t magic);

// k0 in [0,den) with (k0*num) mod den == frac (num, den coprime): shifting the output index
// by k0 makes every stream's phase sequence the canonical r -> (r*num) mod den.
uint32_t phase_index_of(uint32_t num, uint32_t den, uint32_t frac);

}  // namespace speexhip
