#include "stream_plan.h"

namespace speexhip {
namespace {
inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// Outputs k >= 0 with last + (frac + k*num) div den < limit.
inline uint64_t outputs_before(uint32_t num, uint32_t den, int64_t last, uint32_t frac,
                               int64_t limit) {
  if (limit <= last) return 0;
  return ceil_div(static_cast<uint64_t>(limit - last) * den - frac, num);
}
}  // namespace

uint32_t produced_closed_form(uint32_t num, uint32_t den, uint32_t in_frames,
                              uint32_t out_capacity, StreamPos pos) {
  const uint64_t by_input = outputs_before(num, den, pos.last, pos.frac, in_frames);
  return static_cast<uint32_t>(by_input < out_capacity ? by_input : out_capacity);
}

CallPlan plan_call(uint32_t num, uint32_t den, uint32_t in_frames, uint32_t out_capacity,
                   StreamPos pos, uint32_t block_out) {
  CallPlan plan;
  plan.begin = pos;
  int64_t last = pos.last;
  uint64_t frac = pos.frac;
  uint64_t in_left = in_frames, out_left = out_capacity;

  // Fast-forward over whole 160-frame blocks in closed form.  While a block can never emit
  // more than kBlockOut outputs and the call's capacity is not yet in reach, every block
  // consumes exactly kBlockIn frames and ends with its position past the block, so the state
  // after b blocks is the state after P(b) = #outputs starting before frame 160*b.
  const uint64_t per_block_max = ceil_div(static_cast<uint64_t>(kBlockIn) * den, num) + 1;
  if (per_block_max <= block_out && in_left > 2 * kBlockIn && out_left > 2 * per_block_max) {
    uint64_t lo = 0, hi = in_left / kBlockIn - 1;  // keep at least one block for the loop below
    const uint64_t room = out_left - 2 * per_block_max;
    while (lo < hi) {  // largest b with P(b) <= room
      const uint64_t mid = (lo + hi + 1) / 2;
      if (outputs_before(num, den, last, static_cast<uint32_t>(frac),
                         static_cast<int64_t>(mid * kBlockIn)) <= room)
        lo = mid;
      else
        hi = mid - 1;
    }
    if (lo > 0) {
      const uint64_t made = outputs_before(num, den, last, static_cast<uint32_t>(frac),
                                           static_cast<int64_t>(lo * kBlockIn));
      const uint64_t t = frac + made * num;
      last = last + static_cast<int64_t>(t / den) - static_cast<int64_t>(lo * kBlockIn);
      frac = t % den;
      in_left -= lo * kBlockIn;
      out_left -= made;
    }
  }
  while (in_left && out_left) {
    const uint64_t blk_in = in_left < kBlockIn ? in_left : kBlockIn;
    const uint64_t blk_out = out_left < block_out ? out_left : block_out;
    uint64_t made = outputs_before(num, den, last, static_cast<uint32_t>(frac),
                                   static_cast<int64_t>(blk_in));
    if (made > blk_out) made = blk_out;
    const uint64_t t = frac + made * num;
    const int64_t at = last + static_cast<int64_t>(t / den);
    frac = t % den;
    // process_native: if the position fell short of the block, only that much was consumed
    const uint64_t used = at < static_cast<int64_t>(blk_in) ? static_cast<uint64_t>(at) : blk_in;
    last = at - static_cast<int64_t>(used);
    in_left -= used;
    out_left -= made;
  }
  plan.consumed = static_cast<uint32_t>(in_frames - in_left);
  plan.produced = static_cast<uint32_t>(out_capacity - out_left);
  plan.end.last = static_cast<int32_t>(last);
  plan.end.frac = static_cast<uint32_t>(frac);
  return plan;
}

uint32_t phase_index_of(uint32_t num, uint32_t den, uint32_t frac) {
  if (den <= 1) return 0;
  // modular inverse of num mod den by the extended Euclid recurrence
  int64_t r0 = den, r1 = num % den, s0 = 0, s1 = 1;
  while (r1 != 0) {
    const int64_t q = r0 / r1;
    int64_t t = r0 - q * r1; r0 = r1; r1 = t;
    t = s0 - q * s1; s0 = s1; s1 = t;
  }
  int64_t inv = s0 % static_cast<int64_t>(den);
  if (inv < 0) inv += den;
  return static_cast<uint32_t>((static_cast<uint64_t>(inv) * frac) % den);
}

}  // namespace speexhip
