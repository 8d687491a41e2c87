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
  const uint64_t by_input = outputs_before(num, den, pos.last, pos.frac,
                                           static_cast<int64_t>(in_frames) + pos.magic);
  return static_cast<uint32_t>(by_input < out_capacity ? by_input : out_capacity);
}

namespace {
struct Cursor {
  int64_t last;
  uint64_t frac;
};

// One run of the FIR loop over `nin` buffered frames with room for `cap` outputs, then the
// history shift (process_native, resample.c:878-902): returns outputs made, *used = frames
// that entered the history (all of them unless the run was output-bound).
inline uint64_t run_block(uint32_t num, uint32_t den, Cursor &c, uint64_t nin, uint64_t cap,
                          uint64_t *used) {
  uint64_t made = outputs_before(num, den, c.last, static_cast<uint32_t>(c.frac), static_cast<int64_t>(nin));
  if (made > cap) made = cap;
  const uint64_t t = c.frac + made * num;
  const int64_t at = c.last + static_cast<int64_t>(t / den);
  c.frac = t % den;
  *used = at < static_cast<int64_t>(nin) ? static_cast<uint64_t>(at) : nin;
  c.last = at - static_cast<int64_t>(*used);
  return made;
}

// The block loop over real input (resample.c:941-958, 988-1030 without pending frames).
void walk_input(uint32_t num, uint32_t den, Cursor &c, uint64_t &in_left, uint64_t &out_left,
                uint64_t block_in, uint64_t block_out) {
  // Fast-forward over whole blocks in closed form.  While a block can never emit more than
  // block_out outputs and the call's capacity is not yet in reach, every block consumes
  // exactly block_in frames and ends with its position past the block, so the state after b
  // blocks is the state after P(b) = #outputs starting before frame block_in*b.
  const uint64_t per_block_max = ceil_div(block_in * den, num) + 1;
  if (per_block_max <= block_out && in_left > 2 * block_in && out_left > 2 * per_block_max) {
    uint64_t lo = 0, hi = in_left / block_in - 1;  // keep at least one block for the loop below
    const uint64_t room = out_left - 2 * per_block_max;
    while (lo < hi) {  // largest b with P(b) <= room
      const uint64_t mid = (lo + hi + 1) / 2;
      if (outputs_before(num, den, c.last, static_cast<uint32_t>(c.frac),
                         static_cast<int64_t>(mid * block_in)) <= room)
        lo = mid;
      else
        hi = mid - 1;
    }
    if (lo > 0) {
      const uint64_t made = outputs_before(num, den, c.last, static_cast<uint32_t>(c.frac),
                                           static_cast<int64_t>(lo * block_in));
      const uint64_t t = c.frac + made * num;
      c.last = c.last + static_cast<int64_t>(t / den) - static_cast<int64_t>(lo * block_in);
      c.frac = t % den;
      in_left -= lo * block_in;
      out_left -= made;
    }
  }
  while (in_left && out_left) {
    const uint64_t nin = in_left < block_in ? in_left : block_in;
    uint64_t used = 0;
    const uint64_t made = run_block(num, den, c, nin, out_left < block_out ? out_left : block_out, &used);
    in_left -= used;
    out_left -= made;
  }
}
}  // namespace

CallPlan plan_call(uint32_t num, uint32_t den, uint32_t in_frames, uint32_t out_capacity,
                   StreamPos pos, const EntryRules &rules) {
  CallPlan plan;
  plan.begin = pos;
  Cursor c{pos.last, pos.frac};
  uint64_t magic = pos.magic;
  uint64_t in_left = in_frames, out_left = out_capacity;
  const uint64_t block_in = rules.block_in ? rules.block_in : 1;
  const uint64_t block_out = rules.float_entry ? ~0ull : rules.block_out;
  uint64_t made, used;

  if (rules.float_entry) {
    if (magic) {  // resample.c:938-939 (speex_resampler_magic :904-922)
      made = run_block(num, den, c, magic, out_left, &used);
      magic -= used;
      out_left -= made;
    }
  } else {
    while (magic && in_left && out_left) {  // resample.c:988-1030 with pending frames
      uint64_t room = out_left < block_out ? out_left : block_out;
      made = run_block(num, den, c, magic, room, &used);
      magic -= used;
      room -= made;
      out_left -= made;
      if (!magic) {  // the rest of this block's output room goes to the first input block
        made = run_block(num, den, c, in_left < block_in ? in_left : block_in, room, &used);
        in_left -= used;
        out_left -= made;
      }
    }
  }
  if (!magic) walk_input(num, den, c, in_left, out_left, block_in, block_out);

  plan.consumed = static_cast<uint32_t>(in_frames - in_left);
  plan.produced = static_cast<uint32_t>(out_capacity - out_left);
  plan.magic_used = static_cast<uint32_t>(pos.magic - magic);
  plan.end.last = static_cast<int32_t>(c.last);
  plan.end.frac = static_cast<uint32_t>(c.frac);
  plan.end.magic = static_cast<uint32_t>(magic);
  return plan;
}

Realign realign_history(uint32_t old_taps, uint32_t new_taps, uint32_t magic) {
  Realign r;
  r.new_magic = magic;
  if (new_taps > old_taps) {
    // longer filter: the pending frames go back behind `magic` leading zeros ("as if nothing
    // had happened", resample.c:738-747), which makes an augmented line of aug-1 frames ...
    const uint32_t aug = old_taps + 2 * magic;
    if (new_taps > aug) {  // ... still short: left-pad with silence, move the position (:748-758)
      const uint32_t lead = new_taps - aug;
      r.shift = -static_cast<int64_t>(lead) - magic;
      r.new_magic = 0;
      r.last_delta = static_cast<int32_t>(lead / 2);
    } else {  // ... long enough: its last q frames become pending again (:759-764)
      const uint32_t q = (aug - new_taps) / 2;
      r.shift = static_cast<int64_t>(q) - magic;
      r.new_magic = q;
    }
  } else if (new_taps < old_taps) {  // shorter: drop d frames in front, d more become pending (:766-782)
    const uint32_t d = (old_taps - new_taps) / 2;
    r.shift = d;
    r.new_magic = d + magic;
  }
  return r;
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
