/*
 * speexhip_resampler.h -- C ABI of libspeexhip.so, the MI355X (gfx950) implementation of the
 * Speex polyphase-FIR resampler hot path.
 *
 * This is the drop-in boundary for the reference's FFI layer: the five functions the
 * reference exports from its WASM module (scripts/build_emscripten.sh:20) and calls from
 * src/index.ts:6-16, with the signatures of deps/speex/speex_resampler.h.  Symbols carry the
 * prefix `speexhip_` (the reference renames its own with RANDOM_PREFIX for the same reason,
 * deps/speex/speex_resampler.h:50-79).  Plain pointers and sizes only; no HIP/torch types.
 *
 * Numerical contract: output int16 PCM is within +-1 LSB of the reference on the same input
 * (SPEEXHIP_MODE_FAST_FIXED, the default, and SPEEXHIP_MODE_FAST) or bit-identical to it (SPEEXHIP_MODE_EXACT).  Stream
 * bookkeeping (frames consumed / produced per call, resample.c:878-902,968-1036) is always
 * identical to the reference.
 *
 * There is NO CPU fallback: without a usable gfx950 device speexhip_resampler_init() fails
 * with SPEEXHIP_ERR_DEVICE.
 */
#ifndef SPEEXHIP_RESAMPLER_H
#define SPEEXHIP_RESAMPLER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPEEXHIP_API __attribute__((visibility("default")))

/* Error codes: 0..5 are the reference's enum (deps/speex/speex_resampler.h:104-113);
 * 6 is new and reports a HIP runtime/device failure; 7 is returned by the ..._take calls only. */
enum {
  SPEEXHIP_ERR_SUCCESS = 0,
  SPEEXHIP_ERR_ALLOC_FAILED = 1,
  SPEEXHIP_ERR_BAD_STATE = 2,
  SPEEXHIP_ERR_INVALID_ARG = 3,
  SPEEXHIP_ERR_PTR_OVERLAP = 4,
  SPEEXHIP_ERR_OVERFLOW = 5,
  SPEEXHIP_ERR_DEVICE = 6,
  SPEEXHIP_ERR_NO_BLOCK = 7,   /* ..._take: no pinned result block free right now; the state is untouched */
  SPEEXHIP_ERR_MAX_ERROR
};

enum {
  SPEEXHIP_MODE_FAST = 0,      /* +-1 LSB; the filters the reference sums in fp64 (quality 9, 10) in fp64.  Launches that
                                  cannot fill the chip may split an output's sum over several waves (tap-range shares): up
                                  to 2x faster on one-stream calls of long decimators, but the last bit of a sample then
                                  depends on how the stream was cut into calls and on what shared its launch.  Opt-in
                                  since round 6 (set_mode, or SPEEXHIP_MODE=fast) */
  SPEEXHIP_MODE_EXACT = 1,     /* bit-identical arithmetic order */
  SPEEXHIP_MODE_FAST_F32 = 2,  /* FAST with one fp32 FMA chain for every filter (the fast path of rounds 1-3):
                                  narrower than the reference's accumulator at quality 9 and 10, still +-1 LSB */
  SPEEXHIP_MODE_FAST_FIXED = 3 /* THE DEFAULT (round 6).  FAST with a pinned summation order: no tap-range shares, the one
                                  launch-time choice that re-associates an output's sum.  Like the reference (whose output
                                  for 64 KiB chunks equals its output for one chunk) the bytes of a stream do
                                  not depend on how it is cut into chunks, on how many streams share a launch or on the
                                  GPU's size; +-1 LSB of the reference.  Costs nothing on launches that fill the chip
                                  (every BASELINE config); one-stream calls of long decimators run
                                  at the unshared speed (48k -> 11.025k stereo: 30 us against 14 for a 2^20-frame call) */
};

/* Which of the reference's inner kernels the (rates, quality) pair selects
 * (deps/speex/resample.c:647-648,682-698). */
enum {
  SPEEXHIP_KERNEL_DIRECT_SINGLE = 0,
  SPEEXHIP_KERNEL_DIRECT_DOUBLE = 1,
  SPEEXHIP_KERNEL_INTERPOLATE_SINGLE = 2,
  SPEEXHIP_KERNEL_INTERPOLATE_DOUBLE = 3
};

typedef struct SpeexHipResamplerState_ SpeexHipResamplerState;
typedef struct SpeexHipBatch_ SpeexHipBatch;

/* ------------------------------------------------------------------------------------------
 * The reference surface (what src/index.ts binds).
 * ---------------------------------------------------------------------------------------- */

/* Replaces speex_resampler_init (deps/speex/speex_resampler.h:127-131, resample.c:794).
 * Returns NULL and sets *err (INVALID_ARG for nb_channels==0, a zero rate, quality<0 or >10;
 * DEVICE when no gfx950 device is usable; ALLOC_FAILED). */
SPEEXHIP_API SpeexHipResamplerState *speexhip_resampler_init(uint32_t nb_channels, uint32_t in_rate,
                                                             uint32_t out_rate, int quality,
                                                             int *err);

/* Which GPU a state lives on (round 5).  The reference's model is many SpeexResampler instances in one process
 * (src/index.ts:18-45: one shared module, one state per instance); on a node with several MI355X the states of
 * one process spread over them by a process-wide rule read from the environment when a state is made:
 *   SPEEXHIP_DEVICE=k         every state on device k
 *   SPEEXHIP_DEVICES=all      every new state on the device with the fewest live states (round 6; ties in the order k mod
 *                             n, k + 1 mod n, ... for state number k: a fresh process deals its states round-robin)
 *   SPEEXHIP_DEVICES=0,2,5    ... on the (k mod 3)-th listed device
 *   neither                   the calling thread's current HIP device (the behaviour before round 5)
 * speexhip_resampler_init / _init_frac / speexhip_batch_init follow the rule; the ..._init_on forms name the device
 * (device < 0 = the rule).  A state's calls may come from any thread with any current device: every entry point
 * switches to the state's device and back.  Init fails with SPEEXHIP_ERR_DEVICE for a device the node does not have.
 * (SPEEXHIP_ALIAS_DEVICES=n, tests only: n logical devices, logical d on physical d mod the real count -- pools, table
 * caches, streams and this rule key on the logical ordinal, so a 1-GPU box walks the multi-device paths.) */
SPEEXHIP_API int speexhip_device_count(void);   /* usable logical devices; <= 0: none (no CPU fallback) */
/* The one-time costs of a process that the first states would otherwise pay -- the runtime's start (90-180 ms), the
 * pool's four shared streams per device (20 + 3 x 8 ms), the copy engines' first copy (8 ms); a state itself is 0.04 ms
 * of filter design and 0.2 ms of uploads -- paid now, on `device` or (device < 0) on every device the placement rule
 * can choose.  Blocking; call it from a thread of its own beside the application's other start-up work.  The N-API addon
 * does, at import, and SpeexResampler.initPromise resolves behind it -- where the reference compiles its WASM module
 * (src/index.ts:18-19, :31).  Optional: without it the first states pay as before.  Returns an error code. */
SPEEXHIP_API int speexhip_warmup(int device);
SPEEXHIP_API SpeexHipResamplerState *speexhip_resampler_init_on(int device, uint32_t nb_channels, uint32_t in_rate,
                                                                uint32_t out_rate, int quality, int *err);

/* Replaces speex_resampler_destroy (speex_resampler.h:157, resample.c:868). */
SPEEXHIP_API void speexhip_resampler_destroy(SpeexHipResamplerState *st);

/* Replaces speex_resampler_process_interleaved_int (speex_resampler.h:217-221,
 * resample.c:1061).  `in`/`out` are HOST pointers to interleaved s16 frames; *in_len /
 * *out_len are frames per channel: in = available / capacity, out = consumed / written.
 * Synchronous.  `in` may be NULL (zeros), as in the reference. */
SPEEXHIP_API int speexhip_resampler_process_interleaved_int(SpeexHipResamplerState *st,
                                                            const int16_t *in, uint32_t *in_len,
                                                            int16_t *out, uint32_t *out_len);

/* Float I/O entry point: replaces speex_resampler_process_interleaved_float
 * (speex_resampler.h:202-206, resample.c:1038-1059 -> :927-963).  Same stream state as the
 * int16 call (int and float calls may be mixed); samples are not rounded or saturated, and a
 * 160-frame input block is not limited to 1024 outputs (resample.c:943 vs :982-991). */
SPEEXHIP_API int speexhip_resampler_process_interleaved_float(SpeexHipResamplerState *st,
                                                              const float *in, uint32_t *in_len,
                                                              float *out, uint32_t *out_len);

/* Replaces speex_resampler_get_rate (speex_resampler.h:237-239, resample.c:1089). */
SPEEXHIP_API void speexhip_resampler_get_rate(SpeexHipResamplerState *st, uint32_t *in_rate,
                                              uint32_t *out_rate);

/* ------------------------------------------------------------------------------------------
 * The rest of the reference's C API around the path: mid-stream control.
 * Not reachable from src/index.ts, but part of deps/speex/speex_resampler.h.  A change of the
 * filter length keeps the stream continuous exactly as the reference does: the history is
 * re-aligned and, when the filter gets shorter, the frames that no longer fit are kept as
 * pending input ("magic samples", resample.c:727-782, 904-922).  These calls wait for the
 * device (they touch the stream history) -- control plane, not hot path.
 * ---------------------------------------------------------------------------------------- */

/* Replaces speex_resampler_init_frac (speex_resampler.h:133-150, resample.c:799). */
SPEEXHIP_API SpeexHipResamplerState *speexhip_resampler_init_frac(uint32_t nb_channels,
                                                                  uint32_t ratio_num, uint32_t ratio_den,
                                                                  uint32_t in_rate, uint32_t out_rate,
                                                                  int quality, int *err);
/* Replace speex_resampler_set_rate / set_rate_frac / get_ratio (speex_resampler.h:223-262,
 * resample.c:1084-1151).  OVERFLOW when a phase numerator cannot be carried to the new
 * denominator: as in the reference (:1119-1134) get_rate / get_ratio then report the NEW rates and
 * a repeat of the same call returns SUCCESS without doing anything; unlike the reference, which
 * goes on with phase numerators on two denominators, processing continues consistently with the
 * OLD ratio and filter until a later filter change succeeds -- from then on get_rate / get_ratio
 * report the filter in force again. */
SPEEXHIP_API int speexhip_resampler_set_rate(SpeexHipResamplerState *st, uint32_t in_rate, uint32_t out_rate);
SPEEXHIP_API int speexhip_resampler_set_rate_frac(SpeexHipResamplerState *st, uint32_t ratio_num,
                                                  uint32_t ratio_den, uint32_t in_rate, uint32_t out_rate);
SPEEXHIP_API void speexhip_resampler_get_ratio(SpeexHipResamplerState *st, uint32_t *ratio_num,
                                               uint32_t *ratio_den);
/* Replace speex_resampler_set_quality / get_quality (speex_resampler.h:264-277, resample.c:1153-1168). */
SPEEXHIP_API int speexhip_resampler_set_quality(SpeexHipResamplerState *st, int quality);
SPEEXHIP_API void speexhip_resampler_get_quality(SpeexHipResamplerState *st, int *quality);
/* Replace speex_resampler_get_input_latency / get_output_latency (speex_resampler.h:305-315,
 * resample.c:1190-1198). */
SPEEXHIP_API int speexhip_resampler_get_input_latency(SpeexHipResamplerState *st);
SPEEXHIP_API int speexhip_resampler_get_output_latency(SpeexHipResamplerState *st);
/* Replace speex_resampler_skip_zeros / reset_mem (speex_resampler.h:317-332,
 * resample.c:1200-1220).  reset_mem restates the reference's multi-channel behaviour: its
 * single memset run covers channels*(filt_len-1) floats of a buffer whose channel lines are
 * mem_alloc_size apart, so only the first channel(s) are silenced. */
SPEEXHIP_API int speexhip_resampler_skip_zeros(SpeexHipResamplerState *st);
SPEEXHIP_API int speexhip_resampler_reset_mem(SpeexHipResamplerState *st);

/* Replace speex_resampler_process_int / _process_float (speex_resampler.h:169-191,
 * resample.c:968-1036, 927-963): ONE channel of the state, host buffers whose consecutive samples
 * are `input stride` / `output stride` apart (speex_resampler_set/get_input/output_stride,
 * speex_resampler.h:285-307, resample.c:1170-1188; both 1 after init, resample.c:842-843).  As in
 * the reference every channel keeps its own position (last_sample / samp_frac_num /
 * magic_samples, resample.c:135-137), so channels may be advanced unevenly; an interleaved call on
 * such a state then handles channel after channel with the caller's lengths and reports the
 * lengths of the last one (resample.c:1061-1082).  These run the bit-exact kernel on one channel
 * in either mode. */
SPEEXHIP_API int speexhip_resampler_process_int(SpeexHipResamplerState *st, uint32_t channel_index,
                                                const int16_t *in, uint32_t *in_len, int16_t *out,
                                                uint32_t *out_len);
SPEEXHIP_API int speexhip_resampler_process_float(SpeexHipResamplerState *st, uint32_t channel_index,
                                                  const float *in, uint32_t *in_len, float *out,
                                                  uint32_t *out_len);
SPEEXHIP_API void speexhip_resampler_set_input_stride(SpeexHipResamplerState *st, uint32_t stride);
SPEEXHIP_API void speexhip_resampler_get_input_stride(SpeexHipResamplerState *st, uint32_t *stride);
SPEEXHIP_API void speexhip_resampler_set_output_stride(SpeexHipResamplerState *st, uint32_t stride);
SPEEXHIP_API void speexhip_resampler_get_output_stride(SpeexHipResamplerState *st, uint32_t *stride);

/* When a filter change cannot build its filter -- the new filter length overflows
 * (resample.c:620-621, 641-655) or memory runs out (here: device memory) -- the reference keeps
 * its old filter length and history, keeps the NEW rates / ratio / quality, installs
 * resampler_basic_zero (resample.c:561-591, 785-791) and returns RESAMPLER_ERR_ALLOC_FAILED: from
 * then on every processing call writes zeros with the right lengths, moves the counters, and
 * returns ALLOC_FAILED too, until a later set_rate / set_quality succeeds.  Same here. */

/* Replaces speex_resampler_strerror (speex_resampler.h:338, resample.c:1222-1239); same
 * strings for codes 0..4, the reference's "Unknown error..." text for 5 and out-of-range
 * codes, and a HIP message for SPEEXHIP_ERR_DEVICE. */
SPEEXHIP_API const char *speexhip_resampler_strerror(int err);

/* ------------------------------------------------------------------------------------------
 * Extensions (not part of the reference surface).
 * ---------------------------------------------------------------------------------------- */

/* Same call with DEVICE pointers (inputs/outputs resident in HBM).  The work is enqueued on
 * `hip_stream` (a hipStream_t passed as void*; NULL = the default stream) and the call
 * returns without waiting for the GPU.  The stream position advances on the host at once
 * (it is integer arithmetic, independent of the audio), so *in_len / *out_len are final on
 * return.  d_in must stay valid until the stream has executed the call.  Calls on one state are
 * ordered: the next call -- on whatever stream -- first waits for this one on the device (an event
 * recorded on `hip_stream` at that moment), and control calls and destroy wait for `hip_stream` on
 * the host (for this state's last call only, never for the device: other states' launches keep
 * running).  `hip_stream` must therefore stay valid until the state's NEXT call of any kind --
 * or be handed back with speexhip_resampler_release_stream() before the caller destroys it (a
 * destroyed stream's handle cannot be recognised afterwards: this runtime dereferences it). */
SPEEXHIP_API int speexhip_resampler_process_interleaved_int_device(SpeexHipResamplerState *st,
                                                                   const int16_t *d_in,
                                                                   uint32_t *in_len, int16_t *d_out,
                                                                   uint32_t *out_len,
                                                                   void *hip_stream);

/* The host-buffer calls with the result left in a PINNED block owned by the caller afterwards (release it with
 * speexhip_block_release).  Same counters, same samples as speexhip_resampler_process_interleaved_int / _float with
 * *out_len as the capacity; *out_block = NULL when no frame was produced.  The kernel writes the block straight through
 * PCIe, so the samples cross memory once on their way out, not twice (device or pinned buffer, then a copy into the
 * caller's buffer).  Blocks are carved out of one pinned slab (SPEEXHIP_TAKE_MB, default 64 MiB, made by the first
 * such call), never allocated per call: while the caller holds so many blocks that none fits, the call returns
 * SPEEXHIP_ERR_NO_BLOCK WITHOUT touching the state, and the caller makes the copying call instead.  The N-API addon
 * hands the block to JavaScript as an external Buffer: src/index.ts:111-115 returns a fresh Buffer the caller owns,
 * and so does it. */
SPEEXHIP_API int speexhip_resampler_process_interleaved_int_take(SpeexHipResamplerState *st, const int16_t *in,
                                                                 uint32_t *in_len, uint32_t *out_len,
                                                                 int16_t **out_block);
SPEEXHIP_API int speexhip_resampler_process_interleaved_float_take(SpeexHipResamplerState *st, const float *in,
                                                                   uint32_t *in_len, uint32_t *out_len,
                                                                   float **out_block);
SPEEXHIP_API void speexhip_block_release(void *block);

/* Pinned blocks for INPUT (round 6) -- the mirror of the ..._take calls.  speexhip_block_acquire(bytes) hands the caller a
 * block of the same pinned slabs to FILL (NULL: none free right now, or SPEEXHIP_TAKE_MB=0; use an ordinary buffer then);
 * speexhip_block_release gives it back, any time after the calls that read it have returned.  Every host-buffer entry
 * point -- process_interleaved_int / _float, the ..._take forms, process_chunks_*, process_many_* -- recognises such a
 * block (and, for buffers of 256 KB and more, memory the caller pinned itself with hipHostMalloc / hipHostRegister) as
 * its input or output and uses it IN PLACE: the kernel reads the chunk through PCIe while it writes the result through
 * PCIe -- both directions of the link at once, one launch, no staging copy -- where a pageable buffer goes through the
 * runtime's staged copy first.  Bytes and counters are those of the same call on pageable buffers.  In the reference this
 * is the copy of the chunk into the WASM module's heap (src/index.ts:71-92) that a caller avoids by decoding straight
 * into the block.  A block may be refilled as soon as the call that read it has returned (the host-buffer calls are
 * synchronous). */
SPEEXHIP_API void *speexhip_block_acquire(uint64_t bytes);

/* The caller is about to destroy the stream of this state's last device-pointer call (one stream per
 * request, say): what that call still has in flight is ordered behind an event of the state's own and
 * the stream is forgotten -- later calls, control calls and destroy wait for the event instead.  Costs one
 * hipEventRecord (about 3 us of stream time on this stack, which is why it is not done after every call). */
SPEEXHIP_API int speexhip_resampler_release_stream(SpeexHipResamplerState *st);

SPEEXHIP_API int speexhip_resampler_process_interleaved_float_device(SpeexHipResamplerState *st,
                                                                     const float *d_in,
                                                                     uint32_t *in_len, float *d_out,
                                                                     uint32_t *out_len,
                                                                     void *hip_stream);

/* Chunk coalescing: n_chunks CONSECUTIVE host-buffer calls on one stream as
 * ONE transfer + ONE launch.  in[i] / in_len[i] / out_len[i] are what call i of
 * speexhip_resampler_process_interleaved_int would be given (in[i] may be NULL = silence); on
 * return in_len[i] / out_len[i] hold what call i consumed / wrote, and `out` holds the calls'
 * outputs back to back (so it needs room for the sum of the capacities).  Bytes and counters
 * are exactly those of the n_chunks separate calls, including frames a capacity-bound call
 * leaves unconsumed: the host plans every call in integer arithmetic first and only the frames
 * really consumed travel to the GPU. */
SPEEXHIP_API int speexhip_resampler_process_chunks_int(SpeexHipResamplerState *st, uint32_t n_chunks,
                                                       const int16_t *const *in, uint32_t *in_len,
                                                       int16_t *out, uint32_t *out_len);
SPEEXHIP_API int speexhip_resampler_process_chunks_float(SpeexHipResamplerState *st, uint32_t n_chunks,
                                                         const float *const *in, uint32_t *in_len,
                                                         float *out, uint32_t *out_len);

/* Many states, one call (round 5; SURVEY 8b: "a batched entry (array of states/buffers) for multi-stream launches").
 * st[i] is an independent single-stream state -- what one `new SpeexResampler(...)` of the reference holds -- and
 * (in[i], in_len[i], out[i], out_len[i]) are the arguments its own speexhip_resampler_process_interleaved_int call
 * would get: HOST pointers, frames per channel, in = available / capacity, out = consumed / written.  Samples and
 * counters of every state are exactly those of the n separate calls; codes[i] (may be NULL) receives call i's return
 * code and the function returns the first one that is not SUCCESS.  What changes is the shape on the GPU: per device
 * ONE transfer in, ONE launch per <= 32 states that share (rates, quality, channels, mode), ONE transfer out -- the
 * shape BASELINE configs[4] is quoted on -- and states that live on different GPUs (SPEEXHIP_DEVICES=all) run side by
 * side, each over its own PCIe link.  States whose channels the per-channel calls moved apart, states in the zero
 * fallback and a state named twice take their own single call, in order.  Synchronous; a state must not be used
 * by another thread meanwhile. */
SPEEXHIP_API int speexhip_resampler_process_many_int(uint32_t n, SpeexHipResamplerState *const *st,
                                                     const int16_t *const *in, uint32_t *in_len,
                                                     int16_t *const *out, uint32_t *out_len, int *codes);
SPEEXHIP_API int speexhip_resampler_process_many_float(uint32_t n, SpeexHipResamplerState *const *st,
                                                       const float *const *in, uint32_t *in_len,
                                                       float *const *out, uint32_t *out_len, int *codes);

/* What the next processing call WOULD consume and produce for (in_len, out_capacity), without
 * touching the state: the counters are integer functions of the stream position alone, so a
 * binding can size its output buffer exactly before the call: the processing calls write
 * exactly `produced` frames, so a buffer of that size suffices even though *out_len is larger
 * (the N-API addon does this; do NOT pass `produced` as the capacity instead -- a tighter
 * capacity can end the block loop before trailing input is consumed).  float_entry selects the
 * float call's rules. */
SPEEXHIP_API int speexhip_resampler_peek(SpeexHipResamplerState *st, uint32_t in_len, uint32_t out_capacity,
                                         int float_entry, uint32_t *consumed, uint32_t *produced);

/* SPEEXHIP_MODE_FAST_FIXED (default; +-1 LSB, bytes a function of the stream alone), SPEEXHIP_MODE_FAST (+-1 LSB, faster on
 * small launches of long filters, bytes may depend on chunking), SPEEXHIP_MODE_EXACT (bit-identical arithmetic order,
 * slower) or SPEEXHIP_MODE_FAST_F32.  The environment variable SPEEXHIP_MODE=exact|fast|fast_f32|fast_fixed sets the
 * initial mode. */
SPEEXHIP_API int speexhip_resampler_set_mode(SpeexHipResamplerState *st, int mode);

typedef struct SpeexHipInfo {
  uint32_t in_rate, out_rate;
  uint32_t num_rate, den_rate;   /* gcd-reduced ratio, resample.c:1125-1128 */
  uint32_t nb_channels;
  int32_t quality;
  uint32_t filt_len;             /* taps per output, resample.c:616-625 */
  uint32_t oversample;           /* resample.c:615,626-635 */
  uint32_t sinc_table_length;    /* floats, resample.c:652,657 */
  int32_t kernel;                /* SPEEXHIP_KERNEL_* */
  int32_t mode;                  /* SPEEXHIP_MODE_* */
  int32_t fast_path;             /* what the fast modes run for this configuration: 2 = period-lane
                                    kernel, 3 = small-ratio sliding-window kernel, 4 / 5 = their
                                    fp64-accumulate twins (FAST mode, quality 9 and 10), 0 = falls
                                    back to the exact kernel (exotic ratios) */
  int32_t last_sample;           /* stream position, resample.c:135 */
  uint32_t samp_frac_num;        /* stream phase, resample.c:136 */
  int32_t device;                /* HIP device ordinal the state lives on */
  uint32_t magic_samples;        /* pending frames buffered after the history, resample.c:137 */
  uint32_t block_in;             /* frames per block: mem_alloc_size-(filt_len-1), resample.c:935 */
  int32_t accumulate_bits;       /* accumulator of what the current mode runs for this filter: 32 (fp32 FMA chain;
                                    exact kernels of the single kinds) or 64 (v_fma_f64 fast kernels, fast_path 4 / 5;
                                    exact kernels of the double kinds: fp64 sums of fp32 products) */
} SpeexHipInfo;

SPEEXHIP_API int speexhip_resampler_get_info(SpeexHipResamplerState *st, SpeexHipInfo *info);
/* The same with the caller's sizeof(SpeexHipInfo): at most `struct_size` bytes are written, so a caller compiled
 * against an older header (the struct grows at its end: accumulate_bits came in 0.2) is not overrun.  ABI note:
 * 0.2 -> 0.3 adds entry points and SPEEXHIP_MODE_FAST_FIXED; SpeexHipInfo and the error codes are unchanged. */
SPEEXHIP_API int speexhip_resampler_get_info2(SpeexHipResamplerState *st, SpeexHipInfo *info, uint32_t struct_size);

/* Copies the last filt_len-1 consumed frames (interleaved float, the reference's `mem`: what
 * the next call's first outputs are computed from; resample.c:898-899) followed by the
 * magic_samples pending frames to the host.  dst holds (filt_len-1+magic_samples)*ch floats. */
SPEEXHIP_API int speexhip_resampler_get_history(SpeexHipResamplerState *st, float *dst);

/* Batched streams: n_streams independent resamplers with one shared (rates, quality,
 * channels) filter, processed by ONE launch per call.  Device pointers; stream s reads
 * d_in + s*in_stream_stride and writes d_out + s*out_stream_stride (strides in int16
 * elements).  in_len[s] / out_len[s] as in the single-stream call.  Asynchronous on
 * `hip_stream`. */
SPEEXHIP_API SpeexHipBatch *speexhip_batch_init(uint32_t n_streams, uint32_t nb_channels,
                                                uint32_t in_rate, uint32_t out_rate, int quality,
                                                int *err);
SPEEXHIP_API SpeexHipBatch *speexhip_batch_init_on(int device, uint32_t n_streams, uint32_t nb_channels,
                                                   uint32_t in_rate, uint32_t out_rate, int quality, int *err);
SPEEXHIP_API void speexhip_batch_destroy(SpeexHipBatch *b);
SPEEXHIP_API int speexhip_batch_set_mode(SpeexHipBatch *b, int mode);
SPEEXHIP_API int speexhip_batch_get_info(SpeexHipBatch *b, uint32_t stream, SpeexHipInfo *info);
SPEEXHIP_API int speexhip_batch_release_stream(SpeexHipBatch *b);  /* see speexhip_resampler_release_stream */
SPEEXHIP_API int speexhip_batch_process_interleaved_int_device(
    SpeexHipBatch *b, const int16_t *d_in, uint64_t in_stream_stride, uint32_t *in_len,
    int16_t *d_out, uint64_t out_stream_stride, uint32_t *out_len, void *hip_stream);

SPEEXHIP_API int speexhip_batch_process_interleaved_float_device(
    SpeexHipBatch *b, const float *d_in, uint64_t in_stream_stride, uint32_t *in_len, float *d_out,
    uint64_t out_stream_stride, uint32_t *out_len, void *hip_stream);

/* Mid-stream control for every stream of a batch (same semantics as the single-stream calls). */
SPEEXHIP_API int speexhip_batch_set_rate_frac(SpeexHipBatch *b, uint32_t ratio_num, uint32_t ratio_den,
                                              uint32_t in_rate, uint32_t out_rate);
SPEEXHIP_API int speexhip_batch_set_quality(SpeexHipBatch *b, int quality);
SPEEXHIP_API int speexhip_batch_skip_zeros(SpeexHipBatch *b);
SPEEXHIP_API int speexhip_batch_reset_mem(SpeexHipBatch *b);
SPEEXHIP_API int speexhip_batch_get_history(SpeexHipBatch *b, uint32_t stream, float *dst);

/* ------------------------------------------------------------------------------------------
 * Host-only pieces of the path, callable without a GPU (used by the CPU test-suite).
 * ---------------------------------------------------------------------------------------- */

/* Filter design (resample.c:605-702): fills *info (rates, num/den, filt_len, oversample,
 * sinc_table_length, kernel) and, when table != NULL, up to table_capacity floats of the
 * sinc table in the reference's layout.  Returns an error code. */
SPEEXHIP_API int speexhip_design_filter(uint32_t in_rate, uint32_t out_rate, int quality,
                                        SpeexHipInfo *info, float *table, uint32_t table_capacity);

/* Same with the ratio given separately from the nominal rates (init_frac / set_rate_frac). */
SPEEXHIP_API int speexhip_design_filter_frac(uint32_t ratio_num, uint32_t ratio_den, uint32_t in_rate,
                                             uint32_t out_rate, int quality, SpeexHipInfo *info,
                                             float *table, uint32_t table_capacity);

/* One call of the stream bookkeeping (resample.c:878-902 inside the block loop :988-1030) in
 * closed form: given in_len frames, out_cap frames of room and the position (*last_sample,
 * *samp_frac_num), returns frames consumed / produced and advances the position. */
SPEEXHIP_API int speexhip_plan_call(uint32_t num_rate, uint32_t den_rate, uint32_t in_len,
                                    uint32_t out_cap, int32_t *last_sample,
                                    uint32_t *samp_frac_num, uint32_t *consumed,
                                    uint32_t *produced);

/* The same for any entry point and state: float_entry selects the float call's rules
 * (resample.c:927-963: no 1024-output cap, pending frames drained once up front) instead of
 * the int16 call's (:968-1036); block_in = frames per block (160 unless a filter has been
 * shortened mid-stream); *magic_samples = pending frames, consumed ahead of the input. */
SPEEXHIP_API int speexhip_plan_call_ex(uint32_t num_rate, uint32_t den_rate, uint32_t in_len,
                                       uint32_t out_cap, int float_entry, uint32_t block_in,
                                       int32_t *last_sample, uint32_t *samp_frac_num,
                                       uint32_t *magic_samples, uint32_t *consumed, uint32_t *produced);

/* What a filter-length change does to a started stream (resample.c:727-782): afterwards the
 * stream holds new_filt_len-1+*new_magic frames, frame j being old frame j+*shift where that
 * exists (the old line has old_filt_len-1+magic frames) and silence elsewhere; last_sample
 * moves by *last_delta.  *phase (may be NULL) is carried from old_den to new_den
 * (resample.c:1130-1139); returns OVERFLOW if that is not representable. */
SPEEXHIP_API int speexhip_plan_filter_change(uint32_t old_filt_len, uint32_t new_filt_len, uint32_t magic,
                                             int64_t *shift, uint32_t *new_magic, int32_t *last_delta,
                                             uint32_t *phase, uint32_t old_den, uint32_t new_den);

/* Library build info: "speexhip <version> gfx950". */
SPEEXHIP_API const char *speexhip_version(void);

/* One channel's position (extension; the reference keeps these in its private state:
 * last_sample[c], samp_frac_num[c], magic_samples[c], resample.c:135-137). */
SPEEXHIP_API int speexhip_resampler_get_channel_position(SpeexHipResamplerState *st, uint32_t channel,
                                                         int32_t *last_sample, uint32_t *samp_frac_num,
                                                         uint32_t *magic_samples);

/* Host-only (no GPU needed, used by the CPU tests): which fast kernel a (ratio, quality, channel count)
 * gets and with what geometry.  out[0] = fast path (2 period kernel, 3 slide kernel, 0 exact kernel only),
 * period kernel: out[1] = phases per wave (10 / 5), out[2] = periods per tile, out[3] = row length (steps),
 * out[4] = LDS window bytes, out[5] = bank padding (floats per period), out[6] = 1 if a second plan with 5
 * phases per wave serves launches of one generation; slide kernel: out[1] = periods per lane, out[3] = row
 * length (steps), out[4] = LDS bytes of a two-wave workgroup, out[7] = tap steps per iteration; period kernel:
 * out[7] = periods per tile of the int16-window plan that int16 calls take instead (0 = none: the float window
 * already holds a full tile, or the layout has no such plan). */
SPEEXHIP_API int speexhip_debug_plan(uint32_t ratio_num, uint32_t ratio_den, int quality, uint32_t channels,
                                     uint32_t out[8]);

/* ... and the round-4 plans beside it: out[0] = 5 / 4 when FAST runs the configuration through the fp64-accumulate
 * period / slide kernel (quality 9, 10: out[1] = phases per wave / periods per lane, out[2] = periods per tile,
 * out[3] = row length, out[4] = LDS bytes, out[5] = bank padding / row stride, out[6] = trips per row, out[7] = tap
 * steps per iteration of the slide kernel, or -- period kernel -- periods per tile of the int16-window plan that int16
 * calls take instead, 0 = none), 6 when the mono filter also has phase-pair plans (wide windows: same
 * fields as the period plan, out[7] = periods per tile of their int16-window plan), 0 otherwise. */
SPEEXHIP_API int speexhip_debug_plan64(uint32_t ratio_num, uint32_t ratio_den, int quality, uint32_t channels,
                                       uint32_t out[8]);

/* Host only (tests): the shape of the period kernel's launch for a first call of `frames` frames on each of `streams`
 * (<= 32) streams -- out[0] = 1 when the launch takes the phase-pair plan, out[1] = phases per wave (r), out[2] = 1 for
 * the int16 window, out[3] = periods per tile, out[4] = tiles per stream, out[5] = phase-group splits, out[6] = waves
 * with a group, out[7] = tap-range shares (1 = none), out[8] = lanes per workgroup, out[9] = 1 when every workgroup
 * fetches the tap rows into L2 behind its window; all zero when the configuration does not run the fp32 period
 * kernel.  The rules are fitted to a 256-CU device, which is what a process without a GPU assumes. */
SPEEXHIP_API int speexhip_debug_launch_shape(uint32_t ratio_num, uint32_t ratio_den, int quality, uint32_t channels,
                                             uint32_t streams, uint32_t frames, int float_io, uint32_t out[10]);

/* Host only (tests): the placement rule above as a pure function -- the device of state number k on a node with
 * `device_count` devices, env_device / env_devices = the values of SPEEXHIP_DEVICE / SPEEXHIP_DEVICES (NULL = unset),
 * current_device = the calling thread's HIP device.  Returns the device, or -1 when the environment names a device
 * the node does not have or cannot be parsed. */
SPEEXHIP_API int speexhip_debug_placement(int device_count, const char *env_device, const char *env_devices,
                                          uint64_t k, int current_device);

/* Round 6: the placement rule with the load taken into account -- what speexhip_resampler_init really applies.  As
 * speexhip_debug_placement, except that SPEEXHIP_DEVICES=all picks the device with the fewest LIVE states (live[d] for d <
 * device_count; destroying a state frees its slot), ties going to the first such device in the order k mod n, k + 1 mod
 * n, ...: a fresh process still deals its states round-robin, a long-running server fills the holes closed connections
 * leave.  A LIST keeps the counter rule (a reproducible assignment is what the caller asked for).  Host only, pure.
 * speexhip_debug_live_states(d): states alive on logical device d right now. */
SPEEXHIP_API int speexhip_debug_placement_live(int device_count, const char *env_device, const char *env_devices,
                                               uint64_t k, int current_device, const uint32_t *live);
SPEEXHIP_API uint32_t speexhip_debug_live_states(int device);

/* Which kind of box is this?  Runs ~0.3 ms of packed fp32 FMAs with LDS reads on every CU and reports the shader
 * clock (GHz) the chip held meanwhile (median / slowest workgroup): the pool's boxes differ by 4-6 %, so bench
 * lines and the perf gate (tests/test_gpu_perf_gate.py) quote it.  Diagnostics; blocks the calling thread. */
SPEEXHIP_API int speexhip_debug_device_clock(double *ghz_median, double *ghz_min);

/* The PCIe link's own rate on this box (round 6): plain pinned copies of `bytes` host -> device (out_gbs[0]), device ->
 * host (out_gbs[1]) and both at once on two streams (out_gbs[2]: GB/s PER DIRECTION while both run), best of `reps`.
 * The roofline of the host-fed legs of bench.py (`end_to_end`, `end_to_end_streams`).  Diagnostics; blocks. */
SPEEXHIP_API int speexhip_debug_pcie_peak(uint64_t bytes, int reps, double out_gbs[3]);

/* Test hook: the n-th next device allocation made while installing a filter fails, as if the
 * device were out of memory (exercises the resampler_basic_zero fallback, and the release of what
 * an aborted install had already allocated, without exhausting HBM); 0 = off. */
SPEEXHIP_API void speexhip_debug_fail_device_allocs(int n);

/* No counterpart in the reference (its state is plain heap memory, speex_resampler_destroy
 * resample.c:858-868 frees it).  Destroying a state here returns its device buffers, pinned staging
 * buffers, stream and events to a process-wide pool, so that the next state -- callers make one per
 * file or per connection, src/test.ts:27 -- does not pay hipHostMalloc / hipHostFree again
 * (csrc/pool.h; idle memory bounded by SPEEXHIP_POOL_MB, default 1024 MiB device + 256 MiB pinned,
 * 0 = no pool).  This hands everything idle back to the driver; returns the bytes released. */
SPEEXHIP_API uint64_t speexhip_release_cached_memory(void);

#ifdef __cplusplus
}
#endif
#endif /* SPEEXHIP_RESAMPLER_H */
