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
 * (SPEEXHIP_MODE_FAST, default) or bit-identical to it (SPEEXHIP_MODE_EXACT).  Stream
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
 * 6 is new and reports a HIP runtime/device failure. */
enum {
  SPEEXHIP_ERR_SUCCESS = 0,
  SPEEXHIP_ERR_ALLOC_FAILED = 1,
  SPEEXHIP_ERR_BAD_STATE = 2,
  SPEEXHIP_ERR_INVALID_ARG = 3,
  SPEEXHIP_ERR_PTR_OVERLAP = 4,
  SPEEXHIP_ERR_OVERFLOW = 5,
  SPEEXHIP_ERR_DEVICE = 6,
  SPEEXHIP_ERR_MAX_ERROR
};

enum { SPEEXHIP_MODE_FAST = 0, SPEEXHIP_MODE_EXACT = 1 };

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
 * return.  d_in must stay valid until the stream has executed the call. */
SPEEXHIP_API int speexhip_resampler_process_interleaved_int_device(SpeexHipResamplerState *st,
                                                                   const int16_t *d_in,
                                                                   uint32_t *in_len, int16_t *d_out,
                                                                   uint32_t *out_len,
                                                                   void *hip_stream);

SPEEXHIP_API int speexhip_resampler_process_interleaved_float_device(SpeexHipResamplerState *st,
                                                                     const float *d_in,
                                                                     uint32_t *in_len, float *d_out,
                                                                     uint32_t *out_len,
                                                                     void *hip_stream);

/* SPEEXHIP_MODE_FAST (default; +-1 LSB) or SPEEXHIP_MODE_EXACT (bit-identical arithmetic
 * order, slower).  The environment variable SPEEXHIP_MODE=exact|fast sets the initial mode. */
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
  int32_t fast_path;             /* what FAST mode runs for this configuration: 2 = period-lane
                                    kernel, 3 = small-ratio sliding-window kernel, 0 = falls back
                                    to the exact kernel (exotic ratios) */
  int32_t last_sample;           /* stream position, resample.c:135 */
  uint32_t samp_frac_num;        /* stream phase, resample.c:136 */
  int32_t device;                /* HIP device ordinal the state lives on */
} SpeexHipInfo;

SPEEXHIP_API int speexhip_resampler_get_info(SpeexHipResamplerState *st, SpeexHipInfo *info);

/* Copies the last filt_len-1 consumed frames (interleaved float, the reference's `mem`: what
 * the next call's first outputs are computed from; resample.c:898-899) to the host.  dst holds
 * (filt_len-1)*ch floats. */
SPEEXHIP_API int speexhip_resampler_get_history(SpeexHipResamplerState *st, float *dst);

/* Batched streams: n_streams independent resamplers with one shared (rates, quality,
 * channels) filter, processed by ONE launch per call.  Device pointers; stream s reads
 * d_in + s*in_stream_stride and writes d_out + s*out_stream_stride (strides in int16
 * elements).  in_len[s] / out_len[s] as in the single-stream call.  Asynchronous on
 * `hip_stream`. */
SPEEXHIP_API SpeexHipBatch *speexhip_batch_init(uint32_t n_streams, uint32_t nb_channels,
                                                uint32_t in_rate, uint32_t out_rate, int quality,
                                                int *err);
SPEEXHIP_API void speexhip_batch_destroy(SpeexHipBatch *b);
SPEEXHIP_API int speexhip_batch_set_mode(SpeexHipBatch *b, int mode);
SPEEXHIP_API int speexhip_batch_get_info(SpeexHipBatch *b, uint32_t stream, SpeexHipInfo *info);
SPEEXHIP_API int speexhip_batch_process_interleaved_int_device(
    SpeexHipBatch *b, const int16_t *d_in, uint64_t in_stream_stride, uint32_t *in_len,
    int16_t *d_out, uint64_t out_stream_stride, uint32_t *out_len, void *hip_stream);

SPEEXHIP_API int speexhip_batch_process_interleaved_float_device(
    SpeexHipBatch *b, const float *d_in, uint64_t in_stream_stride, uint32_t *in_len, float *d_out,
    uint64_t out_stream_stride, uint32_t *out_len, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Host-only pieces of the path, callable without a GPU (used by the CPU test-suite).
 * ---------------------------------------------------------------------------------------- */

/* Filter design (resample.c:605-702): fills *info (rates, num/den, filt_len, oversample,
 * sinc_table_length, kernel) and, when table != NULL, up to table_capacity floats of the
 * sinc table in the reference's layout.  Returns an error code. */
SPEEXHIP_API int speexhip_design_filter(uint32_t in_rate, uint32_t out_rate, int quality,
                                        SpeexHipInfo *info, float *table, uint32_t table_capacity);

/* One call of the stream bookkeeping (resample.c:878-902 inside the block loop :988-1030) in
 * closed form: given in_len frames, out_cap frames of room and the position (*last_sample,
 * *samp_frac_num), returns frames consumed / produced and advances the position. */
SPEEXHIP_API int speexhip_plan_call(uint32_t num_rate, uint32_t den_rate, uint32_t in_len,
                                    uint32_t out_cap, int32_t *last_sample,
                                    uint32_t *samp_frac_num, uint32_t *consumed,
                                    uint32_t *produced);

/* Library build info: "speexhip <version> gfx950". */
SPEEXHIP_API const char *speexhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPEEXHIP_RESAMPLER_H */
