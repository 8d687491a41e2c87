/*
 * speex_hip_napi.c -- thin raw N-API (C) addon that binds libspeexhip's C ABI
 * (include/speexhip_resampler.h) for index.js.  It REPLACES the Emscripten module
 * src/speex_wasm.js of the reference: where src/index.ts:59-102 called
 * Module._speex_resampler_init / _process_interleaved_int / _strerror on the WASM heap,
 * index.js calls init() / process() here with Buffers.  No arithmetic happens in this file.
 *
 * Built with plain gcc against /usr/include/node (N-API v3+, no node-addon-api).
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "speexhip_resampler.h"

#define NAPI_OK(call)                                            \
  do {                                                           \
    if ((call) != napi_ok) {                                     \
      napi_throw_error(env, NULL, "speexhip N-API failure: " #call); \
      return NULL;                                               \
    }                                                            \
  } while (0)

static void finalize_state(napi_env env, void *data, void *hint) {
  (void)env;
  (void)hint;
  speexhip_resampler_destroy((SpeexHipResamplerState *)data);
}

static SpeexHipResamplerState *unwrap(napi_env env, napi_value v) {
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || p == NULL) {
    napi_throw_type_error(env, NULL, "expected a resampler handle");
    return NULL;
  }
  return (SpeexHipResamplerState *)p;
}

/* init(channels, inRate, outRate, quality) -> handle; throws Error(strerror(code)) like
 * src/index.ts:63-65 */
static napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  uint32_t ch = 0, in_rate = 0, out_rate = 0;
  int32_t quality = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[0], &ch));
  NAPI_OK(napi_get_value_uint32(env, argv[1], &in_rate));
  NAPI_OK(napi_get_value_uint32(env, argv[2], &out_rate));
  NAPI_OK(napi_get_value_int32(env, argv[3], &quality));
  int err = 0;
  SpeexHipResamplerState *st = speexhip_resampler_init(ch, in_rate, out_rate, quality, &err);
  if (st == NULL) {
    napi_throw_error(env, NULL, speexhip_resampler_strerror(err));
    return NULL;
  }
  napi_value handle;
  NAPI_OK(napi_create_external(env, st, finalize_state, NULL, &handle));
  return handle;
}

/* process(handle, chunk: Buffer, inFrames, outCapacityFrames) -> Buffer (fresh copy of the
 * frames written), the src/index.ts:90-115 sequence without the WASM heap. */
static napi_value Process(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  SpeexHipResamplerState *st = unwrap(env, argv[0]);
  if (st == NULL) return NULL;
  void *in_data = NULL;
  size_t in_bytes = 0;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &in_data, &in_bytes));
  uint32_t in_len = 0, out_len = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[2], &in_len));
  NAPI_OK(napi_get_value_uint32(env, argv[3], &out_len));
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si);
  if ((size_t)in_len * si.nb_channels * 2 > in_bytes) {
    napi_throw_range_error(env, NULL, "input frame count exceeds the chunk");
    return NULL;
  }
  size_t cap_bytes = (size_t)out_len * si.nb_channels * 2;
  int16_t *tmp = (int16_t *)malloc(cap_bytes ? cap_bytes : 2);
  if (tmp == NULL) {
    napi_throw_error(env, NULL, speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED));
    return NULL;
  }
  int rc = speexhip_resampler_process_interleaved_int(st, (const int16_t *)in_data, &in_len, tmp,
                                                      &out_len);
  if (rc != 0) {
    free(tmp);
    napi_throw_error(env, NULL, speexhip_resampler_strerror(rc));
    return NULL;
  }
  napi_value out;
  void *copied = NULL;
  napi_status s = napi_create_buffer_copy(env, (size_t)out_len * si.nb_channels * 2, tmp, &copied, &out);
  free(tmp);
  NAPI_OK(s);
  return out;
}

static napi_value SetMode(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  SpeexHipResamplerState *st = unwrap(env, argv[0]);
  if (st == NULL) return NULL;
  int32_t mode = 0;
  NAPI_OK(napi_get_value_int32(env, argv[1], &mode));
  int rc = speexhip_resampler_set_mode(st, mode);
  if (rc != 0) napi_throw_error(env, NULL, speexhip_resampler_strerror(rc));
  return NULL;
}

static napi_value GetInfo(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  SpeexHipResamplerState *st = unwrap(env, argv[0]);
  if (st == NULL) return NULL;
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si);
  napi_value obj, v;
  NAPI_OK(napi_create_object(env, &obj));
#define PUT_U32(name)                                     \
  NAPI_OK(napi_create_uint32(env, si.name, &v));          \
  NAPI_OK(napi_set_named_property(env, obj, #name, v));
#define PUT_I32(name)                                     \
  NAPI_OK(napi_create_int32(env, si.name, &v));           \
  NAPI_OK(napi_set_named_property(env, obj, #name, v));
  PUT_U32(in_rate) PUT_U32(out_rate) PUT_U32(num_rate) PUT_U32(den_rate) PUT_U32(nb_channels)
  PUT_I32(quality) PUT_U32(filt_len) PUT_U32(oversample) PUT_U32(sinc_table_length) PUT_I32(kernel)
  PUT_I32(mode) PUT_I32(fast_path) PUT_I32(last_sample) PUT_U32(samp_frac_num) PUT_I32(device)
  return obj;
}

/* getRate(handle) -> [inRate, outRate]  (speex_resampler_get_rate; exported but unused by
 * src/index.ts, kept for surface parity) */
static napi_value GetRate(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  SpeexHipResamplerState *st = unwrap(env, argv[0]);
  if (st == NULL) return NULL;
  uint32_t a = 0, b = 0;
  speexhip_resampler_get_rate(st, &a, &b);
  napi_value arr, v;
  NAPI_OK(napi_create_array_with_length(env, 2, &arr));
  NAPI_OK(napi_create_uint32(env, a, &v));
  NAPI_OK(napi_set_element(env, arr, 0, v));
  NAPI_OK(napi_create_uint32(env, b, &v));
  NAPI_OK(napi_set_element(env, arr, 1, v));
  return arr;
}

static napi_value StrError(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t code = 0;
  NAPI_OK(napi_get_value_int32(env, argv[0], &code));
  napi_value s;
  NAPI_OK(napi_create_string_utf8(env, speexhip_resampler_strerror(code), NAPI_AUTO_LENGTH, &s));
  return s;
}

static napi_value Version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value s;
  NAPI_OK(napi_create_string_utf8(env, speexhip_version(), NAPI_AUTO_LENGTH, &s));
  return s;
}

static napi_value ModuleInit(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"init", NULL, Init, NULL, NULL, NULL, napi_default, NULL},
      {"process", NULL, Process, NULL, NULL, NULL, napi_default, NULL},
      {"setMode", NULL, SetMode, NULL, NULL, NULL, napi_default, NULL},
      {"getInfo", NULL, GetInfo, NULL, NULL, NULL, napi_default, NULL},
      {"getRate", NULL, GetRate, NULL, NULL, NULL, napi_default, NULL},
      {"strerror", NULL, StrError, NULL, NULL, NULL, napi_default, NULL},
      {"version", NULL, Version, NULL, NULL, NULL, napi_default, NULL},
  };
  if (napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props) != napi_ok)
    napi_throw_error(env, NULL, "speexhip: cannot define exports");
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
