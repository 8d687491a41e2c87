/*
 * speex_hip_napi.c -- thin raw N-API (C) addon that binds libspeexhip's C ABI
 * (include/speexhip_resampler.h) for index.js.  It REPLACES the Emscripten module
 * src/speex_wasm.js of the reference: where src/index.ts:59-102 called
 * Module._speex_resampler_init / _process_interleaved_int / _strerror on the WASM heap,
 * index.js calls init() / process() here with Buffers.  No arithmetic happens in this file.
 *
 * Beyond the reference surface it carries the SURVEY 8(f) rows: float I/O (N2), mid-stream
 * control (N3), explicit destroy (N4), chunk coalescing and an asynchronous call that keeps
 * the event loop free while the GPU works (N1).
 *
 * Built with plain gcc against /usr/include/node (N-API v3+, no node-addon-api).
 */
#include <node_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
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

/* What a JS handle points at.  `st` becomes NULL on destroy(); the lock serialises calls that
 * reach one state from libuv pool threads (processAsync). */
typedef struct {
  SpeexHipResamplerState *st;
  pthread_mutex_t lock;
} Handle;

static void finalize_state(napi_env env, void *data, void *hint) {
  (void)env;
  (void)hint;
  Handle *h = (Handle *)data;
  if (h->st != NULL) speexhip_resampler_destroy(h->st);
  pthread_mutex_destroy(&h->lock);
  free(h);
}

static Handle *unwrap_handle(napi_env env, napi_value v) {
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || p == NULL) {
    napi_throw_type_error(env, NULL, "expected a resampler handle");
    return NULL;
  }
  return (Handle *)p;
}

/* Every entry point that touches a state holds the handle's lock from before it reads anything of
 * the state until it is done with it: a processAsync job on a libuv pool thread may be running the
 * same state, and everything below -- counters (peek), filter data (setRate frees and rebuilds
 * device tables), even the st pointer (destroy) -- changes under such a job.  Returns the state
 * with the lock HELD, or NULL (exception pending, lock released). */
static SpeexHipResamplerState *lock_state(napi_env env, napi_value v, Handle **hp) {
  Handle *h = unwrap_handle(env, v);
  if (h == NULL) return NULL;
  pthread_mutex_lock(&h->lock);
  if (h->st == NULL) {
    pthread_mutex_unlock(&h->lock);
    napi_throw_error(env, NULL, speexhip_resampler_strerror(SPEEXHIP_ERR_BAD_STATE));
    return NULL;
  }
  *hp = h;
  return h->st;
}
#define UNLOCK(h) pthread_mutex_unlock(&(h)->lock)
/* NAPI_OK for use while the lock is held */
#define NAPI_OK_LOCKED(h, call)                                     \
  do {                                                              \
    if ((call) != napi_ok) {                                        \
      UNLOCK(h);                                                    \
      napi_throw_error(env, NULL, "speexhip N-API failure: " #call); \
      return NULL;                                                  \
    }                                                               \
  } while (0)

/* init(channels, inRate, outRate, quality) -> handle; throws Error(strerror(code)) like
 * src/index.ts:63-65 */
static napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  uint32_t ch = 0, in_rate = 0, out_rate = 0;
  int32_t quality = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[0], &ch));
  NAPI_OK(napi_get_value_uint32(env, argv[1], &in_rate));
  NAPI_OK(napi_get_value_uint32(env, argv[2], &out_rate));
  NAPI_OK(napi_get_value_int32(env, argv[3], &quality));
  /* optional fifth argument: the GPU (speexhip_resampler_init_on); absent / negative = the library's placement rule
   * (SPEEXHIP_DEVICE, SPEEXHIP_DEVICES=all: instance k of the process on GPU k mod the device count) */
  int32_t device = -1;
  if (argc >= 5) {
    napi_valuetype t;
    if (napi_typeof(env, argv[4], &t) == napi_ok && t == napi_number) NAPI_OK(napi_get_value_int32(env, argv[4], &device));
  }
  int err = 0;
  SpeexHipResamplerState *st = speexhip_resampler_init_on(device, ch, in_rate, out_rate, quality, &err);
  if (st == NULL) {
    napi_throw_error(env, NULL, speexhip_resampler_strerror(err));
    return NULL;
  }
  Handle *h = (Handle *)malloc(sizeof(Handle));
  if (h == NULL) {
    speexhip_resampler_destroy(st);
    napi_throw_error(env, NULL, speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED));
    return NULL;
  }
  h->st = st;
  pthread_mutex_init(&h->lock, NULL);
  napi_value handle;
  NAPI_OK(napi_create_external(env, h, finalize_state, NULL, &handle));
  return handle;
}

/* destroy(handle): release the device state now instead of at garbage collection (the
 * reference never calls speex_resampler_destroy: SURVEY 8f row N4). */
static napi_value Destroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = unwrap_handle(env, argv[0]);
  if (h == NULL) return NULL;
  pthread_mutex_lock(&h->lock);
  if (h->st != NULL) speexhip_resampler_destroy(h->st);
  h->st = NULL;
  pthread_mutex_unlock(&h->lock);
  return NULL;
}

/* Optional Buffer argument: null / undefined -> NULL pointer (the reference's in == NULL). */
static int buffer_or_null(napi_env env, napi_value v, void **data, size_t *bytes) {
  napi_valuetype t;
  *data = NULL;
  *bytes = 0;
  if (napi_typeof(env, v, &t) != napi_ok) return 0;
  if (t == napi_null || t == napi_undefined) return 1;
  return napi_get_buffer_info(env, v, data, bytes) == napi_ok;
}

/* external Buffers over pinned blocks of the library (process_common) */
static const size_t kExternalMin = 4096;            /* below this a copy is cheaper than a finalizer */
/* (process-wide: the addon may be loaded by several environments -- the main thread and worker_threads -- and
 *  finalizers run on whichever thread owns the Buffer: every access is an atomic builtin) */
static size_t g_external_bytes = 0;
static size_t g_take_calls = 0, g_take_no_block = 0; /* stats(): results left in pinned blocks / slab full, copied instead */
static size_t g_pinned_chunks = 0;                   /* stats(): allocChunk() calls served with a pinned block */
static int g_no_external = 0; /* SPEEXHIP_NAPI_COPY=1: always copy (A/B, tests) */
static void finalize_block(napi_env env, void *data, void *hint) {
  const size_t bytes = (size_t)hint;
  int64_t adjusted = 0;
  speexhip_block_release(data);
  __atomic_fetch_sub(&g_external_bytes, bytes, __ATOMIC_RELAXED);
  (void)napi_adjust_external_memory(env, -(int64_t)bytes, &adjusted);
}

/* process / processFloat (handle, chunk: Buffer|null, inFrames, outCapacityFrames) -> Buffer
 * (fresh copy of the frames written): the src/index.ts:90-115 sequence without the WASM heap. */
static napi_value process_common(napi_env env, napi_callback_info info, size_t sample_bytes) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void *in_data = NULL;
  size_t in_bytes = 0;
  if (!buffer_or_null(env, argv[1], &in_data, &in_bytes)) {
    napi_throw_type_error(env, NULL, "chunk must be a Buffer or null");
    return NULL;
  }
  uint32_t in_len = 0, out_len = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[2], &in_len));
  NAPI_OK(napi_get_value_uint32(env, argv[3], &out_len));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si);
  const size_t frame_bytes = (size_t)si.nb_channels * sample_bytes;
  if (in_data != NULL && (size_t)in_len * frame_bytes > in_bytes) {
    UNLOCK(h);
    napi_throw_range_error(env, NULL, "input frame count exceeds the chunk");
    return NULL;
  }
  /* the counters are known before any GPU work: size the result exactly and let the library write
   * straight into it (one copy instead of three).  peek and the call see the same state: the lock
   * is held from before the peek until the call has returned. */
  uint32_t will_use = 0, will_make = 0;
  speexhip_resampler_peek(st, in_len, out_len, sample_bytes == 4, &will_use, &will_make);
  const size_t made_bytes = (size_t)will_make * frame_bytes;
  /* Round 4: results of a few KB and more stay where the kernel wrote them -- a pinned block of the library's
   * pool -- and become an EXTERNAL Buffer (a fresh, caller-owned Buffer like src/index.ts:111-115 returns; its
   * finalizer hands the block back to the pool): the samples cross memory once on their way out instead of twice.
   * Bounded: the blocks come out of one pinned slab (SPEEXHIP_TAKE_MB, 64 MiB); while JavaScript holds so many of
   * them that none fits (Buffers kept alive, or a garbage collector that has not run yet) the library says NO_BLOCK
   * without touching the state and the copying path below serves the call. */
  int take_rc = SPEEXHIP_ERR_NO_BLOCK;
  void *block = NULL;
  uint32_t take_in = in_len, take_out = out_len;
  if (made_bytes >= kExternalMin && !g_no_external)
    take_rc = sample_bytes == 2
                  ? speexhip_resampler_process_interleaved_int_take(st, (const int16_t *)in_data, &take_in, &take_out, (int16_t **)&block)
                  : speexhip_resampler_process_interleaved_float_take(st, (const float *)in_data, &take_in, &take_out, (float **)&block);
  if (take_rc == SPEEXHIP_ERR_NO_BLOCK && made_bytes >= kExternalMin && !g_no_external)
    __atomic_fetch_add(&g_take_no_block, 1, __ATOMIC_RELAXED);
  if (take_rc != SPEEXHIP_ERR_NO_BLOCK) { /* (NO_BLOCK: the state is untouched -> the copying path below) */
    int rc = take_rc;
    uint32_t cap_len = take_out;
    UNLOCK(h);
    if (rc != 0 || cap_len != will_make || block == NULL) {
      speexhip_block_release(block);
      napi_throw_error(env, NULL, speexhip_resampler_strerror(rc != 0 ? rc : SPEEXHIP_ERR_BAD_STATE));
      return NULL;
    }
    napi_value ext;
    __atomic_fetch_add(&g_external_bytes, made_bytes, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_take_calls, 1, __ATOMIC_RELAXED);
    const napi_status ext_status = napi_create_external_buffer(env, made_bytes, block, finalize_block, (void *)made_bytes, &ext);
    if (ext_status != napi_ok) {
      /* The state has already advanced and the audio exists only in the block: a runtime that refuses external
       * Buffers (napi_no_external_buffers_allowed: Electron, V8 sandbox) or is out of memory for the wrapper must not
       * cost the caller its samples.  Copy them into an ordinary Buffer, hand the block back, and stop asking for
       * external Buffers for good when the runtime does not allow them. */
      __atomic_fetch_sub(&g_external_bytes, made_bytes, __ATOMIC_RELAXED);
      if ((int)ext_status == 22 /* napi_no_external_buffers_allowed (Node >= 18.?; not in this header) */) g_no_external = 1;
      napi_value copy;
      void *copied = NULL;
      const napi_status copy_status = napi_create_buffer_copy(env, made_bytes, block, &copied, &copy);
      speexhip_block_release(block);
      if (copy_status != napi_ok) {
        napi_throw_error(env, NULL, "speexhip N-API failure: napi_create_external_buffer and napi_create_buffer_copy");
        return NULL;
      }
      return copy;
    }
    int64_t adjusted = 0;
    (void)napi_adjust_external_memory(env, (int64_t)made_bytes, &adjusted); /* the collector sees what the Buffer holds */
    return ext;
  }
  napi_value out;
  void *dst = NULL;
  NAPI_OK_LOCKED(h, napi_create_buffer(env, (size_t)will_make * frame_bytes, &dst, &out));
  uint64_t nowhere = 0;
  if (dst == NULL) dst = &nowhere; /* empty Buffer: nothing will be written, but NULL means "no buffer" */
  /* out_len stays the caller's capacity (a smaller one could end the call's block loop early and
   * leave trailing input unconsumed); the library writes exactly will_make frames */
  int rc = sample_bytes == 2
               ? speexhip_resampler_process_interleaved_int(st, (const int16_t *)in_data, &in_len,
                                                            (int16_t *)dst, &out_len)
               : speexhip_resampler_process_interleaved_float(st, (const float *)in_data, &in_len,
                                                              (float *)dst, &out_len);
  UNLOCK(h);
  if (rc != 0 || out_len != will_make) {
    napi_throw_error(env, NULL, speexhip_resampler_strerror(rc != 0 ? rc : SPEEXHIP_ERR_BAD_STATE));
    return NULL;
  }
  return out;
}
static napi_value Process(napi_env env, napi_callback_info info) { return process_common(env, info, 2); }
static napi_value ProcessFloat(napi_env env, napi_callback_info info) { return process_common(env, info, 4); }

/* processChunks(handle, chunks: Buffer[], inFrames: number[], outCapacities: number[]) -> Buffer[]
 * n consecutive process() calls as one transfer + one launch
 * (speexhip_resampler_process_chunks_int); the i-th Buffer is what the i-th call returns. */
static napi_value ProcessChunks(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  uint32_t n = 0;
  NAPI_OK(napi_get_array_length(env, argv[1], &n));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si);
  const size_t frame_bytes = (size_t)si.nb_channels * 2;
  const int16_t **ptrs = (const int16_t **)calloc(n ? n : 1, sizeof(*ptrs));
  uint32_t *in_len = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
  uint32_t *out_len = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
  int16_t *tmp = NULL;
  napi_value result = NULL;
  const char *fail = NULL;
  size_t total_cap = 0;
  if (ptrs == NULL || in_len == NULL || out_len == NULL) fail = speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED);
  for (uint32_t i = 0; fail == NULL && i < n; i++) {
    napi_value c, a, b;
    void *data = NULL;
    size_t bytes = 0;
    if (napi_get_element(env, argv[1], i, &c) != napi_ok || napi_get_element(env, argv[2], i, &a) != napi_ok ||
        napi_get_element(env, argv[3], i, &b) != napi_ok || !buffer_or_null(env, c, &data, &bytes) ||
        napi_get_value_uint32(env, a, &in_len[i]) != napi_ok ||
        napi_get_value_uint32(env, b, &out_len[i]) != napi_ok) {
      fail = "processChunks expects (handle, Buffer[], number[], number[])";
    } else if (data != NULL && (size_t)in_len[i] * frame_bytes > bytes) {
      fail = "input frame count exceeds the chunk";
    }
    ptrs[i] = (const int16_t *)data;
    total_cap += out_len[i];
  }
  if (fail == NULL) {
    tmp = (int16_t *)malloc(total_cap * frame_bytes + 2);
    if (tmp == NULL) fail = speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED);
  }
  if (fail == NULL) {
    int rc = speexhip_resampler_process_chunks_int(st, n, ptrs, in_len, tmp, out_len);
    if (rc != 0) fail = speexhip_resampler_strerror(rc);
  }
  UNLOCK(h);
  if (fail == NULL && napi_create_array_with_length(env, n, &result) == napi_ok) {
    size_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
      napi_value buf;
      void *copied = NULL;
      if (napi_create_buffer_copy(env, (size_t)out_len[i] * frame_bytes, (char *)tmp + off, &copied, &buf) != napi_ok ||
          napi_set_element(env, result, i, buf) != napi_ok) {
        fail = "speexhip N-API failure: building the result array";
        break;
      }
      off += (size_t)out_len[i] * frame_bytes;
    }
  }
  free(tmp);
  free(ptrs);
  free(in_len);
  free(out_len);
  if (fail != NULL) {
    napi_throw_error(env, NULL, fail);
    return NULL;
  }
  return result;
}

/* processChunksAsync(handle, chunks[], inFrames[], outCapacities[]) -> Promise<Buffer[]>: processChunks on a libuv pool
 * thread (the Transform's `pipeline` option: chunks that arrive while a call is in flight leave together as the next) */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref handle_ref;
  napi_ref *chunk_refs;
  Handle *h;
  uint32_t n, channels;
  const int16_t **ptrs;
  uint32_t *in_len, *out_len;
  int16_t *tmp;
  int rc;
  char errmsg[256];
} ChunksJob;

static void chunks_free(napi_env env, ChunksJob *j) {
  if (j == NULL) return;
  if (j->chunk_refs != NULL)
    for (uint32_t i = 0; i < j->n; i++)
      if (j->chunk_refs[i] != NULL) napi_delete_reference(env, j->chunk_refs[i]);
  if (j->handle_ref != NULL) napi_delete_reference(env, j->handle_ref);
  free(j->chunk_refs);
  free(j->ptrs);
  free(j->in_len);
  free(j->out_len);
  free(j->tmp);
  free(j);
}
static void chunks_execute(napi_env env, void *data) {
  (void)env;
  ChunksJob *j = (ChunksJob *)data;
  pthread_mutex_lock(&j->h->lock);
  j->rc = j->h->st == NULL ? SPEEXHIP_ERR_BAD_STATE
                           : speexhip_resampler_process_chunks_int(j->h->st, j->n, j->ptrs, j->in_len, j->tmp, j->out_len);
  if (j->rc != 0) snprintf(j->errmsg, sizeof(j->errmsg), "%s", speexhip_resampler_strerror(j->rc));
  pthread_mutex_unlock(&j->h->lock);
}
static void chunks_complete(napi_env env, napi_status status, void *data) {
  ChunksJob *j = (ChunksJob *)data;
  napi_value v, result;
  int ok = status == napi_ok && j->rc == 0 && napi_create_array_with_length(env, j->n, &result) == napi_ok;
  if (ok) {
    size_t off = 0;
    const size_t frame_bytes = (size_t)j->channels * 2;
    for (uint32_t i = 0; ok && i < j->n; i++) {
      napi_value buf;
      void *copied = NULL;
      ok = napi_create_buffer_copy(env, (size_t)j->out_len[i] * frame_bytes, (char *)j->tmp + off, &copied, &buf) == napi_ok &&
           napi_set_element(env, result, i, buf) == napi_ok;
      off += (size_t)j->out_len[i] * frame_bytes;
    }
  }
  if (ok) {
    napi_resolve_deferred(env, j->deferred, result);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, j->rc != 0 ? j->errmsg : "speexhip: asynchronous call failed", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  napi_delete_async_work(env, j->work);
  chunks_free(env, j);
}
static napi_value ProcessChunksAsync(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  uint32_t n = 0;
  NAPI_OK(napi_get_array_length(env, argv[1], &n));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si); /* only the channel count is used: it never changes */
  UNLOCK(h);
  const size_t frame_bytes = (size_t)si.nb_channels * 2, m = n ? n : 1;
  ChunksJob *j = (ChunksJob *)calloc(1, sizeof(ChunksJob));
  const char *fail = NULL;
  if (j != NULL) {
    j->n = n;
    j->h = h;
    j->channels = si.nb_channels;
    j->chunk_refs = (napi_ref *)calloc(m, sizeof(napi_ref));
    j->ptrs = (const int16_t **)calloc(m, sizeof(*j->ptrs));
    j->in_len = (uint32_t *)calloc(m, sizeof(uint32_t));
    j->out_len = (uint32_t *)calloc(m, sizeof(uint32_t));
  }
  if (j == NULL || !j->chunk_refs || !j->ptrs || !j->in_len || !j->out_len) fail = speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED);
  size_t total_cap = 0;
  for (uint32_t i = 0; fail == NULL && i < n; i++) {
    napi_value c, a, b;
    void *data = NULL;
    size_t bytes = 0;
    if (napi_get_element(env, argv[1], i, &c) != napi_ok || napi_get_element(env, argv[2], i, &a) != napi_ok ||
        napi_get_element(env, argv[3], i, &b) != napi_ok || !buffer_or_null(env, c, &data, &bytes) ||
        napi_get_value_uint32(env, a, &j->in_len[i]) != napi_ok || napi_get_value_uint32(env, b, &j->out_len[i]) != napi_ok) {
      fail = "processChunksAsync expects (handle, Buffer[], number[], number[])";
    } else if (data != NULL && (size_t)j->in_len[i] * frame_bytes > bytes) {
      fail = "input frame count exceeds the chunk";
    } else if (data != NULL && napi_create_reference(env, c, 1, &j->chunk_refs[i]) != napi_ok) { /* keep the bytes alive */
      fail = "speexhip N-API failure: napi_create_reference";
    }
    j->ptrs[i] = (const int16_t *)data;
    total_cap += j->out_len[i];
  }
  napi_value promise = NULL, name;
  if (fail == NULL && (j->tmp = (int16_t *)malloc(total_cap * frame_bytes + 2)) == NULL) fail = speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED);
  if (fail == NULL &&
      (napi_create_reference(env, argv[0], 1, &j->handle_ref) != napi_ok || napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
       napi_create_string_utf8(env, "speexhip.processChunksAsync", NAPI_AUTO_LENGTH, &name) != napi_ok ||
       napi_create_async_work(env, NULL, name, chunks_execute, chunks_complete, j, &j->work) != napi_ok ||
       napi_queue_async_work(env, j->work) != napi_ok))
    fail = "speexhip N-API failure: queueing processChunksAsync";
  if (fail != NULL) {
    chunks_free(env, j);
    napi_throw_error(env, NULL, fail);
    return NULL;
  }
  return promise;
}

/* processAsync(handle, chunk, inFrames, outCapacityFrames) -> Promise<Buffer>: the same call on
 * a libuv pool thread, so H2D + kernels + D2H do not block the event loop.  The caller
 * (index.js) chains the promises of one instance, so calls on one state stay in order. */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref chunk_ref, handle_ref;
  Handle *h;
  const int16_t *in;
  int16_t *out;
  uint32_t in_len, out_len, channels;
  int rc;
  char errmsg[256];
} AsyncJob;

static void async_execute(napi_env env, void *data) {
  (void)env;
  AsyncJob *j = (AsyncJob *)data;
  pthread_mutex_lock(&j->h->lock);
  if (j->h->st == NULL) {
    j->rc = SPEEXHIP_ERR_BAD_STATE;
  } else {
    j->rc = speexhip_resampler_process_interleaved_int(j->h->st, j->in, &j->in_len, j->out, &j->out_len);
  }
  /* DEVICE errors keep their text per thread: fetch it on the thread that failed */
  if (j->rc != 0) snprintf(j->errmsg, sizeof(j->errmsg), "%s", speexhip_resampler_strerror(j->rc));
  pthread_mutex_unlock(&j->h->lock);
}

static void async_complete(napi_env env, napi_status status, void *data) {
  AsyncJob *j = (AsyncJob *)data;
  napi_value v;
  if (status == napi_ok && j->rc == 0) {
    void *copied = NULL;
    if (napi_create_buffer_copy(env, (size_t)j->out_len * j->channels * 2, j->out, &copied, &v) == napi_ok)
      napi_resolve_deferred(env, j->deferred, v);
    else
      status = napi_generic_failure;
  }
  if (status != napi_ok || j->rc != 0) {
    napi_value msg;
    napi_create_string_utf8(env, j->rc != 0 ? j->errmsg : "speexhip: asynchronous call failed", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  if (j->chunk_ref) napi_delete_reference(env, j->chunk_ref);
  napi_delete_reference(env, j->handle_ref);
  napi_delete_async_work(env, j->work);
  free(j->out);
  free(j);
}

static napi_value ProcessAsync(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void *in_data = NULL;
  size_t in_bytes = 0;
  if (!buffer_or_null(env, argv[1], &in_data, &in_bytes)) {
    napi_throw_type_error(env, NULL, "chunk must be a Buffer or null");
    return NULL;
  }
  uint32_t in_len = 0, out_len = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[2], &in_len));
  NAPI_OK(napi_get_value_uint32(env, argv[3], &out_len));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si); /* only the channel count is used: it never changes */
  UNLOCK(h);
  if (in_data != NULL && (size_t)in_len * si.nb_channels * 2 > in_bytes) {
    napi_throw_range_error(env, NULL, "input frame count exceeds the chunk");
    return NULL;
  }
  AsyncJob *j = (AsyncJob *)calloc(1, sizeof(AsyncJob));
  if (j != NULL) j->out = (int16_t *)malloc((size_t)out_len * si.nb_channels * 2 + 2);
  if (j == NULL || j->out == NULL) {
    free(j);
    napi_throw_error(env, NULL, speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED));
    return NULL;
  }
  j->h = h;
  j->in = (const int16_t *)in_data;
  j->in_len = in_len;
  j->out_len = out_len;
  j->channels = si.nb_channels;
  napi_value promise, name;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  if (in_data != NULL) NAPI_OK(napi_create_reference(env, argv[1], 1, &j->chunk_ref)); /* keep the bytes alive */
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->handle_ref));
  NAPI_OK(napi_create_string_utf8(env, "speexhip.processAsync", NAPI_AUTO_LENGTH, &name));
  NAPI_OK(napi_create_async_work(env, NULL, name, async_execute, async_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

/* ---- mid-stream control: each throws Error(strerror(code)) on failure ---- */
static napi_value control_result(napi_env env, int rc) {
  if (rc != 0) napi_throw_error(env, NULL, speexhip_resampler_strerror(rc));
  return NULL;
}

static napi_value SetRate(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  uint32_t v[4] = {0, 0, 0, 0};
  for (size_t i = 1; i < argc && i < 5; i++) NAPI_OK_LOCKED(h, napi_get_value_uint32(env, argv[i], &v[i - 1]));
  /* (handle, inRate, outRate) or (handle, ratioNum, ratioDen, inRate, outRate) */
  const int rc = argc >= 5 ? speexhip_resampler_set_rate_frac(st, v[0], v[1], v[2], v[3])
                           : speexhip_resampler_set_rate(st, v[0], v[1]);
  UNLOCK(h);
  return control_result(env, rc);
}

static napi_value SetQuality(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  int32_t q = 0;
  NAPI_OK_LOCKED(h, napi_get_value_int32(env, argv[1], &q));
  const int rc = speexhip_resampler_set_quality(st, q);
  UNLOCK(h);
  return control_result(env, rc);
}

static napi_value SkipZeros(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  const int rc = speexhip_resampler_skip_zeros(st);
  UNLOCK(h);
  return control_result(env, rc);
}

static napi_value ResetMem(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  const int rc = speexhip_resampler_reset_mem(st);
  UNLOCK(h);
  return control_result(env, rc);
}

static napi_value pair_u32(napi_env env, uint32_t a, uint32_t b) {
  napi_value arr, v;
  NAPI_OK(napi_create_array_with_length(env, 2, &arr));
  NAPI_OK(napi_create_uint32(env, a, &v));
  NAPI_OK(napi_set_element(env, arr, 0, v));
  NAPI_OK(napi_create_uint32(env, b, &v));
  NAPI_OK(napi_set_element(env, arr, 1, v));
  return arr;
}

/* getLatency(handle) -> [inputLatencyFrames, outputLatencyFrames] */
static napi_value GetLatency(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  const uint32_t in_lat = (uint32_t)speexhip_resampler_get_input_latency(st);
  const uint32_t out_lat = (uint32_t)speexhip_resampler_get_output_latency(st);
  UNLOCK(h);
  return pair_u32(env, in_lat, out_lat);
}

static napi_value SetMode(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  int32_t mode = 0;
  NAPI_OK_LOCKED(h, napi_get_value_int32(env, argv[1], &mode));
  const int rc = speexhip_resampler_set_mode(st, mode);
  UNLOCK(h);
  return control_result(env, rc);
}

static napi_value GetInfo(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  SpeexHipInfo si;
  speexhip_resampler_get_info(st, &si);
  UNLOCK(h);
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
  PUT_U32(magic_samples) PUT_U32(block_in) PUT_I32(accumulate_bits)
  return obj;
}

/* getRate(handle) -> [inRate, outRate]  (speex_resampler_get_rate; exported but unused by
 * src/index.ts, kept for surface parity) */
static napi_value GetRate(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  Handle *h = NULL;
  SpeexHipResamplerState *st = lock_state(env, argv[0], &h);
  if (st == NULL) return NULL;
  uint32_t a = 0, b = 0;
  speexhip_resampler_get_rate(st, &a, &b);
  UNLOCK(h);
  return pair_u32(env, a, b);
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

/* releaseCachedMemory() -> bytes handed back to the driver (speexhip_release_cached_memory) */
static napi_value ReleaseCachedMemory(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  NAPI_OK(napi_create_double(env, (double)speexhip_release_cached_memory(), &v));
  return v;
}

/* ---- many states, one call (speexhip_resampler_process_many_int): SpeexResamplerBatch.processChunks and the
 * coalescer of processChunkAsync calls issued in one tick (index.js) ---- */
typedef struct {
  uint32_t n;
  Handle **h;               /* per entry */
  Handle **locked;          /* the distinct handles, sorted by address (lock order) */
  uint32_t n_locked;
  SpeexHipResamplerState **st;
  const int16_t **in;
  int16_t **out;
  uint32_t *in_len, *out_len, *will_make, *channels;
  int *codes;
  /* asynchronous form */
  napi_async_work work;
  napi_deferred deferred;
  napi_ref *refs;           /* handles, chunks, result Buffers: 3 per entry (chunk ref may be NULL) */
  napi_ref result_ref;      /* the result array */
  int rc;
  int ran;                  /* the library call was made: codes[] says which entries succeeded */
  char errmsg[256];
} ManyJob;

static void many_free(napi_env env, ManyJob *j) {
  if (j == NULL) return;
  if (j->refs != NULL) {
    for (uint32_t i = 0; i < 3 * j->n; i++)
      if (j->refs[i] != NULL) napi_delete_reference(env, j->refs[i]);
  }
  if (j->result_ref != NULL) napi_delete_reference(env, j->result_ref);
  free(j->h);
  free(j->locked);
  free(j->st);
  free(j->in);
  free(j->out);
  free(j->in_len);
  free(j->out_len);
  free(j->will_make);
  free(j->channels);
  free(j->codes);
  free(j->refs);
  free(j);
}

static int cmp_handle(const void *a, const void *b) {
  const uintptr_t x = (uintptr_t)*(Handle *const *)a, y = (uintptr_t)*(Handle *const *)b;
  return x < y ? -1 : x > y;
}
static void many_lock(ManyJob *j) {
  for (uint32_t k = 0; k < j->n_locked; k++) pthread_mutex_lock(&j->locked[k]->lock);
}
static void many_unlock(ManyJob *j) {
  for (uint32_t k = j->n_locked; k-- > 0;) pthread_mutex_unlock(&j->locked[k]->lock);
}
/* the call itself, locks held; 0 or a code with its text in errmsg */
static void many_run(ManyJob *j) {
  for (uint32_t i = 0; i < j->n; i++) {
    j->st[i] = j->h[i]->st;
    if (j->st[i] == NULL) {
      j->rc = SPEEXHIP_ERR_BAD_STATE;
      snprintf(j->errmsg, sizeof(j->errmsg), "%s", speexhip_resampler_strerror(j->rc));
      return;
    }
  }
  /* The result Buffers were sized from the states' counters when the call was prepared (will_make).  The asynchronous
   * form released the locks in between: if a state has moved since (another call of that instance slipped in), the
   * library would now write a different number of frames -- possibly more than the Buffer holds.  Ask again, under the
   * locks, BEFORE anything is written, and refuse the step whole. */
  for (uint32_t i = 0; i < j->n; i++) {
    uint32_t will_use = 0, will_make = 0;
    speexhip_resampler_peek(j->st[i], j->in_len[i], j->out_len[i], 0, &will_use, &will_make);
    if (will_make != j->will_make[i]) {
      j->rc = SPEEXHIP_ERR_BAD_STATE;
      snprintf(j->errmsg, sizeof(j->errmsg), "%s (a state of this step was used by another call while the step was pending)",
               speexhip_resampler_strerror(j->rc));
      return;
    }
  }
  j->rc = speexhip_resampler_process_many_int(j->n, j->st, j->in, j->in_len, j->out, j->out_len, j->codes);
  j->ran = 1;
  for (uint32_t i = 0; i < j->n; i++)
    if (j->codes[i] == 0 && j->out_len[i] != j->will_make[i]) {
      j->codes[i] = SPEEXHIP_ERR_BAD_STATE;
      if (j->rc == 0) j->rc = SPEEXHIP_ERR_BAD_STATE;
    }
  if (j->rc != 0) snprintf(j->errmsg, sizeof(j->errmsg), "%s", speexhip_resampler_strerror(j->rc));
}

/* Parses (handles[], chunks[], inFrames[], caps[]), sizes the results from the states' counters and creates the
 * result Buffers; returns the job with *result = the array of Buffers, or NULL with an exception pending. */
static ManyJob *many_prepare(napi_env env, napi_callback_info info, napi_value *result, int keep_refs) {
  size_t argc = 4;
  napi_value argv[4];
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < 4) {
    napi_throw_type_error(env, NULL, "processMany expects (handle[], Buffer[], number[], number[])");
    return NULL;
  }
  uint32_t n = 0;
  if (napi_get_array_length(env, argv[0], &n) != napi_ok) {
    napi_throw_type_error(env, NULL, "processMany expects (handle[], Buffer[], number[], number[])");
    return NULL;
  }
  ManyJob *j = (ManyJob *)calloc(1, sizeof(ManyJob));
  const size_t m = n ? n : 1;
  if (j != NULL) {
    j->n = n;
    j->h = (Handle **)calloc(m, sizeof(Handle *));
    j->locked = (Handle **)calloc(m, sizeof(Handle *));
    j->st = (SpeexHipResamplerState **)calloc(m, sizeof(*j->st));
    j->in = (const int16_t **)calloc(m, sizeof(*j->in));
    j->out = (int16_t **)calloc(m, sizeof(*j->out));
    j->in_len = (uint32_t *)calloc(m, sizeof(uint32_t));
    j->out_len = (uint32_t *)calloc(m, sizeof(uint32_t));
    j->will_make = (uint32_t *)calloc(m, sizeof(uint32_t));
    j->channels = (uint32_t *)calloc(m, sizeof(uint32_t));
    j->codes = (int *)calloc(m, sizeof(int));
    if (keep_refs) j->refs = (napi_ref *)calloc(3 * m, sizeof(napi_ref));
  }
  if (j == NULL || !j->h || !j->locked || !j->st || !j->in || !j->out || !j->in_len || !j->out_len || !j->will_make ||
      !j->channels || !j->codes || (keep_refs && !j->refs)) {
    many_free(env, j);
    napi_throw_error(env, NULL, speexhip_resampler_strerror(SPEEXHIP_ERR_ALLOC_FAILED));
    return NULL;
  }
  const char *fail = NULL;
  int type_error = 0;
  for (uint32_t i = 0; fail == NULL && i < n; i++) {
    napi_value hv, c, a, b;
    void *p = NULL, *data = NULL;
    size_t bytes = 0;
    if (napi_get_element(env, argv[0], i, &hv) != napi_ok || napi_get_value_external(env, hv, &p) != napi_ok || p == NULL ||
        napi_get_element(env, argv[1], i, &c) != napi_ok || !buffer_or_null(env, c, &data, &bytes) ||
        napi_get_element(env, argv[2], i, &a) != napi_ok || napi_get_value_uint32(env, a, &j->in_len[i]) != napi_ok ||
        napi_get_element(env, argv[3], i, &b) != napi_ok || napi_get_value_uint32(env, b, &j->out_len[i]) != napi_ok) {
      fail = "processMany expects (handle[], Buffer[], number[], number[])";
      type_error = 1;
      break;
    }
    j->h[i] = (Handle *)p;
    j->in[i] = (const int16_t *)data;
    /* (the chunk's byte count is checked against its frames below, once the channel count is known) */
    j->channels[i] = (uint32_t)bytes;  /* borrowed until then */
    if (keep_refs) {
      if (napi_create_reference(env, hv, 1, &j->refs[3 * i]) != napi_ok ||
          (data != NULL && napi_create_reference(env, c, 1, &j->refs[3 * i + 1]) != napi_ok))
        fail = "speexhip N-API failure: napi_create_reference";
    }
  }
  if (fail == NULL) {
    memcpy(j->locked, j->h, n * sizeof(Handle *));
    qsort(j->locked, n, sizeof(Handle *), cmp_handle);
    for (uint32_t k = 0; k < n; k++) {
      if (k > 0 && j->locked[k] == j->locked[k - 1]) {
        fail = "processMany: a state may appear once per call";
        break;
      }
    }
    j->n_locked = n;
  }
  if (fail == NULL && napi_create_array_with_length(env, n, result) != napi_ok) fail = "speexhip N-API failure: napi_create_array";
  if (fail == NULL) {
    many_lock(j);
    for (uint32_t i = 0; fail == NULL && i < n; i++) {
      SpeexHipResamplerState *st = j->h[i]->st;
      if (st == NULL) {
        fail = speexhip_resampler_strerror(SPEEXHIP_ERR_BAD_STATE);
        break;
      }
      SpeexHipInfo si;
      speexhip_resampler_get_info(st, &si);
      const size_t bytes = j->channels[i], frame_bytes = (size_t)si.nb_channels * 2;
      j->channels[i] = si.nb_channels;
      if (j->in[i] != NULL && (size_t)j->in_len[i] * frame_bytes > bytes) {
        fail = "input frame count exceeds the chunk";
        break;
      }
      uint32_t will_use = 0;
      speexhip_resampler_peek(st, j->in_len[i], j->out_len[i], 0, &will_use, &j->will_make[i]);
      napi_value buf;
      void *dst = NULL;
      /* Round 6: like process(), results of a few KB and more are pinned blocks of the library handed to JavaScript
       * as external Buffers -- the library recognises its own blocks among the out[] pointers and lets the kernel
       * write them in place (no copy out of a staging buffer).  No block free, or a runtime without external
       * Buffers: an ordinary Buffer, as before. */
      const size_t made_bytes = (size_t)j->will_make[i] * frame_bytes;
      int have = 0;
      if (made_bytes >= kExternalMin && !g_no_external) {
        void *block = speexhip_block_acquire(made_bytes);
        if (block != NULL) {
          if (napi_create_external_buffer(env, made_bytes, block, finalize_block, (void *)made_bytes, &buf) == napi_ok) {
            int64_t adjusted = 0;
            __atomic_fetch_add(&g_external_bytes, made_bytes, __ATOMIC_RELAXED);
            __atomic_fetch_add(&g_take_calls, 1, __ATOMIC_RELAXED);
            (void)napi_adjust_external_memory(env, (int64_t)made_bytes, &adjusted);
            dst = block;
            have = 1;
          } else {
            speexhip_block_release(block);
          }
        } else {
          __atomic_fetch_add(&g_take_no_block, 1, __ATOMIC_RELAXED);
        }
      }
      if ((!have && napi_create_buffer(env, made_bytes, &dst, &buf) != napi_ok) ||
          napi_set_element(env, *result, i, buf) != napi_ok ||
          (keep_refs && napi_create_reference(env, buf, 1, &j->refs[3 * i + 2]) != napi_ok)) {
        fail = "speexhip N-API failure: building the result array";
        break;
      }
      static uint64_t nowhere = 0;
      j->out[i] = (int16_t *)(dst != NULL ? dst : (void *)&nowhere);
    }
    if (fail != NULL || keep_refs) many_unlock(j); /* (the synchronous form keeps the locks until the call is done) */
  }
  if (fail != NULL) {
    many_free(env, j);
    if (type_error)
      napi_throw_type_error(env, NULL, fail);
    else
      napi_throw_error(env, NULL, fail);
    return NULL;
  }
  return j;
}

/* processMany(handles[], chunks[], inFrames[], outCapacities[]) -> Buffer[]: entry i is what
 * process(handles[i], chunks[i], inFrames[i], outCapacities[i]) returns. */
static napi_value ProcessMany(napi_env env, napi_callback_info info) {
  napi_value result = NULL;
  ManyJob *j = many_prepare(env, info, &result, 0);
  if (j == NULL) return NULL;
  many_run(j); /* (locks held since many_prepare) */
  many_unlock(j);
  const int rc = j->rc;
  char msg[256];
  snprintf(msg, sizeof(msg), "%s", j->errmsg);
  many_free(env, j);
  if (rc != 0) {
    napi_throw_error(env, NULL, msg);
    return NULL;
  }
  return result;
}

static void many_execute(napi_env env, void *data) {
  (void)env;
  ManyJob *j = (ManyJob *)data;
  many_lock(j);
  many_run(j);
  many_unlock(j);
}
static void many_complete(napi_env env, napi_status status, void *data) {
  ManyJob *j = (ManyJob *)data;
  napi_value v;
  if (status == napi_ok && j->rc == 0 && napi_get_reference_value(env, j->result_ref, &v) == napi_ok) {
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, j->rc != 0 ? j->errmsg : "speexhip: asynchronous call failed", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    /* When the library call was made, every entry has an outcome of its own (codes[i]; the states of the entries
     * that succeeded HAVE advanced and their audio is in results[i]): the error carries both, so that a caller who
     * coalesced independent instances can settle each one by itself (index.js, flushTick) -- in the reference only the
     * offending instance fails. */
    napi_value codes, results, c;
    if (status == napi_ok && j->ran && napi_create_array_with_length(env, j->n, &codes) == napi_ok &&
        napi_get_reference_value(env, j->result_ref, &results) == napi_ok) {
      for (uint32_t i = 0; i < j->n; i++)
        if (napi_create_int32(env, j->codes[i], &c) == napi_ok) napi_set_element(env, codes, i, c);
      napi_set_named_property(env, v, "codes", codes);
      napi_set_named_property(env, v, "results", results);
    }
    napi_reject_deferred(env, j->deferred, v);
  }
  napi_delete_async_work(env, j->work);
  many_free(env, j);
}
/* processManyAsync(...) -> Promise<Buffer[]>: the same call on a libuv pool thread.  The results are sized on the
 * calling thread from the states' counters, so no state of the call may be used until the promise settles (index.js
 * keeps one call per instance in flight). */
static napi_value ProcessManyAsync(napi_env env, napi_callback_info info) {
  napi_value result = NULL, promise, name;
  ManyJob *j = many_prepare(env, info, &result, 1);
  if (j == NULL) return NULL;
  if (napi_create_reference(env, result, 1, &j->result_ref) != napi_ok ||
      napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
      napi_create_string_utf8(env, "speexhip.processManyAsync", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, NULL, name, many_execute, many_complete, j, &j->work) != napi_ok ||
      napi_queue_async_work(env, j->work) != napi_ok) {
    many_free(env, j);
    napi_throw_error(env, NULL, "speexhip N-API failure: queueing processManyAsync");
    return NULL;
  }
  return promise;
}

/* warmup() -> Promise<number>: speexhip_warmup(-1) on a libuv pool thread; resolves with its return code (never rejects:
 * a box without a GPU is reported by the first real call, like the reference's errors) */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  int rc;
} WarmJob;
static void warm_execute(napi_env env, void *data) {
  (void)env;
  ((WarmJob *)data)->rc = speexhip_warmup(-1);
}
static void warm_complete(napi_env env, napi_status status, void *data) {
  WarmJob *j = (WarmJob *)data;
  napi_value v;
  (void)status;
  napi_create_int32(env, j->rc, &v);
  napi_resolve_deferred(env, j->deferred, v);
  napi_delete_async_work(env, j->work);
  free(j);
}
static napi_value Warmup(napi_env env, napi_callback_info info) {
  (void)info;
  WarmJob *j = (WarmJob *)calloc(1, sizeof(WarmJob));
  napi_value promise, name;
  if (j == NULL || napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
      napi_create_string_utf8(env, "speexhip.warmup", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, NULL, name, warm_execute, warm_complete, j, &j->work) != napi_ok ||
      napi_queue_async_work(env, j->work) != napi_ok) {
    free(j);
    napi_throw_error(env, NULL, "speexhip N-API failure: queueing warmup");
    return NULL;
  }
  return promise;
}

/* allocChunk(bytes) -> Buffer over a pinned block of the library (speexhip_block_acquire) for the caller to FILL -- read
 * a file or a socket into it, let a decoder write it -- and pass to process() / processMany(): the library recognises its
 * own blocks and lets the kernel read the chunk in place through PCIe, where an ordinary Buffer is first copied into
 * staging memory (the reference copies every chunk into the WASM heap, src/index.ts:71-92).  The block goes back to the
 * pool when the Buffer is collected.  No block free, a runtime without external Buffers, or SPEEXHIP_NAPI_COPY=1: an
 * ordinary Buffer of that size -- same results, the usual staging. */
static napi_value AllocChunk(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], buf;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  double want = 0;
  if (argc < 1 || napi_get_value_double(env, argv[0], &want) != napi_ok || !(want >= 0) || want > 4294967295.0) {
    napi_throw_range_error(env, NULL, "allocChunk expects a size in bytes");
    return NULL;
  }
  const size_t bytes = (size_t)want;
  if (bytes >= kExternalMin && !g_no_external) {
    void *block = speexhip_block_acquire(bytes);
    if (block != NULL) {
      if (napi_create_external_buffer(env, bytes, block, finalize_block, (void *)bytes, &buf) == napi_ok) {
        int64_t adjusted = 0;
        __atomic_fetch_add(&g_external_bytes, bytes, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_pinned_chunks, 1, __ATOMIC_RELAXED);
        (void)napi_adjust_external_memory(env, (int64_t)bytes, &adjusted);
        return buf;
      }
      speexhip_block_release(block);
    }
  }
  void *data = NULL;
  NAPI_OK(napi_create_buffer(env, bytes, &data, &buf));
  return buf;
}

/* deviceCount() -> GPUs the library can place states on (speexhip_device_count) */
static napi_value DeviceCount(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  NAPI_OK(napi_create_int32(env, speexhip_device_count(), &v));
  return v;
}

/* stats() -> { externalBytes, takeCalls, takeNoBlock }: bytes of pinned result blocks JavaScript holds as external
 * Buffers, results delivered that way, and calls that found the slab full and copied instead (diagnostics) */
static napi_value Stats(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value obj, v;
  NAPI_OK(napi_create_object(env, &obj));
  NAPI_OK(napi_create_double(env, (double)__atomic_load_n(&g_external_bytes, __ATOMIC_RELAXED), &v));
  NAPI_OK(napi_set_named_property(env, obj, "externalBytes", v));
  NAPI_OK(napi_create_double(env, (double)__atomic_load_n(&g_take_calls, __ATOMIC_RELAXED), &v));
  NAPI_OK(napi_set_named_property(env, obj, "takeCalls", v));
  NAPI_OK(napi_create_double(env, (double)__atomic_load_n(&g_take_no_block, __ATOMIC_RELAXED), &v));
  NAPI_OK(napi_set_named_property(env, obj, "takeNoBlock", v));
  NAPI_OK(napi_create_double(env, (double)__atomic_load_n(&g_pinned_chunks, __ATOMIC_RELAXED), &v));
  NAPI_OK(napi_set_named_property(env, obj, "pinnedChunks", v));
  return obj;
}

/* NAPI_MODULE_INIT: the well-known symbol napi_register_module_v1, so that every environment of the process -- the main
 * thread and each worker_threads Worker -- can load the addon (a module that registers itself from a static constructor
 * loads once per process).  The addon keeps no per-environment state: handles are externals, the counters above are
 * atomics, the library underneath is thread-safe per state. */
NAPI_MODULE_INIT() {
  const char *copy_env = getenv("SPEEXHIP_NAPI_COPY");
  g_no_external = copy_env != NULL && copy_env[0] != '\0' && copy_env[0] != '0';
  napi_property_descriptor props[] = {
      {"init", NULL, Init, NULL, NULL, NULL, napi_default, NULL},
      {"destroy", NULL, Destroy, NULL, NULL, NULL, napi_default, NULL},
      {"process", NULL, Process, NULL, NULL, NULL, napi_default, NULL},
      {"processFloat", NULL, ProcessFloat, NULL, NULL, NULL, napi_default, NULL},
      {"processChunks", NULL, ProcessChunks, NULL, NULL, NULL, napi_default, NULL},
      {"processAsync", NULL, ProcessAsync, NULL, NULL, NULL, napi_default, NULL},
      {"processChunksAsync", NULL, ProcessChunksAsync, NULL, NULL, NULL, napi_default, NULL},
      {"setRate", NULL, SetRate, NULL, NULL, NULL, napi_default, NULL},
      {"setQuality", NULL, SetQuality, NULL, NULL, NULL, napi_default, NULL},
      {"skipZeros", NULL, SkipZeros, NULL, NULL, NULL, napi_default, NULL},
      {"resetMem", NULL, ResetMem, NULL, NULL, NULL, napi_default, NULL},
      {"getLatency", NULL, GetLatency, NULL, NULL, NULL, napi_default, NULL},
      {"setMode", NULL, SetMode, NULL, NULL, NULL, napi_default, NULL},
      {"getInfo", NULL, GetInfo, NULL, NULL, NULL, napi_default, NULL},
      {"getRate", NULL, GetRate, NULL, NULL, NULL, napi_default, NULL},
      {"strerror", NULL, StrError, NULL, NULL, NULL, napi_default, NULL},
      {"version", NULL, Version, NULL, NULL, NULL, napi_default, NULL},
      {"releaseCachedMemory", NULL, ReleaseCachedMemory, NULL, NULL, NULL, napi_default, NULL},
      {"processMany", NULL, ProcessMany, NULL, NULL, NULL, napi_default, NULL},
      {"processManyAsync", NULL, ProcessManyAsync, NULL, NULL, NULL, napi_default, NULL},
      {"deviceCount", NULL, DeviceCount, NULL, NULL, NULL, napi_default, NULL},
      {"warmup", NULL, Warmup, NULL, NULL, NULL, napi_default, NULL},
      {"stats", NULL, Stats, NULL, NULL, NULL, napi_default, NULL},
      {"allocChunk", NULL, AllocChunk, NULL, NULL, NULL, napi_default, NULL},
  };
  if (napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props) != napi_ok)
    napi_throw_error(env, NULL, "speexhip: cannot define exports");
  return exports;
}

