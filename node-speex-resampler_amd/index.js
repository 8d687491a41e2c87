'use strict';
// index.js -- drop-in replacement for the reference's app/index.js (compiled src/index.ts):
// same exports (default SpeexResampler, named SpeexResamplerTransform), same constructor
// arguments, same thrown messages, same lazy init, same grow-only output-capacity rule --
// but the work is done on an MI355X through the N-API addon speex_hip_napi.node ->
// libspeexhip.so (HIP kernels) instead of the Emscripten module src/speex_wasm.js.
// Hand-written CommonJS (no tsc in the build image); types live in index.d.ts.
Object.defineProperty(exports, '__esModule', { value: true });
const { Transform } = require('stream');
const path = require('path');

// Counterpart of `SpeexWasm()` (reference src/index.ts:18-19): a promise for the native
// module; `speexModule` is only set once it resolves, so calling processChunk before
// awaiting initPromise throws exactly like the reference does.
let speexModule;
const globalModulePromise = new Promise((resolve, reject) => {
  try {
    resolve(require(path.join(__dirname, 'speex_hip_napi.node')));
  } catch (e) {
    reject(e);
  }
}).then((m) => {
  // The reference compiles its WASM module here, behind initPromise (src/index.ts:18-19).  The counterpart: the GPU
  // runtime's start, the library's shared streams and the copy engines' first copy -- 0.2-0.35 s that the first
  // instances would otherwise pay inside their first processChunk -- run on a pool thread while the application goes
  // on loading; initPromise resolves behind them.  A failure (no GPU) is not reported here: the first processChunk
  // throws Error(strerror(code)) as always.  SPEEXHIP_NO_WARMUP=1 skips it.
  const done = () => { speexModule = m; return m; };
  if (process.env.SPEEXHIP_NO_WARMUP === '1' || typeof m.warmup !== 'function') return done();
  return m.warmup().then(done, done);
});

class SpeexResampler {
  /**
   * Same four arguments as the reference class: channel count (>= 1), input and output sample
   * rates in Hz, Speex quality 0..10 (default 7; higher = longer filter).
   */
  constructor(channels, inRate, outRate, quality = 7, options = undefined) {
    this.channels = channels;
    this.inRate = inRate;
    this.outRate = outRate;
    this.quality = quality;
    // Extension (ignored by reference-style callers): { device: k } pins this instance to GPU k.  Without it the
    // library's process-wide rule places the instance when its native state is made (first call): the calling
    // thread's current GPU, or -- SPEEXHIP_DEVICES=all -- instance number k of the process on GPU k mod the GPU count.
    this._device = options && Number.isInteger(options.device) ? options.device : -1;
    this._resamplerPtr = undefined; // native handle (the reference keeps a WASM pointer here)
    this._outBufferSize = -1;       // bytes; grow-only, drives the capacity rule below
    this._inFlight = 0;             // processChunkAsync calls not yet settled (extension)
  }

  // The reference's calls are synchronous and a state must not be used concurrently
  // (src/index.ts:50-116).  While a processChunkAsync call of this instance is running on a pool
  // thread, a synchronous call would interleave with it in an order the caller cannot know, so it
  // is refused; await the pending promise(s) first.
  _refuseWhileAsyncPending(what) {
    if (this._inFlight > 0) {
      throw new Error(what + ' called while a processChunkAsync call of this instance is pending; await it first');
    }
  }

  /**
   * One chunk of interleaved s16le PCM in, the resampled chunk out (a fresh Buffer), synchronously.
   */
  processChunk(chunk) {
    if (!speexModule) {
      throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
    }
    this._refuseWhileAsyncPending('processChunk');
    // reference src/index.ts:55-57 (channels === 0 gives NaN !== 0 and lands here too)
    if (chunk.length % (this.channels * Uint16Array.BYTES_PER_ELEMENT) !== 0) {
      throw new Error('Chunk length should be a multiple of channels * 2 bytes');
    }
    if (!this._resamplerPtr) {
      // throws Error(strerror(code)) and leaves _resamplerPtr unset, so a bad configuration
      // fails again on every call (reference src/index.ts:59-66)
      this._resamplerPtr = speexModule.init(this.channels >>> 0, this.inRate >>> 0,
        this.outRate >>> 0, this.quality | 0, this._device);
    }
    // reference src/index.ts:80-87: the output buffer only ever grows ...
    const outBufferLengthTarget = Math.ceil(chunk.length * this.outRate / this.inRate);
    if (this._outBufferSize < outBufferLengthTarget) {
      this._outBufferSize = outBufferLengthTarget;
    }
    // ... and its size in frames (truncated like setValue(..., 'i32')) caps this call's
    // output (src/index.ts:95).  Input frames the cap leaves unconsumed are dropped, as in the
    // reference, which never reads in_len back (src/index.ts:108).
    const inFrames = chunk.length / this.channels / Uint16Array.BYTES_PER_ELEMENT;
    const outCapacity = (this._outBufferSize / this.channels / Uint16Array.BYTES_PER_ELEMENT) | 0;
    return speexModule.process(this._resamplerPtr, chunk, inFrames | 0, outCapacity);
  }

  // ------------------------------------------------------------------------------------------
  // Extensions (not in the reference's class).  Nothing above depends on them.
  // ------------------------------------------------------------------------------------------

  /** Checks + lazy init + the capacity rule of processChunk, without the call itself. */
  _prepare(chunk, bytesPerSample) {
    if (!speexModule) {
      throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
    }
    if (chunk.length % (this.channels * bytesPerSample) !== 0) {
      throw new Error('Chunk length should be a multiple of channels * ' + bytesPerSample + ' bytes');
    }
    this._ensureNative();
    const target = Math.ceil(chunk.length * this.outRate / this.inRate);
    if (this._outBufferSize < target) {
      this._outBufferSize = target;
    }
    return [(chunk.length / this.channels / bytesPerSample) | 0,
      (this._outBufferSize / this.channels / bytesPerSample) | 0];
  }

  _ensureNative() {
    if (!speexModule) {
      throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
    }
    if (!this._resamplerPtr) {
      this._resamplerPtr = speexModule.init(this.channels >>> 0, this.inRate >>> 0,
        this.outRate >>> 0, this.quality | 0, this._device);
    }
    return this._resamplerPtr;
  }

  /**
   * processChunk for a list of consecutive chunks in ONE GPU launch (SURVEY 8f row N1).
   * Returns one Buffer per chunk, byte-identical to calling processChunk on each in turn
   * (the capacity rule is applied per original chunk).
   */
  processChunks(chunks) {
    this._refuseWhileAsyncPending('processChunks');
    const inFrames = [];
    const caps = [];
    for (const chunk of chunks) {
      const [f, cap] = this._prepare(chunk, Uint16Array.BYTES_PER_ELEMENT);
      inFrames.push(f);
      caps.push(cap);
    }
    return speexModule.processChunks(this._resamplerPtr, chunks, inFrames, caps);
  }

  /** processChunks off the event loop (one pool-thread call for the whole list); chained with the instance's other async calls. */
  processChunksAsync(chunks) {
    const inFrames = [], caps = [];
    try {
      for (const chunk of chunks) {
        const [f, cap] = this._prepare(chunk, Uint16Array.BYTES_PER_ELEMENT);
        inFrames.push(f);
        caps.push(cap);
      }
    } catch (e) {
      return Promise.reject(e);
    }
    const run = () => speexModule.processChunksAsync(this._resamplerPtr, chunks, inFrames, caps);
    this._inFlight++;
    const settle = () => { this._inFlight--; };
    const p = (this._pending || Promise.resolve()).then(run, run);
    this._pending = p.then(settle, settle);
    return p;
  }

  /**
   * processChunk that does not block the event loop: the transfer and the kernels run on a
   * libuv pool thread.  Calls on one instance are chained, so their order (and therefore the
   * stream) is the order of the calls.  Resolves to the same bytes processChunk returns.
   */
  processChunkAsync(chunk) {
    let args;
    try {
      args = this._prepare(chunk, Uint16Array.BYTES_PER_ELEMENT);
    } catch (e) {
      return Promise.reject(e);
    }
    // Calls of DIFFERENT instances that become ready in the same tick leave as ONE native call
    // (speexhip_resampler_process_many_int: per GPU one transfer in, one launch per <= 32 instances with the same
    // rates / quality / channels, one transfer out) -- a server's many connections are many instances
    // (reference src/index.ts:18-45), each with a small chunk per tick.  A lone call takes the single-state path.
    const run = () => new Promise((resolve, reject) => {
      tickQueue.push({ handle: this._resamplerPtr, chunk, inFrames: args[0], cap: args[1], resolve, reject });
      if (!tickScheduled) {
        tickScheduled = true;
        // (process.nextTick from inside a promise continuation runs once the continuations that are ready now have
        //  run: the calls of one tick are all in the queue by then)
        process.nextTick(flushTick);
      }
    });
    this._inFlight++;
    const settle = () => { this._inFlight--; };
    const p = (this._pending || Promise.resolve()).then(run, run);
    this._pending = p.then(settle, settle);
    return p;
  }

  /**
   * Float I/O (SURVEY 8f row N2, speex_resampler_process_interleaved_float): chunk is
   * interleaved float32 PCM; the result is float32, neither rounded nor clipped.  Same stream
   * state as processChunk; same grow-only capacity rule (in bytes of float32).
   */
  processChunkFloat(chunk) {
    this._refuseWhileAsyncPending('processChunkFloat');
    const [f, cap] = this._prepare(chunk, Float32Array.BYTES_PER_ELEMENT);
    return speexModule.processFloat(this._resamplerPtr, chunk, f, cap);
  }

  /** Mid-stream control (SURVEY 8f row N3; speex_resampler_set_rate / set_quality / ...). */
  setRate(inRate, outRate) {
    this._refuseWhileAsyncPending('setRate');
    if (this._resamplerPtr) speexModule.setRate(this._resamplerPtr, inRate >>> 0, outRate >>> 0);
    this.inRate = inRate;
    this.outRate = outRate;
  }

  setQuality(quality) {
    this._refuseWhileAsyncPending('setQuality');
    if (this._resamplerPtr) speexModule.setQuality(this._resamplerPtr, quality | 0);
    this.quality = quality;
  }

  /**
   * Arithmetic of this instance's kernels: 'fast_fixed' (the default since round 6: every sample within +-1 LSB of the
   * reference, fp64 sums where the reference has them -- quality 9 and 10 -- and, like the reference's, bytes that do not
   * depend on how the stream is cut into chunks, on how many streams share a launch or on the GPU), 'fast' (the same
   * tolerance; small launches of long filters may split a sum over several waves: up to 2x faster there, but the last bit
   * then depends on the chunking), 'exact' (bit-identical to the reference, slower), 'fast_f32' (one fp32 FMA chain for
   * every filter: the fast path of the first releases).  The environment variable SPEEXHIP_MODE sets the initial mode of
   * every instance.
   */
  setMode(mode) {
    this._refuseWhileAsyncPending('setMode');
    const code = { fast: 0, exact: 1, fast_f32: 2, fast_fixed: 3 }[mode];
    if (code === undefined) throw new Error("mode must be 'fast', 'exact', 'fast_f32' or 'fast_fixed'");
    speexModule.setMode(this._ensureNative(), code);
  }

  /** Start half a filter in, so the stream does not begin with the filter's ramp-up. */
  skipZeros() {
    this._refuseWhileAsyncPending('skipZeros');
    speexModule.skipZeros(this._ensureNative());
  }

  resetMem() {
    this._refuseWhileAsyncPending('resetMem');
    speexModule.resetMem(this._ensureNative());
  }

  /** Frames of delay the filter adds, counted at the input rate / at the output rate. */
  get inputLatency() { return speexModule.getLatency(this._ensureNative())[0]; }

  get outputLatency() { return speexModule.getLatency(this._ensureNative())[1]; }

  /**
   * The tail the reference never emits (SURVEY 8f row N4): feeds inputLatency frames of silence
   * and returns what comes out -- the response to the last real input frames.
   */
  flush() {
    this._refuseWhileAsyncPending('flush');
    const ptr = this._ensureNative();
    const frames = speexModule.getLatency(ptr)[0];
    const cap = Math.ceil(frames * this.outRate / this.inRate) + 1;
    return speexModule.process(ptr, null, frames, cap);
  }

  /** Release the GPU state now (otherwise it goes with garbage collection). */
  destroy() {
    this._refuseWhileAsyncPending('destroy');
    if (this._resamplerPtr) speexModule.destroy(this._resamplerPtr);
    this._resamplerPtr = undefined;
    this._outBufferSize = -1;
  }

  /** GPU this instance lives on (-1 until its native state exists). */
  get device() { return this._resamplerPtr ? speexModule.getInfo(this._resamplerPtr).device : this._device; }
}
SpeexResampler.initPromise = globalModulePromise;
/** Extension: GPUs the library can place instances on (<= 0: none -- there is no CPU fallback). */
SpeexResampler.deviceCount = () => {
  if (!speexModule) {
    throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
  }
  return speexModule.deviceCount();
};

/**
 * Extension (round 6): a Buffer of `bytes` bytes over a PINNED block of the library for the caller to fill -- read a file
 * or a socket into it, let a decoder write into it -- and hand to processChunk / processChunks / SpeexResamplerBatch.
 * The library recognises its own blocks and lets the GPU read the chunk where it lies, through PCIe, while it writes the
 * result block: no staging copy (the reference copies every chunk into the WASM heap, src/index.ts:71-92).  Results are
 * the same bytes as for an ordinary Buffer.  When no block is free an ordinary Buffer comes back (same results, the usual
 * staging).  The chunk may be refilled as soon as the call that read it has returned / its promise has settled.
 */
SpeexResampler.allocChunk = (bytes) => {
  if (!speexModule) {
    throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
  }
  return speexModule.allocChunk(bytes);
};

// processChunkAsync calls that became ready in this tick (see there)
const tickQueue = [];
let tickScheduled = false;
function flushTick() {
  tickScheduled = false;
  const jobs = tickQueue.splice(0, tickQueue.length);
  if (jobs.length === 0) return;
  if (jobs.length === 1) {
    const j = jobs[0];
    speexModule.processAsync(j.handle, j.chunk, j.inFrames, j.cap).then(j.resolve, j.reject);
    return;
  }
  let promise;
  try {
    promise = speexModule.processManyAsync(jobs.map((j) => j.handle), jobs.map((j) => j.chunk),
      jobs.map((j) => j.inFrames), jobs.map((j) => j.cap));
  } catch (e) {
    for (const j of jobs) j.reject(e);
    return;
  }
  promise.then((outs) => { jobs.forEach((j, i) => j.resolve(outs[i])); },
    (e) => {
      // The instances of a tick are independent (the reference's model: one failing instance does not touch the
      // others): when the native call ran, its error carries every entry's own code and result -- entries that
      // succeeded have advanced and get their audio, only the offending ones reject.
      if (e && Array.isArray(e.codes) && Array.isArray(e.results) && e.codes.length === jobs.length) {
        const first = e.codes.find((c) => c !== 0);
        jobs.forEach((j, i) => {
          if (e.codes[i] === 0) j.resolve(e.results[i]);
          else j.reject(e.codes[i] === first ? e : new Error(speexModule.strerror(e.codes[i])));
        });
      } else {
        for (const j of jobs) j.reject(e);
      }
    });
}

/**
 * Extension: n independent streams with one (channels, rates, quality) -- n SpeexResampler instances, each with its
 * own grow-only capacity rule (reference src/index.ts:80-95 per instance) -- whose chunks of one step travel and run
 * together: per GPU one transfer in, ONE launch per <= 32 streams, one transfer out.  With SPEEXHIP_DEVICES=all (or
 * options.devices = [0, 1, ...]) the streams spread over the node's GPUs, stream k on the k-th of them modulo their
 * number, and the GPUs work side by side.  Results are byte-identical (mode 'exact') / within +-1 LSB (default) to
 * calling processChunk on n separate instances.
 */
class SpeexResamplerBatch {
  constructor(nStreams, channels, inRate, outRate, quality = 7, options = undefined) {
    if (!Number.isInteger(nStreams) || nStreams < 1) throw new Error('nStreams must be a positive integer');
    this.channels = channels;
    this.inRate = inRate;
    this.outRate = outRate;
    this.quality = quality;
    const devices = options && Array.isArray(options.devices) && options.devices.length > 0 ? options.devices : null;
    this.streams = [];
    for (let k = 0; k < nStreams; k++) {
      this.streams.push(new SpeexResampler(channels, inRate, outRate, quality,
        devices ? { device: devices[k % devices.length] } : undefined));
    }
  }

  get length() { return this.streams.length; }

  _gather(chunks) {
    if (!Array.isArray(chunks) || chunks.length !== this.streams.length) {
      throw new Error('processChunks expects one chunk (or null) per stream: ' + this.streams.length);
    }
    // every chunk is checked before any stream's state is touched (lazy init, capacity rule): a step is refused whole
    if (!speexModule) {
      throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
    }
    for (let k = 0; k < chunks.length; k++) {
      if (chunks[k] === null || chunks[k] === undefined) continue;
      if (chunks[k].length % (this.channels * Uint16Array.BYTES_PER_ELEMENT) !== 0) {
        throw new Error('Chunk length should be a multiple of channels * 2 bytes');
      }
    }
    const picked = { index: [], handles: [], chunks: [], inFrames: [], caps: [] };
    for (let k = 0; k < chunks.length; k++) {
      if (chunks[k] === null || chunks[k] === undefined) continue; // this stream sits the step out
      const [f, cap] = this.streams[k]._prepare(chunks[k], Uint16Array.BYTES_PER_ELEMENT);
      picked.index.push(k);
      picked.handles.push(this.streams[k]._resamplerPtr);
      picked.chunks.push(chunks[k]);
      picked.inFrames.push(f);
      picked.caps.push(cap);
    }
    return picked;
  }

  /**
   * One chunk of interleaved s16le PCM per stream (null / undefined: the stream sits this step out) in, one fresh
   * Buffer per stream out (null for streams that sat out): entry k is what streams[k].processChunk(chunks[k]) returns.
   */
  processChunks(chunks) {
    for (const r of this.streams) r._refuseWhileAsyncPending('SpeexResamplerBatch.processChunks');
    const g = this._gather(chunks);
    const outs = g.handles.length > 0 ? speexModule.processMany(g.handles, g.chunks, g.inFrames, g.caps) : [];
    const result = new Array(this.streams.length).fill(null);
    g.index.forEach((k, i) => { result[k] = outs[i]; });
    return result;
  }

  /** processChunks off the event loop; steps are chained, so the streams advance in the order of the calls. */
  processChunksAsync(chunks) {
    let g;
    try {
      g = this._gather(chunks);
    } catch (e) {
      return Promise.reject(e);
    }
    const run = () => (g.handles.length > 0
      ? speexModule.processManyAsync(g.handles, g.chunks, g.inFrames, g.caps) : Promise.resolve([]))
      .then((outs) => {
        const result = new Array(this.streams.length).fill(null);
        g.index.forEach((k, i) => { result[k] = outs[i]; });
        return result;
      });
    // (while the step is pending the synchronous methods of its streams refuse, like after processChunkAsync: the
    //  results were sized from the streams' counters when the step was queued)
    for (const k of g.index) this.streams[k]._inFlight++;
    const settle = () => { for (const k of g.index) this.streams[k]._inFlight--; };
    // The step waits for the batch's previous step AND for what its streams have pending of their own
    // (streams[k].processChunkAsync), and later asynchronous calls of those streams wait for the step: a stream's calls
    // stay in the order they were made, whichever object they were made through (ADVICE r5 -- the native call
    // checks the same thing again under its locks and refuses the step rather than overrun a result).
    const before = [this._pending].concat(g.index.map((k) => this.streams[k]._pending)).filter(Boolean);
    const p = Promise.all(before.map((q) => q.then(() => {}, () => {}))).then(run);
    const done = p.then(settle, settle);
    this._pending = done;
    for (const k of g.index) this.streams[k]._pending = done;
    return p;
  }

  setMode(mode) { for (const r of this.streams) r.setMode(mode); }

  destroy() { for (const r of this.streams) r.destroy(); }
}
/**
 * Extension.  Destroyed (or collected) states leave their device buffers, pinned staging buffers and
 * filter tables in a process-wide pool for the next `new SpeexResampler` (DESIGN 3.5); this hands
 * whatever is idle there back to the driver and returns the number of bytes.
 */
SpeexResampler.releaseCachedMemory = () => {
  if (!speexModule) {
    throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
  }
  return speexModule.releaseCachedMemory();
};

const EMPTY_BUFFER = Buffer.alloc(0);

class SpeexResamplerTransform extends Transform {
  /**
   * Same arguments as SpeexResampler (reference src/index.ts:121-137).  The optional fifth
   * argument is an extension and changes nothing unless set:
   *   coalesceChunks: n  -- hold up to n aligned chunks and resample them in one GPU launch
   *                         (bytes out unchanged, they just arrive n chunks at a time)
   *   async: true        -- run each call off the event loop (processChunkAsync)
   *   flushTail: true    -- at end of stream also emit SpeexResampler.flush()
   *   pipeline: true     -- off the event loop AND batched by load: a chunk that arrives while a call is in flight is
   *                         held, and everything held leaves as ONE call when that one returns (an idle stream sends each
   *                         chunk at once; a busy one sends few, large launches).  Same bytes out.
   *   pinned: true       -- every chunk is copied ONCE into a pinned chunk of the library (SpeexResampler.allocChunk: a ring
   *                         of them, reused) and the GPU reads it there -- in the modes that hold chunks (coalesceChunks,
   *                         async, pipeline) that is the copy they make anyway, landing in pinned memory instead of an
   *                         ordinary Buffer.  Same bytes out.  (A producer that can fill chunks from allocChunk itself needs
   *                         no option: pinned chunks are recognised wherever they arrive.)
   */
  constructor(channels, inRate, outRate, quality = 7, options = undefined) {
    super();
    this.channels = channels;
    this.inRate = inRate;
    this.outRate = outRate;
    this.quality = quality;
    this.resampler = new SpeexResampler(channels, inRate, outRate, quality);
    this._alignementBuffer = EMPTY_BUFFER;
    this._options = options || {};
    this._held = [];
    if (this._options.pipeline) {
      this._busy = false;       // a processChunksAsync call is in flight
      this._waiting = null;     // the end of the stream, parked until the pipeline has drained
      this._parked = null;      // a _transform callback parked while maxHeld chunks are held
      this._flush = (callback) => {
        const finish = () => {
          try {
            if (this._options.flushTail) this.push(this.resampler.flush());
            callback();
          } catch (e) {
            callback(e);
          }
        };
        if (!this._busy && this._held.length === 0) finish();
        else this._waiting = finish;
        this._pump();
      };
    } else if (this._options.coalesceChunks > 1 || this._options.flushTail) {
      // only defined when asked for: the reference has no _flush
      this._flush = (callback) => {
        try {
          this._emitHeld();
          if (this._options.flushTail) this.push(this.resampler.flush());
          callback();
        } catch (e) {
          callback(e);
        }
      };
    }
  }

  // pipeline: send what is held as one asynchronous call; when it returns, push its results in order and go again
  _pump() {
    if (this._busy || this._held.length === 0) return;
    const batch = this._held;
    this._held = [];
    this._busy = true;
    if (this._parked) {  // room again: the producer may go on
      const cb = this._parked;
      this._parked = null;
      cb();
    }
    this.resampler.processChunksAsync(batch).then((outs) => {
      for (const out of outs) this.push(out);
      for (const h of batch) this._recycle(h);
      this._busy = false;
      if (this._held.length > 0) {
        this._pump();
      } else if (this._waiting) {
        const w = this._waiting;
        this._waiting = null;
        w();
      }
    }, (e) => {
      this._busy = false;
      this._waiting = null;
      this._parked = null;
      this.destroy(e);
    });
  }

  // pinned: a copy of `chunk` in a pinned chunk of the library (a plain copy when none is to be had).  Chunks come from a
  // small free list by size class; one is free again once the call that read it has returned -- _recycle().
  _pinnedCopy(chunk) {
    if (chunk.length === 0) return chunk;
    if (!this._ring) this._ring = new Map();
    let cls = 4096;
    while (cls < chunk.length) cls *= 2;
    const free = this._ring.get(cls);
    const slab = free && free.length > 0 ? free.pop() : SpeexResampler.allocChunk(cls);
    chunk.copy(slab, 0, 0, chunk.length);
    const view = slab.slice(0, chunk.length);
    view._slab = slab;
    view._cls = cls;
    return view;
  }

  _recycle(view) {
    if (!view || !view._slab || !this._ring) return;
    let free = this._ring.get(view._cls);
    if (!free) this._ring.set(view._cls, free = []);
    if (free.length < 8) free.push(view._slab);
  }

  // the copy a held chunk needs anyway (it must not change under us while it waits for its call)
  _own(chunk) { return this._options.pinned ? this._pinnedCopy(chunk) : Buffer.from(chunk); }

  _emitHeld() {
    if (this._held.length === 0) return;
    const held = this._held;
    this._held = [];
    for (const out of this.resampler.processChunks(held)) this.push(out);
    for (const h of held) this._recycle(h);
  }

  _transform(chunk, encoding, callback) {
    if (this._options.pipeline) {
      try {
        // (a copy: the held chunk must not change under us while it waits for its call)
        this._held.push(this._own(this._align(chunk)));
      } catch (e) {
        callback(e);
        return;
      }
      this._pump();
      // accept the next chunk at once -- unless so much is held already that the producer should wait for the GPU
      if (this._held.length < (this._options.maxHeld || 256)) callback();
      else this._parked = callback;
      return;
    }
    if (this._options.coalesceChunks > 1 || this._options.async) {
      return this._transformExtended(chunk, callback);
    }
    if (this._options.pinned) {  // the reference's one synchronous call per chunk, on a pinned copy of the chunk
      let out;
      try {
        const own = this._pinnedCopy(this._align(chunk));
        out = this.resampler.processChunk(own);
        this._recycle(own);
      } catch (err) {
        callback(err);
        return;
      }
      callback(null, out);
      return;
    }
    return this._transformReference(chunk, encoding, callback);
  }

  /**
   * Whole frames of (carry ++ chunk); the 0 .. channels*2-1 bytes that do not complete a frame
   * are carried to the next chunk in `_alignementBuffer` (the field name, typo included, is the
   * reference's: src/index.ts:148-154 keeps it on the instance).
   */
  _align(chunk) {
    const frameBytes = this.channels * Uint16Array.BYTES_PER_ELEMENT;
    const carry = this._alignementBuffer;
    const joined = carry.length > 0 ? Buffer.concat([carry, chunk]) : chunk;
    const whole = joined.length - (joined.length % frameBytes);
    // (a copy: the carried bytes must survive the caller reusing its chunk)
    this._alignementBuffer = whole === joined.length ? EMPTY_BUFFER : Buffer.from(joined.slice(whole));
    return whole === joined.length ? joined : joined.slice(0, whole);
  }

  _transformExtended(chunk, callback) {
    try {
      // copy: a held chunk must not change under us while it waits for its launch
      const aligned = this._own(this._align(chunk));
      if (this._options.coalesceChunks > 1) {
        this._held.push(aligned);
        if (this._held.length >= this._options.coalesceChunks) this._emitHeld();
        callback();
      } else {
        this.resampler.processChunkAsync(aligned).then((res) => {
          this._recycle(aligned);
          callback(null, res);
        }, callback);
      }
    } catch (e) {
      callback(e);
    }
  }

  // the reference's behaviour: one synchronous call per chunk, output pushed at once
  _transformReference(chunk, encoding, callback) {
    let out;
    try {
      out = this.resampler.processChunk(this._align(chunk));
    } catch (err) {
      callback(err);
      return;
    }
    callback(null, out);
  }
}

exports.SpeexResamplerTransform = SpeexResamplerTransform;
exports.SpeexResamplerBatch = SpeexResamplerBatch;
exports.default = SpeexResampler;
