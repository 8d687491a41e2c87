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
}).then((m) => { speexModule = m; return m; });

class SpeexResampler {
  /**
   * @param channels Number of channels, minimum is 1, no maximum
   * @param inRate frequency in Hz for the input chunk
   * @param outRate frequency in Hz for the target chunk
   * @param quality number from 1 to 10, default to 7 (0 is accepted too, as in the reference)
   */
  constructor(channels, inRate, outRate, quality = 7) {
    this.channels = channels;
    this.inRate = inRate;
    this.outRate = outRate;
    this.quality = quality;
    this._resamplerPtr = undefined; // native handle (the reference keeps a WASM pointer here)
    this._outBufferSize = -1;       // bytes; grow-only, drives the capacity rule below
  }

  /**
   * Resample a chunk of audio.
   * @param chunk interleaved PCM data in signed 16bits int
   */
  processChunk(chunk) {
    if (!speexModule) {
      throw new Error('You need to wait for SpeexResampler.initPromise before calling this method');
    }
    // reference src/index.ts:55-57 (channels === 0 gives NaN !== 0 and lands here too)
    if (chunk.length % (this.channels * Uint16Array.BYTES_PER_ELEMENT) !== 0) {
      throw new Error('Chunk length should be a multiple of channels * 2 bytes');
    }
    if (!this._resamplerPtr) {
      // throws Error(strerror(code)) and leaves _resamplerPtr unset, so a bad configuration
      // fails again on every call (reference src/index.ts:59-66)
      this._resamplerPtr = speexModule.init(this.channels >>> 0, this.inRate >>> 0,
        this.outRate >>> 0, this.quality | 0);
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
}
SpeexResampler.initPromise = globalModulePromise;

const EMPTY_BUFFER = Buffer.alloc(0);

class SpeexResamplerTransform extends Transform {
  /** Same arguments as SpeexResampler (reference src/index.ts:121-137). */
  constructor(channels, inRate, outRate, quality = 7) {
    super();
    this.channels = channels;
    this.inRate = inRate;
    this.outRate = outRate;
    this.quality = quality;
    this.resampler = new SpeexResampler(channels, inRate, outRate, quality);
    this._alignementBuffer = EMPTY_BUFFER;
  }

  _transform(chunk, encoding, callback) {
    let chunkToProcess = chunk;
    if (this._alignementBuffer.length > 0) {
      chunkToProcess = Buffer.concat([this._alignementBuffer, chunk]);
      this._alignementBuffer = EMPTY_BUFFER;
    }
    // whole frames only; the 0..(channels*2-1) trailing bytes wait for the next chunk
    // (reference src/index.ts:148-154)
    const extraneousBytesCount = chunkToProcess.length % (this.channels * Uint16Array.BYTES_PER_ELEMENT);
    if (extraneousBytesCount !== 0) {
      this._alignementBuffer = Buffer.from(chunkToProcess.slice(chunkToProcess.length - extraneousBytesCount));
      chunkToProcess = chunkToProcess.slice(0, chunkToProcess.length - extraneousBytesCount);
    }
    try {
      const res = this.resampler.processChunk(chunkToProcess);
      callback(null, res);
    } catch (e) {
      callback(e);
    }
  }
}

exports.SpeexResamplerTransform = SpeexResamplerTransform;
exports.default = SpeexResampler;
