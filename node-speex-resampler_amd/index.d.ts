/// <reference types="node" />
// Type declarations of the drop-in (hand-written).  The reference ships tsc output that types `channels`, `inRate` and
// `outRate` as `any` (app/index.d.ts:4-6, 23, 31-33, 44): so do the constructors and fields below -- a TypeScript caller
// that compiles against the reference (rates held as strings, say: they are coerced like the reference coerces them)
// compiles against the drop-in.
import { Transform, TransformCallback } from 'stream';

/**
 * Options of the streaming wrapper; all of them are extensions and default to off.  Without options the
 * Transform is the reference's, byte for byte, and it is also the FASTEST way through a pipe: measured on one
 * MI355X with the seven files of the reference's test (1.7 MB each, 64 KiB chunks, profiles/r02_node_bench.json),
 * the plain pipe takes 0.76-1.49 ms per file.  The options below buy something else than throughput.
 */
export interface SpeexResamplerTransformOptions {
    /**
     * Hold up to n chunks and resample them in one GPU launch (same bytes out).  Trades LATENCY -- the first
     * output waits for n chunks -- for fewer launches: 20-55 % less time per file at n = 8 on five of the seven test
     * tuples (round 5: the held run stays on pinned memory), a loss on mono 24k -> 48k and stereo 24k -> 48k q10.
     */
    coalesceChunks?: number;
    /**
     * Run each call off the event loop (N-API async work).  Trades THROUGHPUT for an event loop that stays
     * free while the GPU works: a file takes 1.5-2.2x as long as through the plain pipe (two thread hand-overs
     * per 64 KiB chunk cost more than the 25-50 us call they wrap).  For servers that must not block; not a
     * way to go faster.
     */
    async?: boolean;
    /** at end of stream also emit the filter's tail */
    flushTail?: boolean;
    /**
     * Off the event loop AND batched by load: a chunk that arrives while a call is in flight is held, and everything
     * held leaves as ONE call when that one returns -- an idle stream sends each chunk at once, a busy one a few large
     * launches.  Same bytes out.  What `async` is for -- an event loop that stays free -- at about the plain pipe's
     * speed instead of 1.5-4x its time (profiles/r05_node_bench.json: 0.63-0.90 ms per 1.7 MB file, plain pipe
     * 0.47-1.08, `async` 0.83-3.6).  `maxHeld` (default 256) bounds what is held before the producer is made to wait.
     */
    pipeline?: boolean;
    maxHeld?: number;
    /**
     * Copy every chunk ONCE into a pinned chunk of the library (a small ring of `SpeexResampler.allocChunk` Buffers,
     * reused) so that the GPU reads it in place.  With `coalesceChunks`, `async` or `pipeline` this is the copy those
     * modes make anyway, landing in pinned memory; in the plain mode it replaces the library's own copy into its bounce
     * buffer.  Same bytes out.  A producer that fills chunks from `allocChunk` itself needs no option.
     */
    pinned?: boolean;
}

/**
 * Speex resampler whose filter runs on an MI355X.  Constructor, `processChunk`, `initPromise`
 * and the instance fields are the reference's public surface; the rest are extensions.
 */
declare class SpeexResampler {
    /** resolves once the native module is loaded; `processChunk` throws before that */
    static initPromise: Promise<unknown>;

    channels: any;
    inRate: any;
    outRate: any;
    /** Speex quality, 0..10 */
    quality: number;
    /** native state handle, created by the first call */
    _resamplerPtr: unknown;
    /** grow-only byte size that caps the frames one call may return */
    _outBufferSize: number;

    /**
     * The reference's four arguments.  `options.device` (extension) pins the instance to one GPU; without it the
     * library places it when its native state is made: the process's current GPU, or -- environment
     * SPEEXHIP_DEVICES=all -- instance number k of the process on GPU k mod the GPU count (SPEEXHIP_DEVICE=k: all on k).
     */
    constructor(channels: any, inRate: any, outRate: any, quality?: number, options?: { device?: number });

    /** GPU the instance lives on (-1 until the first call made its native state) */
    readonly device: number;
    /** GPUs the library can place instances on */
    static deviceCount(): number;
    /**
     * Extension: a Buffer over a PINNED block of the native library for the caller to fill (read a file or a socket into
     * it, let a decoder write it) and pass to any processChunk* method or to SpeexResamplerBatch: the GPU reads the chunk
     * where it lies, through PCIe, while it writes the result -- no staging copy (the reference copies every chunk into
     * the WASM heap).  Same result bytes as for an ordinary Buffer; an ordinary Buffer comes back when no block is free.
     * Refill it once the call that read it has returned (or its promise has settled).
     */
    static allocChunk(bytes: number): Buffer;

    /**
     * interleaved s16le PCM in, resampled s16le PCM out: a fresh Buffer the caller owns, like the reference's.
     * Results of 4 KB and more are external Buffers over pinned memory of the native library (no copy on the way
     * out; the memory returns to the library when the Buffer is collected); an application that keeps more than
     * 64 MiB of them alive (SPEEXHIP_TAKE_MB) gets ordinary copies beyond that.  SPEEXHIP_NAPI_COPY=1: copies always.
     */
    processChunk(chunk: Buffer): Buffer;

    /** consecutive chunks in one GPU launch; result[i] equals processChunk(chunks[i]) */
    processChunks(chunks: Buffer[]): Buffer[];
    /** the same off the event loop */
    processChunksAsync(chunks: Buffer[]): Promise<Buffer[]>;
    /**
     * processChunk off the event loop; calls on one instance stay in order.  Calls of DIFFERENT instances that become
     * ready in the same tick leave as one native call: one transfer in, one GPU launch per <= 32 instances of equal
     * (channels, rates, quality) and GPU, one transfer out (a server's connections, each with a small chunk per tick).  While one is pending the
     * synchronous methods of the same instance (processChunk, processChunks, processChunkFloat, setRate,
     * setQuality, skipZeros, resetMem, flush, destroy) throw: await the promise first.
     */
    processChunkAsync(chunk: Buffer): Promise<Buffer>;
    /** interleaved float32 PCM in and out (speex_resampler_process_interleaved_float) */
    processChunkFloat(chunk: Buffer): Buffer;

    /** mid-stream control (speex_resampler_set_rate / set_quality / skip_zeros / reset_mem) */
    setRate(inRate: number, outRate: number): void;
    setQuality(quality: number): void;
    /**
     * 'fast_fixed' (the default: +-1 LSB of the reference, fp64 sums at quality 9 / 10, and bytes that -- like the
     * reference's -- do not depend on chunking, batch size or GPU), 'fast' (+-1 LSB; up to 2x faster on one-stream calls of
     * long decimators, but the last bit of a sample may depend on how the stream was cut into chunks), 'exact'
     * (bit-identical to the reference, slower) or 'fast_f32'.  SPEEXHIP_MODE in the environment sets the initial mode.
     */
    setMode(mode: 'fast' | 'exact' | 'fast_f32' | 'fast_fixed'): void;
    skipZeros(): void;
    resetMem(): void;
    /** filter delay in frames at the input rate / at the output rate */
    readonly inputLatency: number;
    readonly outputLatency: number;

    /** the filter's tail: the response to the last inputLatency frames */
    flush(): Buffer;
    /** release the GPU state now instead of at garbage collection */
    destroy(): void;
    /**
     * Extension: hand the idle device / pinned memory that destroyed states left in the process-wide
     * pool (kept for the next `new SpeexResampler`) back to the driver; returns the bytes released.
     */
    static releaseCachedMemory(): number;
}

/** `stream.Transform` around a SpeexResampler; misaligned trailing bytes wait for the next chunk. */
export declare class SpeexResamplerTransform extends Transform {
    channels: any;
    inRate: any;
    outRate: any;
    quality: number;
    resampler: SpeexResampler;
    _alignementBuffer: Buffer;

    constructor(channels: any, inRate: any, outRate: any, quality?: number,
                options?: SpeexResamplerTransformOptions);
    _transform(chunk: Buffer, encoding: string, callback: TransformCallback): void;
}

/**
 * Extension: n independent streams of one (channels, rates, quality), each a SpeexResampler with its own capacity
 * rule, whose chunks of one step run together -- per GPU one transfer in, one launch per <= 32 streams, one transfer
 * out; with SPEEXHIP_DEVICES=all or `options.devices` the streams spread over the node's GPUs (stream k on the k-th
 * listed GPU modulo their number).  Entry k of a result equals `streams[k].processChunk(chunks[k])`.
 */
export declare class SpeexResamplerBatch {
    channels: any;
    inRate: any;
    outRate: any;
    quality: number;
    streams: SpeexResampler[];
    readonly length: number;
    constructor(nStreams: number, channels: any, inRate: any, outRate: any, quality?: number,
                options?: { devices?: number[] });
    /** one chunk per stream (null: the stream sits the step out) -> one fresh Buffer per stream (null likewise) */
    processChunks(chunks: Array<Buffer | null>): Array<Buffer | null>;
    processChunksAsync(chunks: Array<Buffer | null>): Promise<Array<Buffer | null>>;
    setMode(mode: 'fast' | 'exact' | 'fast_f32' | 'fast_fixed'): void;
    destroy(): void;
}

export default SpeexResampler;
