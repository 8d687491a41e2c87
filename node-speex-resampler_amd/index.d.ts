/// <reference types="node" />
import { Transform } from 'stream';
/** Same public surface as the reference's app/index.d.ts; computation runs on an MI355X. */
declare class SpeexResampler {
    channels: any;
    inRate: any;
    outRate: any;
    quality: number;
    _resamplerPtr: any;
    _outBufferSize: number;
    static initPromise: Promise<any>;
    /**
      * Create an SpeexResampler tranform stream.
      * @param channels Number of channels, minimum is 1, no maximum
      * @param inRate frequency in Hz for the input chunk
      * @param outRate frequency in Hz for the target chunk
      * @param quality number from 1 to 10, default to 7, 1 is fast but of bad quality, 10 is slow but best quality
      */
    constructor(channels: any, inRate: any, outRate: any, quality?: number);
    /**
      * Resample a chunk of audio.
      * @param chunk interleaved PCM data in signed 16bits int
      */
    processChunk(chunk: Buffer): Buffer;
    /** Extension: consecutive chunks in one GPU launch; result[i] === processChunk(chunks[i]). */
    processChunks(chunks: Buffer[]): Buffer[];
    /** Extension: processChunk off the event loop; calls on one instance stay in order. */
    processChunkAsync(chunk: Buffer): Promise<Buffer>;
    /** Extension: interleaved float32 PCM in and out (speex_resampler_process_interleaved_float). */
    processChunkFloat(chunk: Buffer): Buffer;
    /** Extensions: mid-stream control (speex_resampler_set_rate / set_quality / skip_zeros / reset_mem). */
    setRate(inRate: number, outRate: number): void;
    setQuality(quality: number): void;
    skipZeros(): void;
    resetMem(): void;
    readonly inputLatency: number;
    readonly outputLatency: number;
    /** Extension: the filter's tail (response to the last inputLatency frames). */
    flush(): Buffer;
    /** Extension: release the GPU state now. */
    destroy(): void;
}
export interface SpeexResamplerTransformOptions {
    /** hold up to n chunks and resample them in one GPU launch (same bytes out) */
    coalesceChunks?: number;
    /** run each call off the event loop */
    async?: boolean;
    /** at end of stream also emit the filter's tail */
    flushTail?: boolean;
}
export declare class SpeexResamplerTransform extends Transform {
    channels: any;
    inRate: any;
    outRate: any;
    quality: number;
    resampler: SpeexResampler;
    _alignementBuffer: Buffer;
    constructor(channels: any, inRate: any, outRate: any, quality?: number, options?: SpeexResamplerTransformOptions);
    _transform(chunk: any, encoding: any, callback: any): void;
}
export default SpeexResampler;
