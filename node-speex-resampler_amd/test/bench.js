'use strict';
// Timing counterpart of the reference's src/test.ts ("Resampled in N ms"): its seven (rates, channels,
// quality) tuples on synthetic PCM of the byte lengths of its resource files, (a) one processChunk over
// the whole buffer, (b) piped through SpeexResamplerTransform in 64 KiB chunks (what createReadStream
// hands over), (c) the same pipe with the coalescing / async extensions.  Host buffers in and out, so
// every figure includes PCIe and the N-API hop -- this is what a Node caller sees, not the kernel rate
// bench.py reports.  The reference's WASM on a CPU, same tuples: profiles/r02_reference_wasm_cpu.json.
//   node test/bench.js [out.json]      (needs an MI355X)
const fs = require('fs');
const { Readable, Writable } = require('stream');
const { performance } = require('perf_hooks');
const mod = require('../index.js');
const SpeexResampler = mod.default;
const { SpeexResamplerTransform, SpeexResamplerBatch } = mod;

const tuples = [
  { bytes: 882044, inRate: 24000, outRate: 48000, channels: 1, quality: 5 },
  { bytes: 1764044, inRate: 24000, outRate: 24000, channels: 2, quality: 5 },
  { bytes: 1764044, inRate: 24000, outRate: 48000, channels: 2, quality: 10 },
  { bytes: 1764044, inRate: 44100, outRate: 48000, channels: 2, quality: 7 },
  { bytes: 1764044, inRate: 44100, outRate: 48000, channels: 2, quality: 10 },
  { bytes: 1764044, inRate: 44100, outRate: 48000, channels: 2, quality: 1 },
  { bytes: 1764044, inRate: 44100, outRate: 24000, channels: 2, quality: 5 },
];

function pcm(bytes, seed) {
  const buf = Buffer.alloc(bytes);
  let s = seed >>> 0;
  for (let i = 0; i + 1 < bytes; i += 2) {
    s = (Math.imul(s, 1664525) + 1013904223) >>> 0;
    buf.writeInt16LE(((s >>> 16) & 0x7fff) - 16384, i);
  }
  return buf;
}

const median = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];

async function pipeOnce(t, data, options) {
  const tr = new SpeexResamplerTransform(t.channels, t.inRate, t.outRate, t.quality, options);
  let out = 0;
  const sink = new Writable({ write(c, _e, cb) { out += c.length; cb(); } });
  const chunks = [];
  for (let off = 0; off < data.length; off += 65536) chunks.push(data.slice(off, Math.min(off + 65536, data.length)));
  const t0 = performance.now();
  Readable.from(chunks, { objectMode: false }).pipe(tr).pipe(sink);
  await new Promise((res, rej) => { sink.on('finish', res); tr.on('error', rej); });
  return [performance.now() - t0, out];
}

(async () => {
  await SpeexResampler.initPromise;
  const rows = [];
  {  // V8 and the stream machinery warm up over the first few hundred calls: not what is measured here
    const t = tuples[3], data = pcm(t.bytes, 1);
    for (let rep = 0; rep < 15; rep++) {
      await pipeOnce(t, data, undefined);
      await pipeOnce(t, data, { async: true });
      await pipeOnce(t, data, { coalesceChunks: 8 });
      await pipeOnce(t, data, { pipeline: true });
    }
  }
  for (const t of tuples) {
    const data = pcm(t.bytes, 12345);
    const row = Object.assign({}, t);
    // (a) whole buffer, fresh state per run like the reference's test; first run (filter design, device
    // buffers) reported separately
    const whole = [];
    for (let rep = 0; rep < 12; rep++) {
      const r = new SpeexResampler(t.channels, t.inRate, t.outRate, t.quality);
      const t0 = performance.now();
      const out = r.processChunk(data);
      whole.push(performance.now() - t0);
      row.out_bytes = out.length;
      if (r.destroy) r.destroy();
    }
    row.whole_first_ms = +whole[0].toFixed(3);
    row.whole_ms = +median(whole.slice(2)).toFixed(3);
    // steady state: one state, the buffer again and again (a long stream in file-sized pieces)
    {
      const r = new SpeexResampler(t.channels, t.inRate, t.outRate, t.quality);
      r.processChunk(data);
      const ts = [];
      for (let rep = 0; rep < 20; rep++) {
        const t0 = performance.now();
        r.processChunk(data);
        ts.push(performance.now() - t0);
      }
      row.steady_ms = +median(ts).toFixed(3);
      if (r.destroy) r.destroy();
    }
    for (const [name, options] of [['pipe_ms', undefined], ['pipe_coalesce8_ms', { coalesceChunks: 8 }],
      ['pipe_async_ms', { async: true }], ['pipe_pipeline_ms', { pipeline: true }]]) {
      const ts = [];
      for (let rep = 0; rep < 7; rep++) ts.push((await pipeOnce(t, data, options))[0]);
      row[name] = +median(ts.slice(1)).toFixed(3);
    }
    row.input_msamples_per_s_whole = +(t.bytes / 2 / row.whole_ms / 1e3).toFixed(1);
    row.input_msamples_per_s_steady = +(t.bytes / 2 / row.steady_ms / 1e3).toFixed(1);
    console.log(JSON.stringify(row));
    rows.push(row);
  }
  // round 6: chunks in pinned memory (SpeexResampler.allocChunk) against ordinary Buffers -- a 2^20-frame stereo chunk through
  // processChunk, 32 instances x 16384 frames through SpeexResamplerBatch, and the 64 KiB pipe with { pinned: true }
  const pinnedRows = {};
  {
    const t = tuples[3];
    const big = pcm(4 << 20, 7);
    const bigPinned = SpeexResampler.allocChunk(big.length);
    big.copy(bigPinned);
    const r = new SpeexResampler(t.channels, t.inRate, t.outRate, t.quality);
    for (const [name, chunk] of [['chunk_4MiB_ordinary_ms', big], ['chunk_4MiB_allocChunk_ms', bigPinned]]) {
      for (let i = 0; i < 4; i++) r.processChunk(chunk);
      const ts = [];
      for (let rep = 0; rep < 30; rep++) {
        const t0 = performance.now();
        r.processChunk(chunk);
        ts.push(performance.now() - t0);
        // (results are external Buffers over pinned blocks; their finalizers run from the event loop -- a server yields
        //  to it all the time, a benchmark loop has to: otherwise the slabs fill up and the calls fall back to copies)
        if (rep % 5 === 4) { if (global.gc) global.gc(); await new Promise((res) => setImmediate(res)); }
      }
      pinnedRows[name] = +median(ts).toFixed(4);
    }
    r.destroy();
    const N = 32, small = pcm(65536, 9);
    const batch = new SpeexResamplerBatch(N, t.channels, t.inRate, t.outRate, t.quality);
    const ordinary = new Array(N).fill(small);
    const pinned = ordinary.map((c) => { const p = SpeexResampler.allocChunk(c.length); c.copy(p); return p; });
    for (const [name, chunks] of [['batch32_64KiB_ordinary_ms', ordinary], ['batch32_64KiB_allocChunk_ms', pinned]]) {
      for (let i = 0; i < 10; i++) batch.processChunks(chunks);
      const ts = [];
      for (let rep = 0; rep < 100; rep++) {
        const t0 = performance.now();
        batch.processChunks(chunks);
        ts.push(performance.now() - t0);
        if (rep % 5 === 4) { if (global.gc) global.gc(); await new Promise((res) => setImmediate(res)); }
      }
      pinnedRows[name] = +median(ts).toFixed(4);
    }
    batch.destroy();
    const data = pcm(t.bytes, 12345);
    for (const [name, options] of [['pipe_ms', undefined], ['pipe_pinned_ms', { pinned: true }],
      ['pipe_pipeline_pinned_ms', { pipeline: true, pinned: true }], ['pipe_coalesce8_pinned_ms', { coalesceChunks: 8, pinned: true }]]) {
      const ts = [];
      for (let rep = 0; rep < 9; rep++) ts.push((await pipeOnce(t, data, options))[0]);
      pinnedRows[name] = +median(ts.slice(1)).toFixed(3);
    }
    console.log(JSON.stringify({ pinned_chunks: pinnedRows }));
  }
  if (process.argv[2]) {
    fs.writeFileSync(process.argv[2], JSON.stringify({
      what: 'index.js drop-in on one MI355X: host Buffers in and out (PCIe + N-API included); whole = one ' +
        'processChunk over the buffer with a fresh state (median of 10 after 2 warm-ups), steady = the same call ' +
        'on a running state, pipe = SpeexResamplerTransform fed 64 KiB chunks; pinned_chunks (round 6) = the same calls on chunks ' +
        'from SpeexResampler.allocChunk (44.1k -> 48k stereo q7)', rows, pinned_chunks: pinnedRows }, null, 1) + '\n');
  }
})().catch((e) => { console.error(e); process.exit(1); });
