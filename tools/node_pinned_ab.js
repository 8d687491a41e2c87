'use strict';
// tools/node_pinned_ab.js -- what SpeexResampler.allocChunk buys a Node caller (round 6), measured with the legs ALTERNATING so
// that both see the same garbage-collector state: a 4 MiB stereo chunk through processChunk, and 32 instances x 64 KiB
// through SpeexResamplerBatch.processChunks, ordinary Buffers against chunks from allocChunk.  Run it twice: as is, and with
// SPEEXHIP_NAPI_COPY=1 (results as ordinary Buffers).   node --expose-gc tools/node_pinned_ab.js
const { performance } = require('perf_hooks');
const mod = require('../node-speex-resampler_amd/index.js');
const SpeexResampler = mod.default;
const { SpeexResamplerBatch } = mod;
const median = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];
const min = (a) => Math.min.apply(null, a);
function pcm(bytes, seed) {
  const buf = Buffer.alloc(bytes);
  let s = seed >>> 0;
  for (let i = 0; i + 1 < bytes; i += 2) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; buf.writeInt16LE(((s >>> 16) & 0x7fff) - 16384, i); }
  return buf;
}
(async () => {
  await SpeexResampler.initPromise;
  const addon = require('../node-speex-resampler_amd/speex_hip_napi.node');
  const out = { napi_copy: process.env.SPEEXHIP_NAPI_COPY === '1' };
  {
    const big = pcm(4 << 20, 7), pinned = SpeexResampler.allocChunk(big.length);
    big.copy(pinned);
    const a = new SpeexResampler(2, 44100, 48000, 7), b = new SpeexResampler(2, 44100, 48000, 7);
    for (let i = 0; i < 5; i++) { a.processChunk(big); b.processChunk(pinned); }
    const ta = [], tb = [];
    for (let rep = 0; rep < 40; rep++) {
      let t0 = performance.now(); a.processChunk(big); ta.push(performance.now() - t0);
      t0 = performance.now(); b.processChunk(pinned); tb.push(performance.now() - t0);
      // (the results are external Buffers over the library's pinned blocks; their finalizers run from the event loop:
      //  a loop that never yields keeps every block until the slabs are full and the calls fall back to copies)
      if (rep % 5 === 4) { if (global.gc) global.gc(); await new Promise((r) => setImmediate(r)); }
    }
    out.chunk_4MiB = { ordinary_ms: +median(ta).toFixed(4), ordinary_min: +min(ta).toFixed(4), allocChunk_ms: +median(tb).toFixed(4), allocChunk_min: +min(tb).toFixed(4) };
    a.destroy(); b.destroy();
  }
  for (const bytes of [65536, 1 << 20]) {
    const N = 32, small = pcm(bytes, 9);
    const ba = new SpeexResamplerBatch(N, 2, 44100, 48000, 7), bb = new SpeexResamplerBatch(N, 2, 44100, 48000, 7);
    const ordinary = new Array(N).fill(small);
    const pinned = ordinary.map((c) => { const p = SpeexResampler.allocChunk(c.length); c.copy(p); return p; });
    for (let i = 0; i < 10; i++) { ba.processChunks(ordinary); bb.processChunks(pinned); }
    const ta = [], tb = [];
    for (let rep = 0; rep < 100; rep++) {
      let t0 = performance.now(); ba.processChunks(ordinary); ta.push(performance.now() - t0);
      t0 = performance.now(); bb.processChunks(pinned); tb.push(performance.now() - t0);
      if (rep % 5 === 4) { if (global.gc) global.gc(); await new Promise((r) => setImmediate(r)); }
    }
    out['batch32_' + bytes] = { ordinary_ms: +median(ta).toFixed(4), ordinary_min: +min(ta).toFixed(4), allocChunk_ms: +median(tb).toFixed(4), allocChunk_min: +min(tb).toFixed(4) };
    ba.destroy(); bb.destroy();
  }
  out.stats = addon.stats();
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
