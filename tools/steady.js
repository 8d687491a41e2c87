'use strict';
// tools/steady.js -- VERDICT r4 #5: 24k->48k stereo q10 through processChunk took 0.16 ms on a fresh state and
// 0.43 ms on a running one (profiles/r04_node_bench.json).  Per call: time, whether the result was an external Buffer
// over a pinned block or a copy (addon.stats()), with and without event-loop turns between the calls.
const { performance } = require('perf_hooks');
const mod = require('../node-speex-resampler_amd/index.js');
const addon = require('../node-speex-resampler_amd/speex_hip_napi.node');
const SpeexResampler = mod.default;
function pcm(bytes, seed) {
  const buf = Buffer.alloc(bytes);
  let s = seed >>> 0;
  for (let i = 0; i + 1 < bytes; i += 2) {
    s = (Math.imul(s, 1664525) + 1013904223) >>> 0;
    buf.writeInt16LE(((s >>> 16) & 0x7fff) - 16384, i);
  }
  return buf;
}
(async () => {
  await SpeexResampler.initPromise;
  for (const t of [{ bytes: 1764044, inRate: 24000, outRate: 48000, channels: 2, quality: 10 },
    { bytes: 1764044, inRate: 44100, outRate: 48000, channels: 2, quality: 7 }]) {
    const data = pcm(t.bytes, 12345);
    for (const yieldBetween of [false, true]) {
      const r = new SpeexResampler(t.channels, t.inRate, t.outRate, t.quality);
      const rows = [];
      for (let rep = 0; rep < 40; rep++) {
        const s0 = addon.stats();
        const t0 = performance.now();
        const out = r.processChunk(data);
        const dt = performance.now() - t0;
        const s1 = addon.stats();
        rows.push(`${dt.toFixed(3)}${s1.takeNoBlock > s0.takeNoBlock ? 'C' : 'x'}`);
        void out;
        if (yieldBetween) await new Promise((res) => setImmediate(res));
      }
      console.log(`${t.inRate}->${t.outRate} q${t.quality} ${yieldBetween ? 'with' : 'no'} event-loop turns; ms per call (x = external Buffer, C = slab full, copied): ${rows.join(' ')}`);
      console.log('  stats', JSON.stringify(addon.stats()));
      r.destroy();
    }
  }
})().catch((e) => { console.error(e); process.exit(1); });
