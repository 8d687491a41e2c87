'use strict';
// Node harness for the drop-in (counterpart of the reference's src/test.ts): the same seven
// (rates, channels, quality) tuples, whole-buffer and piped through SpeexResamplerTransform in
// 64 KiB chunks, with the reference's duration assertion (src/test.ts:40,74) -- plus sample-level
// checks the reference never had: sha1 goldens captured from the reference (tests/golden/
// golden.json; EXACT mode must match them bit-for-bit) and the F5 small-chunk capacity case.
// Needs an MI355X.  `await SpeexResampler.initPromise` first (the reference test forgot to).
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const { Readable } = require('stream');
const mod = require('../index.js');
const SpeexResampler = mod.default;
const { SpeexResamplerTransform } = mod;

const golden = JSON.parse(fs.readFileSync(path.join(__dirname, '../../tests/golden/golden.json')));
const assert = (c, m) => { if (!c) throw new Error(m); };
const sha1 = (b) => crypto.createHash('sha1').update(b).digest('hex');

function lcg(frames, channels, seed) {
  const buf = Buffer.alloc(frames * channels * 2);
  let s = seed >>> 0;
  for (let i = 0; i < frames * channels; i++) {
    s = (Math.imul(s, 1664525) + 1013904223) >>> 0;
    buf.writeUInt16LE(s >>> 16, i * 2);
  }
  return buf;
}

const audioTests = [
  { inRate: 24000, outRate: 48000, channels: 1, quality: 5 },
  { inRate: 24000, outRate: 24000, channels: 2, quality: 5 },
  { inRate: 24000, outRate: 48000, channels: 2, quality: 10 },
  { inRate: 44100, outRate: 48000, channels: 2 },
  { inRate: 44100, outRate: 48000, channels: 2, quality: 10 },
  { inRate: 44100, outRate: 48000, channels: 2, quality: 1 },
  { inRate: 44100, outRate: 24000, channels: 2, quality: 5 },
];

async function promiseBasedTest() {
  for (const t of audioTests) {
    const pcm = lcg(t.inRate * 2, t.channels, 4242); // 2 s of audio
    const r = new SpeexResampler(t.channels, t.inRate, t.outRate, t.quality);
    const t0 = Date.now();
    const res = await r.processChunk(pcm);
    const inDur = pcm.length / t.inRate / 2 / t.channels;
    const outDur = res.length / t.outRate / 2 / t.channels;
    console.log(`${t.inRate}->${t.outRate} ${t.channels}ch q${t.quality || 7}: ${Date.now() - t0} ms, ` +
      `${inDur}s -> ${outDur.toFixed(4)}s`);
    assert(Math.abs(inDur - outDur) < 0.01, `Stream duration not matching target, in: ${inDur}s != out:${outDur}`);
  }
}

async function streamBasedTest() {
  for (const t of audioTests) {
    const pcm = lcg(t.inRate * 2, t.channels, 4242);
    const chunks = [];
    for (let o = 0; o < pcm.length; o += 65536 + 1) chunks.push(pcm.slice(o, o + 65536 + 1)); // misaligned on purpose
    const transform = new SpeexResamplerTransform(t.channels, t.inRate, t.outRate, t.quality);
    let out = Buffer.alloc(0);
    transform.on('data', (d) => { out = Buffer.concat([out, d]); });
    Readable.from(chunks).pipe(transform);
    await new Promise((res, rej) => { transform.on('end', res); transform.on('error', rej); });
    const inDur = pcm.length / t.inRate / 2 / t.channels;
    const outDur = out.length / t.outRate / 2 / t.channels;
    assert(Math.abs(inDur - outDur) < 0.01, `transform duration mismatch ${inDur} vs ${outDur}`);
    assert(out.length % (t.channels * 2) === 0, 'transform emitted a partial frame');
  }
}

function goldenTest() {
  const addon = require('../speex_hip_napi.node');
  let checked = 0;
  for (const c of golden.cases) {
    if (c.input !== 'lcg' || c.chunks === 'ragged' || c.frames > 200000) continue;
    const pcm = lcg(c.frames, c.channels, c.seed);
    assert(sha1(pcm) === c.input_sha1, `${c.name}: input generator mismatch`);
    const sizes = c.chunks === 'whole' ? [pcm.length] : [];
    if (c.chunks !== 'whole') for (let o = 0; o < pcm.length; o += c.chunks) sizes.push(Math.min(c.chunks, pcm.length - o));
    for (const mode of [1, 0]) { // 1 = EXACT (bit-identical), 0 = FAST (+-1 LSB)
      const r = new SpeexResampler(c.channels, c.in_rate, c.out_rate, c.quality);
      let out = Buffer.alloc(0), off = 0, first = true;
      for (const n of sizes) {
        if (first) { r.processChunk(Buffer.alloc(0)); addon.setMode(r._resamplerPtr, mode); first = false; }
        out = Buffer.concat([out, r.processChunk(pcm.slice(off, off + n))]);
        off += n;
      }
      assert(out.length / 2 / c.channels === c.out_frames, `${c.name}: ${out.length / 2 / c.channels} frames, want ${c.out_frames}`);
      if (mode === 1) {
        assert(sha1(out) === c.out_sha1, `${c.name}: EXACT mode differs from the reference bytes`);
      } else if (c.out_full) {
        for (let i = 0; i < c.out_full.length; i++) {
          assert(Math.abs(out.readInt16LE(i * 2) - c.out_full[i]) <= 1, `${c.name}: sample ${i} off by more than 1 LSB`);
        }
      }
    }
    checked++;
  }
  assert(checked >= 8, `only ${checked} golden cases checked`);
  console.log(`golden cases checked through the JS API: ${checked}`);
}

function errorTest() {
  const msg = (f) => { try { f(); return null; } catch (e) { return e.message; } };
  assert(msg(() => new SpeexResampler(2, 44100, 48000).processChunk(Buffer.alloc(7))) ===
    'Chunk length should be a multiple of channels * 2 bytes', 'length check');
  assert(msg(() => new SpeexResampler(2, 44100, 48000, 11).processChunk(Buffer.alloc(8))) === 'Invalid argument.', 'q11');
  const bad = new SpeexResampler(2, 0, 48000);
  assert(msg(() => bad.processChunk(Buffer.alloc(8))) === 'Invalid argument.', 'rate 0');
  assert(msg(() => bad.processChunk(Buffer.alloc(8))) === 'Invalid argument.', 'rate 0 retried');
  assert(new SpeexResampler(2, 44100, 48000).processChunk(Buffer.alloc(0)).length === 0, 'empty chunk');
  assert(new SpeexResampler(1, 8000, 8000, 0).processChunk(Buffer.alloc(320)).length > 0, 'quality 0 accepted');
}

(async () => {
  const early = (() => { try { new SpeexResampler(1, 8000, 8000).processChunk(Buffer.alloc(2)); return null; } catch (e) { return e.message; } })();
  assert(early === 'You need to wait for SpeexResampler.initPromise before calling this method', 'initPromise guard');
  await SpeexResampler.initPromise;
  await promiseBasedTest();
  await streamBasedTest();
  goldenTest();
  errorTest();
  console.log('ALL NODE TESTS PASSED');
})().catch((e) => { console.error(e); process.exit(1); });
