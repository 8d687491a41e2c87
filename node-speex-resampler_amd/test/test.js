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

function lcgFloat(frames, channels, seed) {
  const pcm = lcg(frames, channels, seed);
  const f = new Float32Array(frames * channels);
  for (let i = 0; i < f.length; i++) f[i] = pcm.readInt16LE(i * 2) / 32768;
  return Buffer.from(f.buffer);
}

// SURVEY 8(f) rows N1-N4 through the JS / N-API surface.
async function extensionsTest() {
  const addon = require('../speex_hip_napi.node');
  const mk = (mode) => {
    const r = new SpeexResampler(2, 44100, 48000, 7);
    r.processChunk(Buffer.alloc(0));
    addon.setMode(r._resamplerPtr, mode);
    return r;
  };
  // N1: coalesced chunks and async calls return the bytes of the separate synchronous calls --
  // incl. 100-frame chunks, where the capacity rule drops input (F5)
  const sizes = [4410, 100, 100, 7, 0, 4410, 333, 100, 2, 9000];
  const chunks = sizes.map((n, i) => lcg(n, 2, 900 + i));
  for (const mode of [1, 0]) {
    const a = mk(mode), b = mk(mode), c = mk(mode);
    const want = chunks.map((ch) => a.processChunk(ch));
    const got = b.processChunks(chunks);
    const promised = await Promise.all(chunks.map((ch) => c.processChunkAsync(ch))); // queued at once
    assert(got.length === want.length, 'processChunks: count');
    for (let i = 0; i < want.length; i++) {
      assert(got[i].equals(want[i]), `processChunks: chunk ${i} differs (mode ${mode})`);
      assert(promised[i].equals(want[i]), `processChunkAsync: chunk ${i} differs (mode ${mode})`);
    }
    const ia = addon.getInfo(a._resamplerPtr), ib = addon.getInfo(b._resamplerPtr), ic = addon.getInfo(c._resamplerPtr);
    assert(ia.last_sample === ib.last_sample && ia.samp_frac_num === ib.samp_frac_num, 'processChunks: state');
    assert(ia.last_sample === ic.last_sample && ia.samp_frac_num === ic.samp_frac_num, 'processChunkAsync: state');
  }
  const rejected = await new SpeexResampler(2, 44100, 48000).processChunkAsync(Buffer.alloc(7)).then(() => null, (e) => e.message);
  assert(rejected === 'Chunk length should be a multiple of channels * 2 bytes', 'async length check');

  // A synchronous call while an asynchronous one is pending: the class refuses it (message names
  // the cure); the addon itself -- reached directly, as a misbehaving caller could -- serialises
  // the two calls on the handle's lock: both return, every byte belongs to one of the two possible
  // orders, nothing is written past a Buffer (the result Buffer is sized under the same lock that
  // the call holds).
  {
    const big = lcg(1 << 18, 2, 77), small = lcg(1000, 2, 78);
    const r = mk(1);
    const p = r.processChunkAsync(big);
    let refused = null;
    try { r.processChunk(small); } catch (e) { refused = e.message; }
    assert(refused !== null && refused.indexOf('processChunkAsync call of this instance is pending') > 0, 'sync call during async must be refused');
    for (const f of [() => r.setRate(44100, 32000), () => r.setQuality(3), () => r.flush(), () => r.destroy()]) {
      let m = null;
      try { f(); } catch (e) { m = e.message; }
      assert(m !== null && m.indexOf('pending') > 0, 'control call during async must be refused');
    }
    const first = await p;
    const second = r.processChunk(small);           // accepted again once settled
    const ref = mk(1);
    assert(first.equals(ref.processChunk(big)) && second.equals(ref.processChunk(small)), 'order after the refusal');

    for (let round = 0; round < 8; round++) {
      const a = mk(1), order1 = mk(1), order2 = mk(1);
      const capBig = Math.ceil(big.length * 48000 / 44100 / 4), capSmall = Math.ceil(small.length * 48000 / 44100 / 4);
      const pa = addon.processAsync(a._resamplerPtr, big, big.length / 4, capBig);
      const sync = addon.process(a._resamplerPtr, small, small.length / 4, capSmall);   // races the pool thread
      addon.getInfo(a._resamplerPtr); addon.getLatency(a._resamplerPtr);
      const asyncOut = await pa;
      const b1 = addon.process(order1._resamplerPtr, big, big.length / 4, capBig);
      const s1 = addon.process(order1._resamplerPtr, small, small.length / 4, capSmall);
      const s2 = addon.process(order2._resamplerPtr, small, small.length / 4, capSmall);
      const b2 = addon.process(order2._resamplerPtr, big, big.length / 4, capBig);
      const asyncFirst = asyncOut.equals(b1) && sync.equals(s1), syncFirst = asyncOut.equals(b2) && sync.equals(s2);
      assert(asyncFirst || syncFirst, `racing sync/async calls on one handle: bytes match neither order (round ${round})`);
      const ia = addon.getInfo(a._resamplerPtr), io = addon.getInfo((asyncFirst ? order1 : order2)._resamplerPtr);
      assert(ia.last_sample === io.last_sample && ia.samp_frac_num === io.samp_frac_num, 'racing calls: state');
      // control call racing an async call: must not crash or corrupt (it lands before or after)
      const pc = addon.processAsync(a._resamplerPtr, big, big.length / 4, capBig);
      addon.setQuality(a._resamplerPtr, 3 + (round % 5));
      assert((await pc).length > 0, 'async call racing setQuality');
      for (const x of [a, order1, order2]) x.destroy();
    }
    console.log('sync/async race on one handle: serialised, bytes match one of the two orders');
  }

  // N1: the Transform options leave the bytes alone
  const pcm = lcg(44100, 2, 31337);
  const parts = [];
  for (let o = 0; o < pcm.length; o += 4097) parts.push(pcm.slice(o, o + 4097));
  const pipe = async (options) => {
    const t = new SpeexResamplerTransform(2, 44100, 48000, 7, options);
    let out = Buffer.alloc(0);
    t.on('data', (d) => { out = Buffer.concat([out, d]); });
    Readable.from(parts).pipe(t);
    await new Promise((res, rej) => { t.on('end', res); t.on('error', rej); });
    return [out, t];
  };
  const [plain] = await pipe(undefined);
  const [coalesced] = await pipe({ coalesceChunks: 8 });
  const [asynced] = await pipe({ async: true });
  const [tailed, tt] = await pipe({ coalesceChunks: 5, flushTail: true });
  assert(coalesced.equals(plain), 'Transform coalesceChunks changed the bytes');
  assert(asynced.equals(plain), 'Transform async changed the bytes');
  assert(tailed.slice(0, plain.length).equals(plain), 'Transform flushTail changed the stream');
  const tailFrames = (tailed.length - plain.length) / 4;
  assert(Math.abs(tailFrames - tt.resampler.outputLatency) <= 2, `flush tail of ${tailFrames} frames`);

  // N2 + N3 through the addon: the reference's recorded control scripts (EXACT mode)
  let ops = 0;
  for (const c of golden.control_cases.slice(0, 12)) {
    const h = addon.init(c.channels, c.in_rate, c.out_rate, c.quality);
    addon.setMode(h, 1);
    c.ops.forEach((op, k) => {
      const want = c.results[k];
      const kind = op[0];
      let head = [];
      if (kind === 'int' || kind === 'float' || kind === 'int_null' || kind === 'float_null') {
        const isFloat = kind.startsWith('float');
        const buf = kind.endsWith('_null') ? null : (isFloat ? lcgFloat(op[1], c.channels, op[3]) : lcg(op[1], c.channels, op[3]));
        const out = (isFloat ? addon.processFloat : addon.process)(h, buf, op[1], op[2]);
        head = [out.length / c.channels / (isFloat ? 4 : 2), sha1(out).slice(0, 16)];
        assert(head[0] === want[1] && head[1] === want[2], `${c.name} op ${k} ${JSON.stringify(op)}: output differs`);
      } else {
        let rc = 0;
        try {
          if (kind === 'rate') addon.setRate(h, op[1], op[2]);
          else if (kind === 'ratefrac') addon.setRate(h, op[1], op[2], op[3], op[4]);
          else if (kind === 'quality') addon.setQuality(h, op[1]);
          else if (kind === 'skip') addon.skipZeros(h);
          else addon.resetMem(h);
        } catch (e) { rc = e.message; }
        assert(rc === (want[0] === 0 ? 0 : addon.strerror(want[0])), `${c.name} op ${k}: rc ${rc}`);
      }
      const st = want.slice(want.length - 10);
      const i = addon.getInfo(h), lat = addon.getLatency(h);
      const got = [i.last_sample, i.samp_frac_num, i.magic_samples, i.filt_len, lat[0], lat[1], i.in_rate, i.out_rate, i.num_rate, i.den_rate];
      assert(JSON.stringify(got) === JSON.stringify(st), `${c.name} op ${k}: state ${got} want ${st}`);
      ops++;
    });
    addon.destroy(h);
    let dead = null;
    try { addon.process(h, Buffer.alloc(0), 0, 0); } catch (e) { dead = e.message; }
    assert(dead === 'Bad resampler state.', 'destroyed handle must refuse calls');
  }
  console.log(`control-script ops replayed through the addon: ${ops}`);

  // N3/N4 on the class: setters before and after the lazy init, flush, destroy + re-init
  const r = new SpeexResampler(1, 16000, 48000, 4);
  r.setQuality(6);
  assert(r.processChunk(lcg(1600, 1, 5)).length === 4800 * 2, 'x3 upsampling length');
  r.setRate(16000, 8000);
  assert(addon.getRate(r._resamplerPtr).join() === '16000,8000', 'setRate reached the native state');
  assert(r.inputLatency > 0 && r.outputLatency > 0, 'latencies');
  const f32 = r.processChunkFloat(lcgFloat(1600, 1, 6));
  assert(f32.length % 4 === 0 && f32.length / 4 >= 700 && f32.length / 4 <= 800, 'float call on the same state');
  assert(r.flush().length > 0, 'flush emits the tail');
  r.skipZeros(); r.resetMem();
  r.destroy();
  assert(r._resamplerPtr === undefined && r.processChunk(lcg(160, 1, 7)).length > 0, 'destroy then lazy re-init');
  let msg = null;
  try { r.setQuality(42); } catch (e) { msg = e.message; }
  assert(msg === 'Invalid argument.', 'setQuality(42) must throw the reference message');
}

// States are cheap to make and drop: a destroyed state's device resources wait in a pool for the next
// one, and releaseCachedMemory() hands them back; bytes out must not depend on any of that.
function poolTest() {
  const chunk = lcg(20000, 2, 99);
  const first = new SpeexResampler(2, 44100, 48000, 7);
  const want = sha1(first.processChunk(chunk));
  first.destroy();
  for (let k = 0; k < 20; k++) {
    const r = new SpeexResampler(2, 44100, 48000, 7);
    assert(sha1(r.processChunk(chunk)) === want, 'a recycled state must produce the same bytes');
    if (k % 2) r.destroy();  // the others wait for the garbage collector
  }
  const released = SpeexResampler.releaseCachedMemory();
  assert(released > 0 || process.env.SPEEXHIP_POOL_MB === '0', 'destroyed states left memory in the pool');
  const r = new SpeexResampler(2, 44100, 48000, 7);
  assert(sha1(r.processChunk(chunk)) === want, 'a state made after the release must produce the same bytes');
  r.destroy();
  console.log('state pool: recycled and released, bytes unchanged');
}

// Round 4: results of >= 4 KB are external Buffers over pinned blocks of the library's pool.  They are the
// caller's: writable, untouched by later calls, alive after the state is gone, and their blocks go back to the
// pool when the collector drops them (node --expose-gc lets the test see that; without it the check is skipped).
// setMode: 'exact' reproduces the reference's sha1 goldens through the class itself; 'fast' and 'fast_f32' stay within
// +-1 LSB of it at quality 10 (fp64 accumulate / fp32 chain), 'fast' with fewer samples off by one
function modeTest() {
  const pcm = lcg(60000, 1, 31);
  const outs = {};
  for (const mode of ['exact', 'fast', 'fast_f32']) {
    const r = new SpeexResampler(1, 24000, 48000, 10);
    r.setMode(mode);
    outs[mode] = r.processChunk(pcm);
    r.destroy();
  }
  let bad = false;
  try { new SpeexResampler(1, 24000, 48000, 10).setMode('fastest'); } catch (e) { bad = /mode must be/.test(e.message); }
  assert(bad, 'setMode must refuse unknown modes');
  const off = {};
  for (const mode of ['fast', 'fast_f32']) {
    assert(outs[mode].length === outs.exact.length, 'modes must agree on the counters');
    let n = 0;
    for (let i = 0; i < outs.exact.length; i += 2) {
      const d = Math.abs(outs[mode].readInt16LE(i) - outs.exact.readInt16LE(i));
      assert(d <= 1, `${mode}: a sample ${d} LSB away from the exact kernels`);
      if (d) n++;
    }
    off[mode] = n;
  }
  assert(off.fast <= off.fast_f32, 'fp64 accumulate must not be further from the reference than the fp32 chain');
  console.log(`modes: fast ${off.fast} / fast_f32 ${off.fast_f32} of ${outs.exact.length / 2} samples off by one`);
}

function externalBufferTest() {
  const r = new SpeexResampler(2, 44100, 48000, 7);
  const a = r.processChunk(lcg(30000, 2, 5));
  const keep = sha1(a);
  const copyOfA = Buffer.from(a);
  for (let k = 0; k < 8; k++) r.processChunk(lcg(30000, 2, 6 + k));   // later calls: other blocks
  assert(sha1(a) === keep, 'a returned Buffer must not change under later calls');
  a.writeInt16LE(1234, 0);                                              // the caller may write to it
  assert(a.readInt16LE(0) === 1234 && a.length === copyOfA.length, 'returned Buffers are writable');
  r.destroy();
  assert(a.readInt16LE(0) === 1234 && a.slice(2).equals(copyOfA.slice(2)), 'a returned Buffer outlives its state');
  if (global.gc) {
    let held = [];
    const r2 = new SpeexResampler(2, 44100, 48000, 7);
    for (let k = 0; k < 40; k++) held.push(r2.processChunk(lcg(200000, 2, k)));  // ~35 MB of blocks held
    held = null;
    global.gc();
    r2.destroy();
    const released = SpeexResampler.releaseCachedMemory();
    assert(released > 0 || process.env.SPEEXHIP_POOL_MB === '0' || process.env.SPEEXHIP_NAPI_COPY === '1',
      'collected Buffers must hand their blocks back to the pool');
  }
  console.log('external Buffers: caller-owned, stable, recycled');
}

(async () => {
  const early = (() => { try { new SpeexResampler(1, 8000, 8000).processChunk(Buffer.alloc(2)); return null; } catch (e) { return e.message; } })();
  assert(early === 'You need to wait for SpeexResampler.initPromise before calling this method', 'initPromise guard');
  await SpeexResampler.initPromise;
  await promiseBasedTest();
  await streamBasedTest();
  goldenTest();
  errorTest();
  await extensionsTest();
  poolTest();
  externalBufferTest();
  modeTest();
  console.log('ALL NODE TESTS PASSED');
})().catch((e) => { console.error(e); process.exit(1); });
