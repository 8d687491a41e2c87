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
const { SpeexResamplerTransform, SpeexResamplerBatch } = mod;

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

// VERDICT r5 #8: the round-5 surface -- SpeexResamplerBatch.processChunks and the same-tick coalescer of
// processChunkAsync -- anchored to the REFERENCE's bytes (tests/golden/golden.json: digests captured from the reference's
// own C), not to separate instances of this library: the small-chunk capacity case (F5: 640-byte chunks, input dropped),
// a ragged 8-channel down-sampler, a ragged up-sampler and the reference's own stream test tuple (44.1k -> 48k stereo q7 in
// 64 KiB chunks; its tonal input comes from the Python harness, checked against the golden input digest).  EXACT mode
// must reproduce out_sha1 through both paths; FAST within +-1 LSB of those bytes.  Also (round 6) with every chunk in a
// pinned chunk from SpeexResampler.allocChunk, and through the Transform's `pinned` option.
async function goldenBatchTest() {
  const extra = process.env.SPEEXHIP_TEST_INPUTS ? JSON.parse(fs.readFileSync(process.env.SPEEXHIP_TEST_INPUTS)) : {};
  const names = ['f5_640B_chunks', 'f5_ragged_down_8ch', 'f5_ragged_up', 't_44100_48000_2ch_q7_64k'];
  let ran = 0;
  for (const name of names) {
    const c = golden.cases.find((x) => x.name === name);
    assert(c, `golden case ${name} missing`);
    let pcm;
    if (c.input === 'lcg') pcm = lcg(c.frames, c.channels, c.seed);
    else if (extra[name]) pcm = Buffer.from(extra[name], 'base64');
    else { console.log(`goldenBatchTest: ${name} needs its input from the Python harness (SPEEXHIP_TEST_INPUTS): skipped`); continue; }
    assert(sha1(pcm) === c.input_sha1, `${name}: input differs from the golden input`);
    // the call sequence: recorded per call for ragged cases, a fixed size otherwise
    const fb = 2 * c.channels, sizes = [];
    if (c.chunks === 'ragged') {
      assert(c.calls.length === c.n_calls, `${name}: the golden file holds ${c.calls.length} of ${c.n_calls} calls`);
      for (const row of c.calls) sizes.push(row[0] * fb);
    } else {
      for (let o = 0; o < pcm.length; o += c.chunks) sizes.push(Math.min(c.chunks, pcm.length - o));
    }
    const cut = (own) => {
      const v = [];
      let off = 0;
      for (const n of sizes) {
        let ch = pcm.slice(off, off + n);
        if (own) { const p = SpeexResampler.allocChunk(Math.max(n, 1)); ch.copy(p); ch = p.slice(0, n); }
        v.push(ch);
        off += n;
      }
      return v;
    };
    const N = 3;
    let exactBytes = null;
    for (const mode of ['exact', 'fast']) {
      for (const pinned of [false, true]) {
        const chunks = cut(pinned);
        // (1) the batch class: N streams, each fed the golden stream
        const batch = new SpeexResamplerBatch(N, c.channels, c.in_rate, c.out_rate, c.quality);
        batch.processChunks(new Array(N).fill(Buffer.alloc(0)));
        batch.setMode(mode);
        const outs = Array.from({ length: N }, () => []);
        for (const ch of chunks) batch.processChunks(new Array(N).fill(ch)).forEach((o, k) => outs[k].push(o));
        // (2) the coalescer: N instances whose processChunkAsync calls become ready in the same tick
        const inst = Array.from({ length: N }, () => {
          const r = new SpeexResampler(c.channels, c.in_rate, c.out_rate, c.quality);
          r.setMode(mode);
          return r;
        });
        const couts = Array.from({ length: N }, () => []);
        for (const ch of chunks) (await Promise.all(inst.map((r) => r.processChunkAsync(ch)))).forEach((o, k) => couts[k].push(o));
        for (const [what, list] of [['SpeexResamplerBatch', outs], ['coalescer', couts]]) {
          list.forEach((parts, k) => {
            const out = Buffer.concat(parts);
            assert(out.length / fb === c.out_frames, `${name} ${what} stream ${k} (${mode}${pinned ? ', pinned chunks' : ''}): ${out.length / fb} frames, want ${c.out_frames}`);
            if (mode === 'exact') {
              assert(sha1(out) === c.out_sha1, `${name} ${what} stream ${k}${pinned ? ' (pinned chunks)' : ''}: EXACT bytes differ from the reference digest`);
              exactBytes = out;
            } else {
              for (let i = 0; i < out.length; i += 2) {
                assert(Math.abs(out.readInt16LE(i) - exactBytes.readInt16LE(i)) <= 1, `${name} ${what} stream ${k}: FAST sample ${i / 2} off by more than 1 LSB`);
              }
            }
          });
        }
        batch.destroy();
        inst.forEach((r) => r.destroy());
      }
    }
    // (3) the Transform with `pinned` (alone and with the modes that hold chunks): the golden bytes again
    if (c.chunks !== 'ragged') {
      for (const options of [{ pinned: true }, { pinned: true, coalesceChunks: 4 }, { pinned: true, pipeline: true }, { pinned: true, async: true }]) {
        const t = new SpeexResamplerTransform(c.channels, c.in_rate, c.out_rate, c.quality, options);
        t.resampler.setMode('exact');
        const parts = [];
        t.on('data', (d) => parts.push(d));
        Readable.from(cut(false)).pipe(t);
        await new Promise((res, rej) => { t.on('end', res); t.on('error', rej); });
        assert(sha1(Buffer.concat(parts)) === c.out_sha1, `${name} Transform ${JSON.stringify(options)}: bytes differ from the reference digest`);
      }
    }
    ran++;
  }
  assert(ran >= 3, 'goldenBatchTest ran too few cases');
  // a chunk that starts anywhere inside a pinned block (Buffer.subarray of an allocChunk block): read where it lies, at
  // whatever alignment, with the bytes of the same chunk in an ordinary Buffer
  {
    const n = 2 * 2 * 30000;
    const src = Buffer.alloc(n);
    for (let k = 0; k < n; k += 2) src.writeInt16LE(((k * 2654435761) >>> 17) % 60000 - 30000, k);
    const want = new SpeexResampler(2, 44100, 48000, 7).processChunk(src);
    for (const off of [2, 6, 10, 64]) {
      const block = SpeexResampler.allocChunk(n + 128);
      src.copy(block, off);
      const got = new SpeexResampler(2, 44100, 48000, 7).processChunk(block.subarray(off, off + n));
      assert(Buffer.compare(got, want) === 0, `pinned chunk at byte offset ${off}: bytes differ`);
    }
  }
  const st = require('../speex_hip_napi.node').stats();
  if (process.env.SPEEXHIP_NAPI_COPY !== '1') assert(st.pinnedChunks > 0, 'allocChunk never returned a pinned chunk');
  console.log(`golden cases through SpeexResamplerBatch / the coalescer / pinned chunks / Transform{pinned}: ${ran} (pinned chunks handed out: ${st.pinnedChunks})`);
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
    const d = mk(mode);
    const listed = (await Promise.all([d.processChunksAsync(chunks.slice(0, 4)), d.processChunksAsync(chunks.slice(4))]))
      .reduce((x, y) => x.concat(y), []);
    for (let i = 0; i < want.length; i++) assert(listed[i].equals(want[i]), `processChunksAsync: chunk ${i} differs (mode ${mode})`);
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
  const [piped] = await pipe({ pipeline: true });
  const [pipedTight] = await pipe({ pipeline: true, maxHeld: 2 });
  const [pipedTail] = await pipe({ pipeline: true, flushTail: true });
  assert(piped.equals(plain), 'Transform pipeline changed the bytes');
  assert(pipedTight.equals(plain), 'Transform pipeline (maxHeld 2) changed the bytes');
  const [tailed, tt] = await pipe({ coalesceChunks: 5, flushTail: true });
  assert(pipedTail.equals(tailed), 'Transform pipeline + flushTail differs from coalesce + flushTail');
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
  for (const mode of ['exact', 'fast', 'fast_f32', 'fast_fixed']) {
    const r = new SpeexResampler(1, 24000, 48000, 10);
    r.setMode(mode);
    outs[mode] = r.processChunk(pcm);
    r.destroy();
  }
  let bad = false;
  try { new SpeexResampler(1, 24000, 48000, 10).setMode('fastest'); } catch (e) { bad = /mode must be/.test(e.message); }
  assert(bad, 'setMode must refuse unknown modes');
  const off = {};
  for (const mode of ['fast', 'fast_f32', 'fast_fixed']) {
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
  // 'fast_fixed': the bytes of a stream do not depend on how it is cut into chunks (the reference's property)
  {
    const whole = new SpeexResampler(2, 48000, 11025, 7), cut = new SpeexResampler(2, 48000, 11025, 7);
    // (round 6: no mode is set -- the property belongs to the DEFAULT mode)
    whole.processChunk(Buffer.alloc(0)); cut.processChunk(Buffer.alloc(0));
    const x = lcg(200000, 2, 77);
    const addon = require('../speex_hip_napi.node');
    if (!process.env.SPEEXHIP_MODE) assert(addon.getInfo(whole._resamplerPtr).mode === 3, 'the default mode must be fast_fixed');
    // (through the addon with room to spare: processChunk's capacity rule DROPS input when chunk sizes vary -- F5 --
    //  and four chunks would then not be the stream of the one)
    const one = addon.process(whole._resamplerPtr, x, 200000, 60000);
    const parts = [];
    for (const [a, b] of [[0, 4800], [4800, 4900], [4900, 150000], [150000, 200000]]) {
      parts.push(addon.process(cut._resamplerPtr, x.slice(a * 4, b * 4), b - a, 60000));
    }
    assert(Buffer.concat(parts).length === one.length, "default mode: four chunks, another length");
    assert(Buffer.concat(parts).equals(one), "default mode: four chunks differ from one");
    whole.destroy(); cut.destroy();
  }
  console.log(`modes: fast ${off.fast} / fast_f32 ${off.fast_f32} / fast_fixed ${off.fast_fixed} of ${outs.exact.length / 2} samples off by one`);
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

// Round 5: many streams through ONE native call -- SpeexResamplerBatch.processChunks and the coalescer of
// processChunkAsync calls issued in one tick -- against the same streams as separate SpeexResampler instances
// (the reference's model, src/index.ts:18-45): byte-identical in 'exact' mode, +-1 LSB in the default mode,
// counters and positions equal, including 160-frame (640-byte) chunks where the capacity rule drops input (F5),
// ragged lengths, empty chunks and streams that sit a step out.
async function batchTest() {
  const addon = require('../speex_hip_napi.node');
  const N = 32;
  const close = (got, want, what) => {
    assert(got.length === want.length, `${what}: ${got.length} bytes, want ${want.length}`);
    for (let i = 0; i < want.length; i += 2) {
      assert(Math.abs(got.readInt16LE(i) - want.readInt16LE(i)) <= 1, `${what}: sample ${i / 2} off by more than 1 LSB`);
    }
  };
  for (const mode of ['exact', 'fast']) {
    const batch = new SpeexResamplerBatch(N, 2, 44100, 48000, 7);
    const apart = [];
    for (let k = 0; k < N; k++) apart.push(new SpeexResampler(2, 44100, 48000, 7));
    batch.processChunks(new Array(N).fill(Buffer.alloc(0)));   // lazy init, then the mode
    batch.setMode(mode);
    for (const r of apart) { r.processChunk(Buffer.alloc(0)); r.setMode(mode); }
    // frames per step; k = stream: 640-byte chunks (160 stereo frames) after a long one are the F5 case
    const steps = [(k) => 4410 + k, () => 160, () => 160, (k) => (k % 3 === 0 ? null : 7), () => 0, () => 16384,
      (k) => 160 + (k % 2), (k) => 100000 + 13 * k, () => 160, (k) => (k === 5 ? null : 2000)];
    steps.forEach((frames, step) => {
      const chunks = [];
      for (let k = 0; k < N; k++) {
        const f = frames(k);
        chunks.push(f === null ? null : lcg(f, 2, 1000 * step + k));
      }
      const got = batch.processChunks(chunks);
      for (let k = 0; k < N; k++) {
        if (chunks[k] === null) { assert(got[k] === null, 'a stream that sat out returns null'); continue; }
        const want = apart[k].processChunk(chunks[k]);
        if (mode === 'exact') assert(got[k].equals(want), `batch step ${step} stream ${k}: bytes differ (exact)`);
        else close(got[k], want, `batch step ${step} stream ${k}`);
        const a = addon.getInfo(batch.streams[k]._resamplerPtr), b = addon.getInfo(apart[k]._resamplerPtr);
        assert(a.last_sample === b.last_sample && a.samp_frac_num === b.samp_frac_num, `batch step ${step} stream ${k}: position`);
      }
    });
    // the asynchronous form, two steps queued at once: chained, same bytes
    const c1 = [], c2 = [];
    for (let k = 0; k < N; k++) { c1.push(lcg(3000 + k, 2, 70000 + k)); c2.push(lcg(160, 2, 80000 + k)); }
    const [o1, o2] = await Promise.all([batch.processChunksAsync(c1), batch.processChunksAsync(c2)]);
    for (let k = 0; k < N; k++) {
      const w1 = apart[k].processChunk(c1[k]), w2 = apart[k].processChunk(c2[k]);
      if (mode === 'exact') assert(o1[k].equals(w1) && o2[k].equals(w2), `async batch stream ${k}: bytes differ`);
      else { close(o1[k], w1, `async batch stream ${k}`); close(o2[k], w2, `async batch stream ${k} (2)`); }
    }
    batch.destroy();
    for (const r of apart) r.destroy();
  }
  // the coalescer: instances of THREE different filters call processChunkAsync in the same tick -- one native call,
  // one launch per filter -- and again with a second call per instance queued behind the first
  {
    const kinds = [[2, 44100, 48000, 7], [1, 24000, 48000, 10], [2, 48000, 44100, 5]];
    const mk = () => kinds.map((a) => [0, 1, 2, 3, 4].map(() => {
      const r = new SpeexResampler(a[0], a[1], a[2], a[3]);
      r.processChunk(Buffer.alloc(0));
      r.setMode('exact');
      return r;
    })).reduce((x, y) => x.concat(y), []);
    const fused = mk(), apart = mk();
    const before = addon.stats ? addon.stats() : null;
    const jobs = [], want = [];
    fused.forEach((r, i) => {
      const a = lcg(2000 + 17 * i, r.channels, 500 + i), b = lcg(160, r.channels, 600 + i);
      jobs.push(r.processChunkAsync(a), r.processChunkAsync(b));
      want.push(apart[i].processChunk(a), apart[i].processChunk(b));
    });
    const got = await Promise.all(jobs);
    got.forEach((g, i) => assert(g.equals(want[i]), `coalesced processChunkAsync ${i}: bytes differ`));
    fused.concat(apart).forEach((r) => r.destroy());
    void before;
  }
  // the coalescer under load: 120 instances of four filters, five ticks of ragged chunks each queued at once (a second
  // call per instance behind the first in some ticks) against the same streams through synchronous separate calls
  {
    const kinds = [[2, 44100, 48000, 7], [1, 48000, 16000, 5], [2, 48000, 44100, 10], [4, 44100, 48000, 3]];
    const mk = () => {
      const v = [];
      for (let n = 0; n < 120; n++) {
        const a = kinds[n % kinds.length];
        const r = new SpeexResampler(a[0], a[1], a[2], a[3]);
        r.setMode('exact');
        v.push(r);
      }
      return v;
    };
    const fused = mk(), apart = mk();
    for (let tick = 0; tick < 5; tick++) {
      const jobs = [], want = [];
      fused.forEach((r, n) => {
        const frames = [480, 160, 4096, 1, 960][(n + tick) % 5] + n;
        const a = lcg(frames, r.channels, 7000 * tick + n);
        jobs.push(r.processChunkAsync(a));
        want.push(apart[n].processChunk(a));
        if ((n + tick) % 7 === 0) {
          const b = lcg(160, r.channels, 9000 * tick + n);
          jobs.push(r.processChunkAsync(b));
          want.push(apart[n].processChunk(b));
        }
      });
      const got = await Promise.all(jobs);
      got.forEach((g, k) => assert(g.equals(want[k]), `coalescer under load: tick ${tick} job ${k} differs`));
    }
    fused.concat(apart).forEach((r) => r.destroy());
  }
  // misuse through the addon itself: a state twice in one call, a destroyed state, lengths that do not fit
  {
    const r = new SpeexResampler(2, 44100, 48000, 7);
    r.processChunk(Buffer.alloc(0));
    const h = r._resamplerPtr, c = lcg(100, 2, 1);
    const msg = (f) => { try { f(); return null; } catch (e) { return e.message; } };
    assert(/once per call/.test(msg(() => addon.processMany([h, h], [c, c], [100, 100], [200, 200]))), 'a state twice');
    assert(/exceeds the chunk/.test(msg(() => addon.processMany([h], [c], [101], [200]))), 'frames beyond the chunk');
    assert(addon.processMany([], [], [], []).length === 0, 'empty call');
    r.destroy();
    assert(msg(() => addon.processMany([h], [c], [100], [200])) === 'Bad resampler state.', 'destroyed state in a many-call');
  }
  // placement: an explicit device, a device the node does not have
  {
    const n = SpeexResampler.deviceCount();
    assert(n >= 1, 'deviceCount');
    const r = new SpeexResampler(2, 44100, 48000, 7, { device: n - 1 });
    assert(r.device === n - 1 && r.processChunk(lcg(1000, 2, 3)).length > 0 && r.device === n - 1, 'explicit device');
    r.destroy();
    let m = null;
    try { new SpeexResampler(2, 44100, 48000, 7, { device: n }).processChunk(lcg(10, 2, 3)); } catch (e) { m = e.message; }
    assert(m !== null && /device/.test(m), `a device the node does not have must fail loudly (${m})`);
    if (process.env.SPEEXHIP_DEVICES === 'all' && n > 1) {
      const b = new SpeexResamplerBatch(2 * n, 2, 44100, 48000, 7);
      b.processChunks(new Array(2 * n).fill(Buffer.alloc(0)));
      const seen = b.streams.map((s) => s.device);
      // (round 6: a new state goes to the GPU with the fewest LIVE states -- instances of the tests above that the
      //  collector has not finalized yet still count, so the deal need not start at 0; every GPU must get its share)
      for (let d = 0; d < n; d++) assert(seen.includes(d), `SPEEXHIP_DEVICES=all: devices ${seen}`);
      b.destroy();
      console.log(`placement: ${2 * n} streams on devices ${seen.join(',')}`);
    }
  }
  console.log('batch / coalescer / placement: many streams per native call, bytes of the separate instances');
}

// The addon under worker_threads: every Worker loads it (NAPI_MODULE_INIT), has a state of its own and gets the
// reference's bytes ('exact' mode, sha1 of the main thread's result).
async function workerTest() {
  let wt;
  try { wt = require('worker_threads'); } catch (e) { console.log('worker_threads not available: skipped'); return; }
  const pcm = lcg(50000, 2, 4321);
  const main = new SpeexResampler(2, 44100, 48000, 7);
  main.setMode('exact');
  const want = sha1(main.processChunk(pcm));
  main.destroy();
  const src = `
    const { parentPort, workerData } = require('worker_threads');
    const crypto = require('crypto');
    const R = require(workerData.index).default;
    R.initPromise.then(() => {
      const out = [];
      for (let k = 0; k < 3; k++) {
        const r = new R(2, 44100, 48000, 7);
        r.setMode('exact');
        out.push(crypto.createHash('sha1').update(r.processChunk(Buffer.from(workerData.pcm))).digest('hex'));
        if (k % 2) r.destroy();
      }
      parentPort.postMessage(out);
    }).catch((e) => parentPort.postMessage(['error: ' + e.message]));`;
  const runs = [0, 1].map(() => new Promise((res, rej) => {
    const w = new wt.Worker(src, { eval: true, workerData: { index: path.join(__dirname, '../index.js'), pcm } });
    w.on('message', res);
    w.on('error', rej);
  }));
  for (const hashes of await Promise.all(runs)) {
    assert(hashes.length === 3 && hashes.every((h) => h === want), `a Worker's bytes differ: ${hashes}`);
  }
  console.log('worker_threads: two Workers, states of their own, the main thread\'s bytes');
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
  await batchTest();
  await goldenBatchTest();
  await workerTest();
  console.log('ALL NODE TESTS PASSED');
})().catch((e) => { console.error(e); process.exit(1); });
