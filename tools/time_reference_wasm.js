// Times the REFERENCE package (its shipped WASM build, /root/reference/app) on this container's CPU on the
// seven (file, rates, channels, quality) tuples of its own test (src/test.ts:14-22), the way that test
// does: one processChunk over the whole file.  Build-container only (the reference does not travel to the
// GPU box); writes profiles/r02_reference_wasm_cpu.json, which test/bench.js's numbers are read beside.
//   node tools/time_reference_wasm.js [/root/reference]
'use strict';
const fs = require('fs');
const os = require('os');
const path = require('path');
const { performance } = require('perf_hooks');

const ref = process.argv[2] || '/root/reference';
const SpeexResampler = require(path.join(ref, 'app', 'index.js')).default;
const tuples = [
  ['24000hz_mono_test.pcm', 24000, 48000, 1, 5],
  ['24000hz_test.pcm', 24000, 24000, 2, 5],
  ['24000hz_test.pcm', 24000, 48000, 2, 10],
  ['44100hz_test.pcm', 44100, 48000, 2, 7],
  ['44100hz_test.pcm', 44100, 48000, 2, 10],
  ['44100hz_test.pcm', 44100, 48000, 2, 1],
  ['44100hz_test.pcm', 44100, 24000, 2, 5],
];

(async () => {
  await SpeexResampler.initPromise;
  const rows = [];
  for (const [file, inRate, outRate, ch, q] of tuples) {
    const pcm = fs.readFileSync(path.join(ref, 'resources', file));
    const times = [];
    for (let rep = 0; rep < 5; rep++) {
      const r = new SpeexResampler(ch, inRate, outRate, q);
      const t0 = performance.now();
      const out = r.processChunk(pcm);
      times.push(performance.now() - t0);
      if (rep === 0) rows.push({ file, inRate, outRate, channels: ch, quality: q, in_bytes: pcm.length, out_bytes: out.length });
    }
    times.sort((a, b) => a - b);
    const row = rows[rows.length - 1];
    row.ms_min = +times[0].toFixed(3);
    row.ms_median = +times[2].toFixed(3);
    row.input_msamples_per_s = +(pcm.length / 2 / times[2] / 1e3).toFixed(2);
    console.log(JSON.stringify(row));
  }
  const out = {
    what: 'reference package (WASM, app/index.js) processChunk over the whole file, 5 runs each, one thread',
    where: 'build container CPU: ' + os.cpus()[0].model + ' (' + os.cpus().length + ' logical CPUs), node ' + process.version,
    rows,
  };
  fs.writeFileSync(path.join(__dirname, '..', 'profiles', 'r02_reference_wasm_cpu.json'), JSON.stringify(out, null, 1) + '\n');
})().catch((e) => { console.error(e); process.exit(1); });
