// Golden-generation helper (dev container only): runs the REFERENCE's shipped WASM build
// (/root/reference/app/index.js) over a PCM file split into the given chunk byte sizes and
// prints {sha1, out_frames}.  Called by make_golden.py; nothing here ships or runs on the GPU box.
const fs = require('fs');
const crypto = require('crypto');
const spec = JSON.parse(process.argv[2]);
const Ref = require('/root/reference/app/index.js');
(async () => {
  await Ref.default.initPromise; // src/test.ts forgets this (SURVEY section 0)
  const data = fs.readFileSync(spec.file);
  const r = new Ref.default(spec.channels, spec.in_rate, spec.out_rate, spec.quality);
  const h = crypto.createHash('sha1');
  let off = 0, bytes = 0;
  for (const n of spec.chunks) {
    const out = r.processChunk(data.slice(off, off + n));
    h.update(out);
    bytes += out.length;
    off += n;
  }
  process.stdout.write(JSON.stringify({sha1: h.digest('hex'), out_frames: bytes / 2 / spec.channels}));
})().catch((e) => { console.error(e); process.exit(1); });
