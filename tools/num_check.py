import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, speexhip, oracle as orc
ch, i, o, q, frames = 2, 44100, 48000, 7, 1 << 20
for name, x in (("lcg white noise", orc.lcg_pcm(frames * ch, 12345).reshape(frames, ch)), ("tone (music-like)", orc.tone_pcm(frames, ch, seed=7))):
    want, _ = orc.Oracle(ch, i, o, q).process(x, 1 << 21)
    r = speexhip.Resampler(ch, i, o, q); got, _ = r.process(x, 1 << 21); r.close()
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    # pre-rounding error: the float entry point on the same (integer-valued) samples
    xf = x.astype(np.float32)
    wf, _ = orc.Oracle(ch, i, o, q).process_float(xf, 1 << 21)
    r = speexhip.Resampler(ch, i, o, q); gf, _ = r.process_float(xf, 1 << 21); r.close()
    e = np.abs(gf.astype(np.float64) - wf.astype(np.float64))
    print("%-18s int16: max |diff| %d LSB, mismatch rate %.3e | float (pre-rounding): max |err| %.3e LSB, mean %.3e LSB" % (name, d.max(), (d != 0).mean(), e.max(), e.mean()))
