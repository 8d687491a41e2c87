import os, sys
import numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(R, "node-speex-resampler_amd", "python")); sys.path.insert(0, os.path.join(R, "oracle"))
import speexhip, oracle as orc
for ch, i, o, q in [(8, 80000, 8000, 10), (8, 96000, 8000, 10), (8, 72000, 8000, 10), (6, 96000, 8000, 10), (8, 96000, 8000, 8)]:
    x = (np.random.RandomState(1).randn(60000, ch) * 3000).astype(np.int16)
    r = speexhip.Resampler(ch, i, o, q)
    try:
        got, used = r.process(x, 60000)
        want, wu = orc.Oracle(ch, i, o, q).process(x, 60000)
        print(ch, i, o, q, 'fast_path', r.info()['fast_path'], 'ok', used == wu, int(np.abs(got.astype(np.int32) - want).max()))
    except Exception as e:
        print(ch, i, o, q, 'fast_path', r.info()['fast_path'], 'FAILED', repr(e)[:200])
