"""Helpers shared by the golden-vector tests (input regeneration + call-sequence driver)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle as orc  # noqa: E402
from make_golden import make_input, chunk_plan  # noqa: E402,F401  (pure-python helpers)


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def drive(engine, spec, x):
    """Run `engine` (anything with .process(frames, capacity) and .position()) through the
    processChunk call sequence of a golden case, with the JS wrapper's capacity rule
    (reference src/index.ts:80-87,95).  Returns (out, per_call rows)."""
    ch = spec["channels"]
    outs, per_call, out_buf_size, off = [], [], -1, 0
    for nbytes in chunk_plan(spec, x.size * 2):
        f = nbytes // (2 * ch)
        cap, out_buf_size = orc.wrapper_capacity(nbytes, spec["in_rate"], spec["out_rate"], ch,
                                                 out_buf_size)
        o, used = engine.process(x[off: off + f], cap)
        pos, ph = engine.position()
        per_call.append([f, cap, used, int(o.shape[0]), int(pos), int(ph)])
        outs.append(o)
        off += f
    out = np.concatenate(outs) if outs else np.zeros((0, ch), np.int16)
    return out, per_call
