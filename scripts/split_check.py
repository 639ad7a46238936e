#!/usr/bin/env python3
"""f32 vs f32s (split) on one frame: speed of both modes and the logit difference (run on an MI355X)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
blob = W.synth_blob(depth=50)
fr = W.synth_frame(h, w, index=1)
lows = {}
for dt in ("f32", "f32s"):
    c = P.Context(device=0, dtype=dt)
    m = P.Model(c).control(P.ModelCmd.LoadBlob(blob))
    fp = P.FramePath(c)
    fp.advance(fr, 1.0)
    fp.advance(fr, 1.0)
    t0 = time.perf_counter()
    for _ in range(5):
        fp.advance(fr, 1.0)
    dt_ms = (time.perf_counter() - t0) / 5 * 1e3
    lows[dt] = m.lowres()[0].astype(np.float64)
    print(f"{dt}: {dt_ms:.2f} ms/frame from host buffers", flush=True)
    c.close()
e = np.abs(lows["f32s"] - lows["f32"]).max() / np.abs(lows["f32"]).max()
print(f"f32s vs f32 logits: max rel diff {e:.3e}")
