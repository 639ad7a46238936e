#!/usr/bin/env python3
"""A/B timing of the fused conv3 -> next conv1 launch (conv1x1_b2b.hip) against the two launches it replaces, on the
layer3 / layer2 shapes of a 4K frame (FCN-ResNet50, f16): per-kernel HIP events of whole frames, median over frames.
Variants are chosen by environment (read once per process): INFUR_B2B=0/1, INFUR_B2B_FORM, INFUR_B2B_DMA, INFUR_B2B_STAGGER.

    for f in 1 2; do INFUR_B2B_FORM=$f python scripts/b2b_ab.py; done"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
c = P.Context(device=0, dtype="f16", profile=True)
P.Model(c).control(P.ModelCmd.LoadBlob(W.synth_blob(depth=50)))
fp = P.FramePath(c)
fr = W.synth_frame(h, w)
acc = {}
for it in range(6):
    fp.advance(fr, 1.0)
    if it < 2:
        continue
    for r in c.profile():
        n = r["name"]
        if ("layer3" in n or "layer2" in n) and ("conv3" in n or "conv1" in n) and "layer3.0" not in n and "layer2.0" not in n:
            key = ("L3 " if "layer3" in n else "L2 ") + ("pair" if "+next" in n else n.split(".")[-1]) + " " + r["kernel"]
            acc.setdefault(key, []).append(r["ms"])
tag = " ".join(f"{k}={os.environ[k]}" for k in ("INFUR_B2B", "INFUR_B2B_FORM", "INFUR_B2B_DMA", "INFUR_B2B_STAGGER") if k in os.environ)
print(f"[{tag}] {w}x{h}")
for k in sorted(acc):
    print(f"   {k:60s} median {np.median(acc[k]) * 1e3:7.1f} us  (n={len(acc[k])})")
