#!/usr/bin/env python3
"""f32x (split mode with fp8 cross terms) against f32s and f32: logit error vs the torch-CPU f32 oracle, frames/s (run on an MI355X)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import processors as P
from infur_amd import weights as W
from oracle.infur_oracle import COracle, TorchModel

blob = W.synth_blob()
co, tm = COracle(), TorchModel(blob)
for (w, h) in ((64, 48), (320, 240), (960, 540)):
    fr = W.synth_frame(h, w, index=3)
    tl, ta = tm.forward_lowres(co.pack_normalize(fr))
    for dt in ("f32", "f32s", "f32x", "f16"):
        c = P.Context(device=0, dtype=dt)
        m = P.Model(c).control(P.ModelCmd.LoadBlob(blob))
        P.FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        e = max(np.abs(lo - tl.numpy()).max() / np.abs(tl.numpy()).max(), np.abs(la - ta.numpy()).max() / np.abs(ta.numpy()).max())
        print(f"{w}x{h} {dt:5s} logits rel err {e:.2e}", flush=True)
        c.close()
