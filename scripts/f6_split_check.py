#!/usr/bin/env python3
"""f32s with Winograd F(4x4) vs F(6x6): logit error against the torch-CPU f32 oracle, range monitor (run on an MI355X)."""
import os, sys
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
    for dt, tile in (("f32", 4), ("f32", 6), ("f32s", 4), ("f32s", 6)):
        c = P.Context(device=0, dtype=dt, winograd_tile=tile)
        m = P.Model(c).control(P.ModelCmd.LoadBlob(blob))
        P.FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        e = max(np.abs(lo - tl.numpy()).max() / np.abs(tl.numpy()).max(), np.abs(la - ta.numpy()).max() / np.abs(ta.numpy()).max())
        extra = ""
        if dt == "f32s":
            extra = " range (act, wino, saturated) = %s" % (c.split_range(),)
        print(f"{w}x{h} {dt} F{tile}: logits rel err {e:.2e}{extra}", flush=True)
        c.close()
