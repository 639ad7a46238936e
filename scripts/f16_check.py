"""f16 mode: error vs the f32 oracle and speed (dev script)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd
from oracle.infur_oracle import COracle, TorchModel
for depth, (h, w) in ((50, (96, 128)), (50, (540, 960)), (101, (270, 480))):
    blob = W.synth_blob(depth=depth)
    tm = TorchModel(blob); co = COracle()
    fr = W.synth_frame(h, w, index=1)
    tl, ta = tm.forward_lowres(co.pack_normalize(fr))
    for dt in ("f32", "f16"):
        c = Context(device=0, dtype=dt)
        m = Model(c).control(ModelCmd.LoadBlob(blob))
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        e = np.abs(lo - tl.numpy()).max() / np.abs(tl.numpy()).max()
        ea = np.abs(la - ta.numpy()).max() / np.abs(ta.numpy()).max()
        ref = co.colorcode(co.upsample_bilinear(tl.numpy(), h, w))
        kl_ref, _ = co.argmax(co.upsample_bilinear(tl.numpy(), h, w))
        kl, _ = co.argmax(co.upsample_bilinear(lo, h, w))
        print(f"R{depth} {w}x{h} {dt}: rel err out {e:.2e} aux {ea:.2e}; class map mismatch {(kl != kl_ref).mean():.4%}; mask bytes differ {(rgba != ref).any(-1).mean():.4%}", flush=True)
        c.close()
