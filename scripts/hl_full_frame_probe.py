import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import hostile as H
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd
from oracle.infur_oracle import COracle, TorchModel
co = COracle()
for tag, blob, fr in (("hostile", H.hostile_blob(), H.saturated_frame(1080, 1920, index=2)), ("synthetic", W.synth_blob(), W.synth_frame(1080, 1920, index=3))):
    ref, ref_aux = TorchModel(blob, float64=True).forward_lowres(co.pack_normalize(fr))
    ref, ref_aux = ref.numpy(), ref_aux.numpy()
    for tile in (6, 4):
        c = Context(device=0, dtype="f16hl", winograd_tile=tile)
        m = Model(c).control(ModelCmd.LoadBlob(blob))
        FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        print(tag, "1080p fused F", tile, H.errors(lo, ref), H.errors(la, ref_aux), flush=True)
        c.close()
