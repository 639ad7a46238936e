import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from infur_amd import processors as P
from infur_amd import weights as W
from oracle.infur_oracle import COracle, TorchModel
blob = W.synth_blob()
co, tm = COracle(), TorchModel(blob)
w, h = 960, 540
fr = W.synth_frame(h, w, index=3)
tl, ta = tm.forward_lowres(co.pack_normalize(fr))
fr2 = W.synth_frame(1080, 1920, index=1)
for tile, mincin in ((6, 128), (4, 128), (2, 128), (6, 0xFFFFFFFF)):
    c = P.Context(device=0, dtype="f32x", winograd_tile=tile, winograd_min_cin=mincin)
    m = P.Model(c).control(P.ModelCmd.LoadBlob(blob))
    fp = P.FramePath(c)
    fp.advance(fr, 1.0)
    lo, la = m.lowres()
    e = max(np.abs(lo - tl.numpy()).max() / np.abs(tl.numpy()).max(), np.abs(la - ta.numpy()).max() / np.abs(ta.numpy()).max())
    for _ in range(3): fp.advance(fr2, 1.0)
    t0 = time.perf_counter()
    for _ in range(10): fp.advance(fr2, 1.0)
    dt = (time.perf_counter() - t0) / 10
    print(f"f32x winograd tile {tile} min_cin {mincin}: logits rel err {e:.2e}; 1080p host-path {1/dt:.1f} frames/s", flush=True)
    c.close()
