#!/usr/bin/env python3
"""Two frames in flight: do the two contexts' streams overlap?  Rate of K = 2 contexts whose streams are (a) created by
infur_ctx_create itself, back to back, (b) created by it with other streams created in between, (c) taken from torch's pool
(what bench.py's headline does) -- against one context.   python scripts/ctx_streams.py [dtype]   (i8 = the quantised model)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

dt = sys.argv[1] if len(sys.argv) > 1 else "i8"
H, Wd = 1080, 1920
if dt == "i8":
    from infur_amd import quantize
    blob = quantize.synth_qblob(depth=50)
else:
    blob = W.synth_blob()
frames = [torch.from_numpy(W.synth_frame(H, Wd, index=i)).cuda() for i in range(8)]
masks = [torch.empty((H, Wd, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]


def rate(ctxs, label):
    K = len(ctxs)
    for c in ctxs:
        Model(c).control(ModelCmd.LoadBlob(blob))
    fps = [FramePath(c) for c in ctxs]

    def run(n):
        for i in range(n):
            fps[i % K].advance_dev(frames[i % 8].data_ptr(), Wd, H, 1.0, masks[i % 8].data_ptr(), masks[i % 8].numel())
        for c in ctxs:
            c.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        run(8 * K)
    t0 = time.perf_counter(); run(192); s = time.perf_counter() - t0
    print(f"{dt} {label}: {192 / s:.1f} frames/s", flush=True)
    for c in ctxs:
        c.close()


cdt = "f32" if dt == "i8" else dt
rate([Context(device=0, dtype=cdt)], "one context")
rate([Context(device=0, dtype=cdt) for _ in range(2)], "two contexts, own streams (created back to back)")
a = Context(device=0, dtype=cdt); junk = [torch.cuda.Stream() for _ in range(1)]; b = Context(device=0, dtype=cdt)
rate([a, b], "two contexts, own streams (one other stream created in between)")
st = [torch.cuda.Stream() for _ in range(2)]
rate([Context(device=0, dtype=cdt, stream=s.cuda_stream) for s in st], "two contexts, torch pool streams")
st = [torch.cuda.Stream(priority=-1) for _ in range(2)]
rate([Context(device=0, dtype=cdt, stream=s.cuda_stream) for s in st], "two contexts, torch high-priority streams")
rate([Context(device=0, dtype=cdt) for _ in range(2)], "two contexts, own streams again")
