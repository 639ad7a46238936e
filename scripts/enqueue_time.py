"""How long the host needs to enqueue one frame (82 launches) vs how long the GPU needs to run it: if enqueue << GPU
time the inter-kernel gaps of the one-context form are GPU-side (dependent-kernel barriers), not launch-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

for dtype in ("f32", "f32s", "f16"):
    c = Context(device=0, dtype=dtype, profile=False)
    Model(c).control(ModelCmd.LoadBlob(W.synth_blob()))
    fr = torch.from_numpy(W.synth_frame(1080, 1920)).cuda()
    out = torch.empty((1080, 1920, 4), dtype=torch.uint8, device="cuda")
    fp = FramePath(c)
    for _ in range(6):
        fp.advance_dev(fr.data_ptr(), 1920, 1080, 1.0, out.data_ptr(), out.numel())
    c.synchronize()
    n = 20
    t0 = time.perf_counter(); enq = 0.0
    for _ in range(n):
        a = time.perf_counter()
        fp.advance_dev(fr.data_ptr(), 1920, 1080, 1.0, out.data_ptr(), out.numel())
        enq += time.perf_counter() - a
    c.synchronize()
    tot = time.perf_counter() - t0
    print(f"{dtype}: enqueue {enq / n * 1e3:.3f} ms/frame, GPU {tot / n * 1e3:.3f} ms/frame")
    c.close()
