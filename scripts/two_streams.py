#!/usr/bin/env python3
"""Throughput with K contexts (K streams) on ONE GPU working on different frames at the same time vs one context.
   python scripts/two_streams.py [dtype] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
H, Wd = 1080, 1920
blob = W.synth_blob()
frames = [torch.from_numpy(W.synth_frame(H, Wd, index=i)).cuda() for i in range(8)]
for K in (1, 2, 3):
    ctxs = [Context(device=0, dtype=dt) for _ in range(K)]
    for c in ctxs:
        Model(c).control(ModelCmd.LoadBlob(blob))
    fps = [FramePath(c) for c in ctxs]
    masks = [torch.empty((H, Wd, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]
    def run(n):
        for i in range(n):
            k = i % K
            fps[k].advance_dev(frames[i % 8].data_ptr(), Wd, H, 1.0, masks[i % 8].data_ptr(), masks[i % 8].numel())
        for c in ctxs:
            c.synchronize()
    run(8 * K)
    t0 = time.perf_counter(); run(96); dt_s = time.perf_counter() - t0
    print(f"{dt} K={K}: {96 / dt_s:.1f} frames/s", flush=True)
    for c in ctxs:
        c.close()
