#!/usr/bin/env python3
"""frames/s of the fused frame path, eager launches against hipGraph replay (infur_ctx_set_graph_replay), one context, frames and
masks resident in HBM: where the host's enqueue time bounds the rate (small frames in the fast modes) and where it does not.
    python scripts/graph_rate.py        (on an MI355X; prints a markdown table)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from infur_amd import quantize, weights as W  # noqa: E402
from infur_amd.processors import Context, FramePath, Model, ModelCmd  # noqa: E402

blobs = {"q": quantize.synth_qblob(), "f": W.synth_blob()}


def rate(dtype, w, h, graph, n):
    c = Context(device=0, dtype="f32" if dtype == "i8" else dtype, graph_replay=graph)
    Model(c).control(ModelCmd.LoadBlob(blobs["q" if dtype == "i8" else "f"]))
    fp = FramePath(c)
    d_in = [torch.from_numpy(W.synth_frame(h, w, index=i)).cuda() for i in range(2)]
    d_out = [torch.empty((h, w, 4), dtype=torch.uint8, device="cuda") for _ in d_in]
    for i in range(24):  # settle, capture (one graph per buffer pair)
        fp.advance_dev(d_in[i % 2].data_ptr(), w, h, 1.0, d_out[i % 2].data_ptr(), d_out[i % 2].numel())
    c.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fp.advance_dev(d_in[i % 2].data_ptr(), w, h, 1.0, d_out[i % 2].data_ptr(), d_out[i % 2].numel())
    c.synchronize()
    dt = time.perf_counter() - t0
    st = c.graph_stats()
    c.close()
    return n / dt, st


print("| frame | mode | eager frames/s | graph replay frames/s | gain |\n|---|---|---|---|---|")
for w, h in ((320, 240), (640, 480), (960, 540), (1920, 1080)):
    for dtype in ("i8", "f16", "f32s", "f32"):
        n = max(40, int(2.0e8 / (w * h * (1 if dtype in ("i8", "f16") else 3))))
        e, _ = rate(dtype, w, h, False, n)
        g, st = rate(dtype, w, h, True, n)
        assert st[1] >= n, st
        print(f"| {w}x{h} | {dtype} | {e:.0f} | {g:.0f} | {g / e:.2f}x |", flush=True)
