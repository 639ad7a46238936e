#!/usr/bin/env python3
"""frames/s and the per-kernel table of the quantised (int8) FCN-ResNet50 at 1080p (or W H given).  Run on an MI355X."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from infur_amd import quantize, weights as W  # noqa: E402
from infur_amd.processors import Context, FramePath, Model, ModelCmd  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
qblob = quantize.synth_qblob()
c = Context(device=0, profile=True)
Model(c).control(ModelCmd.LoadBlob(qblob))
fp = FramePath(c)
d_in = [torch.from_numpy(W.synth_frame(h, w, index=i)).cuda() for i in range(4)]
d_out = [torch.empty((h, w, 4), dtype=torch.uint8, device="cuda") for _ in d_in]
for i in range(8):
    fp.advance_dev(d_in[i % 4].data_ptr(), w, h, 1.0, d_out[i % 4].data_ptr(), d_out[i % 4].numel())
c.synchronize()
recs = c.profile()
c.L.infur_profile_enable(c.h, 0)
t0 = time.perf_counter()
n = 64
for i in range(n):
    fp.advance_dev(d_in[i % 4].data_ptr(), w, h, 1.0, d_out[i % 4].data_ptr(), d_out[i % 4].numel())
c.synchronize()
dt = time.perf_counter() - t0
print(f"int8 {w}x{h}: {n / dt:.1f} frames/s, {dt / n * 1e3:.3f} ms per frame (one context); kernel time of one frame {sum(r['ms'] for r in recs):.3f} ms")
for r in recs:
    print(f"{r['name']:40s} {r['kernel']:28s} {r['ms']:8.3f} ms {r['flops'] / max(r['ms'], 1e-9) / 1e9:8.1f} TOP/s {r['bytes'] / max(r['ms'], 1e-9) / 1e6:9.1f} GB/s")
