#!/usr/bin/env python3
"""Read the per-phase cycle sums a -DKTRACE build of conv_igemm.hip records for the layer4 downsample conv
(K loop of the 1frag form): run one 1080p frame with every layer forced to tile configuration CFG."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["INFUR_CONV_CFG"] = sys.argv[1] if len(sys.argv) > 1 else "15"
from infur_amd import _lib  # noqa: E402
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

c = P.Context(device=0, dtype="f32s")
P.Model(c).control(P.ModelCmd.LoadBlob(W.synth_blob(depth=50)))
fp = P.FramePath(c)
fr = W.synth_frame(1080, 1920)
for _ in range(3):
    fp.advance(fr, 1.0)
L = _lib.load()
buf = np.zeros(8 * 8 * 8, np.uint64)
L.infur_debug_ktrace.restype = C.c_int32
assert L.infur_debug_ktrace(C.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(8, 8, 8).astype(np.float64)
names = ["ds_read+wait", "mfma issue", "store(vmcnt wait+cvt+ds_write)", "global load issue", "barrier", "", "", ""]
print("cycles per K loop (32 K steps), mean over blocks 0-7, per wave:")
for q in range(5):
    print(f"  {names[q]:34s}", " ".join(f"{t[:, w, q].mean():9.0f}" for w in range(8)))
tot = t[:, :, :5].sum(axis=2)
print("  total".ljust(36), " ".join(f"{tot[:, w].mean():9.0f}" for w in range(8)))
for q, nm in ((5, "prologue (start -> K loop)"), (6, "K loop"), (7, "epilogue (incl. vmcnt(0))")):
    print(f"  {nm:34s}", " ".join(f"{t[:, w, q].mean():9.0f}" for w in range(8)))
