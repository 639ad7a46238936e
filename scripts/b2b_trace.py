#!/usr/bin/env python3
"""Per-phase shader cycles of conv1x1_b2b.hip (instrumentation build), one workgroup of the last fused launch:

    scripts/b2b_trace.sh            # builds infur_amd/libinfur_hip_trace.so (-DB2B_TRACE) and runs this on an MI355X

Prints, per wave, the cycles spent over all 32 steps in: waiting for its DMA pieces, the barrier, fragment preload + DMA
issue, GEMM 1, epilogue 1 (bias + residual + ReLU + f16, y into LDS), the y stores, GEMM 2."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libinfur_hip_trace.so")
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

os.environ.setdefault("INFUR_B2B", "1")
c = P.Context(device=0, dtype="f16")
P.Model(c).control(P.ModelCmd.LoadBlob(W.synth_blob(depth=50)))
fp = P.FramePath(c)
fr = W.synth_frame(2160, 3840)
for _ in range(3):
    fp.advance(fr, 1.0)
L = _lib.load()
buf = np.zeros(64, np.uint64)
L.infur_debug_b2b_trace.restype = C.c_int32
assert L.infur_debug_b2b_trace(C.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(8, 8).astype(np.float64)
names = ["dma wait", "barrier", "preload+dma", "gemm1", "epilogue1", "y stores", "gemm2", "total loop"]
print("shader cycles per wave, summed over the steps of one workgroup (form %s):" % os.environ.get("INFUR_B2B_FORM", "default"))
for k, nm in enumerate(names):
    print(f"  {nm:12s}", " ".join(f"{t[w, k]:9.0f}" for w in range(8)))
