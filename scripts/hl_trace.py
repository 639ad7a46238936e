#!/usr/bin/env python3
"""Per-phase shader cycles of conv_hl's pipelined K loop for one layer shape (instrumentation build).

    make -C infur_amd/csrc EXTRA="-DHL_TRACE -DHLT_CIN=2048 -DHLT_COUT=512" OUT=../libinfur_hip_trace.so build/conv_hl.o ... (see scripts/hl_trace.sh)
    INFUR_LIB=infur_amd/libinfur_hip_trace.so python scripts/hl_trace.py            # on an MI355X
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import _lib  # noqa: E402
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

c = P.Context(device=0, dtype="f16hl")
P.Model(c).control(P.ModelCmd.LoadBlob(W.synth_blob(depth=50)))
fp = P.FramePath(c)
fr = W.synth_frame(1080, 1920)
for _ in range(3):
    fp.advance(fr, 1.0)
L = _lib.load()
raw = C.CDLL(L._name)
buf = np.zeros(64, np.uint64)
raw.infur_debug_hltrace.restype = C.c_int
assert raw.infur_debug_hltrace(C.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(8, 8).astype(np.float64)
names = ["prologue+half0", "at barrier", "bf8 half (reads, DMA issue, 8 MFMAs)", "f16 half (12 reads, 16 MFMAs)", "vmcnt/lgkmcnt wait", "last half", "epilogue", "-"]
print("shader cycles per wave (sums over the K loop):")
for k, nm in enumerate(names[:7]):
    print(f"  {nm:40s}", " ".join(f"{t[w, k]:9.0f}" for w in range(8)))
print(f"  {'total':40s}", " ".join(f"{t[w, :7].sum():9.0f}" for w in range(8)))
