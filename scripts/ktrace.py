#!/usr/bin/env python3
"""Per-phase shader cycles of the conv kernel for one layer shape (instrumentation build).

    make -C infur_amd/csrc clean && make -C infur_amd/csrc EXTRA="-DKTRACE -DKT_CIN=512 -DKT_COUT=2048"
    python scripts/ktrace.py [dtype] [forced tile configuration | -] [depth] [height] [width]        # on an MI355X
    make -C infur_amd/csrc clean && make -C infur_amd/csrc              # back to the product build

Prints, for the first 8 workgroups of the last launch with Cin == KT_CIN and Cout == KT_COUT, the cycles each
wave spent in the prologue (first loads -> first barrier), the K loop and the epilogue (to the last store's
completion)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
if len(sys.argv) > 2 and sys.argv[2] != "-":
    os.environ["INFUR_CONV_CFG"] = sys.argv[2]
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
fh, fw = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1080, 1920)
from infur_amd import _lib  # noqa: E402
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

c = P.Context(device=0, dtype="f32" if dtype == "i8" else dtype)
if dtype == "i8":  # the quantised model (its own arithmetic: the context's dtype does not matter)
    from infur_amd import quantize

    P.Model(c).control(P.ModelCmd.LoadBlob(quantize.synth_qblob(depth=depth)))
else:
    P.Model(c).control(P.ModelCmd.LoadBlob(W.synth_blob(depth=depth)))
fp = P.FramePath(c)
fr = W.synth_frame(fh, fw)
for _ in range(3):
    fp.advance(fr, 1.0)
L = _lib.load()
if not hasattr(L, "infur_debug_ktrace"):
    sys.exit("libinfur_hip.so was not built with -DKTRACE")
buf = np.zeros(8 * 8 * 4, np.uint64)
L.infur_debug_ktrace.restype = C.c_int32
assert L.infur_debug_ktrace(C.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(8, 8, 4).astype(np.float64)
print("shader cycles, mean over workgroups 0-7, per wave:")
for q, nm in ((0, "prologue"), (1, "K loop"), (2, "epilogue")):
    print(f"  {nm:10s}", " ".join(f"{t[:, w, q].mean():9.0f}" for w in range(8)))
