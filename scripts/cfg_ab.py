#!/usr/bin/env python3
"""Same-box A/B of conv configurations: per-layer HIP-event times of whole frames (median over frames), grouped by stage and
role, with the kernel that ran.  The variant comes from the environment (read once per process), e.g.

    for k in 16 18; do INFUR_CONV_CFG=$k python scripts/cfg_ab.py [h w [depth [dtype]]]; done"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 101
dtype = sys.argv[4] if len(sys.argv) > 4 else "f16"
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # Winograd output tile (0 = default)
c = None if dtype == "i8" else P.Context(device=0, dtype=dtype, profile=True, winograd_tile=tile)
if dtype == "i8":
    from infur_amd import quantize
    c = P.Context(device=0, profile=True)
    P.Model(c).control(P.ModelCmd.LoadBlob(quantize.synth_qblob(depth=depth)))
else:
    P.Model(c).control(P.ModelCmd.LoadBlob(W.synth_blob(depth=depth)))
fp = P.FramePath(c)
fr = W.synth_frame(h, w)
acc, tot = {}, []
for it in range(7):
    fp.advance(fr, 1.0)
    if it < 2:
        continue
    rows = c.profile()
    tot.append(sum(r["ms"] for r in rows))
    for r in rows:
        n = r["name"]
        parts = n.split(".")
        grp = parts[1] if n.startswith("backbone") and len(parts) > 2 else n
        role = parts[-1] if n.startswith("backbone.layer") else ""
        if role == "0" and "downsample" in n:
            role = "downsample"
        blk = "first" if n.startswith("backbone.layer") and parts[2] == "0" else "rest"
        acc.setdefault(f"{grp} {role} {blk} {r['kernel']}", []).append(r["ms"])
env = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("INFUR_"))
print(f"[{env}] {w}x{h} r{depth} {dtype} tile {tile}: frame kernels {np.median(tot):.3f} ms")
for k in sorted(acc):
    print(f"   {k:90s} median {np.median(acc[k]) * 1e3:7.1f} us  (n={len(acc[k])})")
