#!/usr/bin/env python3
"""Per-layer HIP-event times of one frame (one context) in a given mode / depth / size:
    python scripts/layer_times.py f16 50 1920 1080 [substring ...]      (INFUR_LIB_PATH / INFUR_CONV_CFG select builds / forms)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infur_amd import weights as W  # noqa: E402
from infur_amd.processors import Context, FramePath, Model, ModelCmd  # noqa: E402

dtype, depth, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
subs = sys.argv[5:]
if dtype == "i8":
    from infur_amd import quantize
    blob = quantize.synth_qblob(depth=depth)
    c = Context(device=0, dtype="f32", profile=True)
else:
    blob = W.synth_blob(depth=depth)
    c = Context(device=0, dtype=dtype, profile=True)
Model(c).control(ModelCmd.LoadBlob(blob))
fp = FramePath(c, 0)
fr = W.synth_frame(h, w, index=1)
for _ in range(4):
    fp.advance(fr, 1.0)
recs = c.profile()
print(f"== {dtype} R{depth} {w}x{h}: {sum(r['ms'] for r in recs):.3f} ms of kernels")
by = {}
for r in recs:
    by.setdefault(r["kernel"], [0, 0.0])
    by[r["kernel"]][0] += 1
    by[r["kernel"]][1] += r["ms"]
for k, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:44s} x{n:3d} {ms * 1e3:9.1f} us")
for r in recs:
    if subs and any(s in r["name"] or s in r["kernel"] for s in subs):
        print(f"      {r['name']:44s} {r['kernel']:32s} {r['ms'] * 1e3:8.1f} us")
c.close()
