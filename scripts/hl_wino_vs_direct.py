#!/usr/bin/env python3
"""f16hl at 1920x1080: per-layer HIP-event times of the 3x3 convolutions as Winograd F(6x6) / F(4x4) on planes and as direct conv_hl
launches (one context).  Run on an MI355X."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infur_amd import weights as W  # noqa: E402
from infur_amd.processors import Context, FramePath, Model, ModelCmd  # noqa: E402

blob = W.synth_blob()
fr = W.synth_frame(1080, 1920, index=1)
for tile, name in ((6, "F(6x6)"), (4, "F(4x4)"), (-1, "direct")):
    c = Context(device=0, dtype="f16hl", profile=True, winograd_tile=max(tile, 0), winograd_min_cin=0xFFFFFFFF if tile < 0 else 0)
    Model(c).control(ModelCmd.LoadBlob(blob))
    fp = FramePath(c, 0)
    for _ in range(4):
        fp.advance(fr, 1.0)
    recs = c.profile()
    tot = sum(r["ms"] for r in recs)
    per = {}
    for r in recs:
        if "conv2" in r["name"] or "classifier.0" in r["name"]:
            key = r["name"].split(" ")[0]
            per[key] = per.get(key, 0.0) + r["ms"] * 1e3
    print(f"== {name}: frame {tot:.3f} ms; 3x3 convs {sum(per.values()) / 1e3:.3f} ms")
    print("   " + "  ".join(f"{k.replace('backbone.', '')}={v:.0f}" for k, v in per.items()))
    c.close()
