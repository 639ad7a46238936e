#!/usr/bin/env python3
"""Errors of every arithmetic mode x Winograd tile on the hostile parameter set (tests/hostile.py) against a float64
evaluation: per-layer worst and logits, at 320x240 and (with --full) 1920x1080.  Run on an MI355X."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostile as H  # noqa: E402
from infur_amd import weights as W  # noqa: E402
from infur_amd.processors import Context, Model, ModelCmd  # noqa: E402
from oracle.infur_oracle import COracle, TorchModel  # noqa: E402

full = "--full" in sys.argv
blob = H.hostile_blob()
co = COracle()
ref64 = TorchModel(blob, float64=True)
sizes = [(240, 320)] + ([(1080, 1920)] if full else [])
for (h, w) in sizes:
    fr = H.saturated_frame(h, w, index=2)
    taps = {}
    ref, ref_aux = ref64.forward_lowres(co.pack_normalize(fr), taps=taps)
    ref = ref.numpy()
    print(f"== {w}x{h}: |logits| max {np.abs(ref).max():.3g}")
    for dtype in ("f32", "f32s", "f32x", "f16"):
        for tile, name in ((6, "F(6x6)"), (4, "F(4x4)"), (2, "F(2x2)"), (-1, "direct")):
            if dtype == "f16" and tile >= 0:
                continue  # (the f16 mode runs every 3x3 directly)
            c = Context(device=0, dtype=dtype, keep_activations=True, winograd_tile=max(tile, 0),
                        winograd_min_cin=0xFFFFFFFF if tile < 0 else 0)
            m = Model(c).control(ModelCmd.LoadBlob(blob))
            out = []
            m.advance(fr, out)
            lo, _ = m.lowres()
            worst, wname = 0.0, ""
            wrel, wrname = 0.0, ""
            for i, spec in enumerate(W.graph(50)):
                r = taps[spec.name].numpy()
                buf = np.empty(r.shape, np.float32)
                cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
                c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
                e, er = H.errors(buf, r)
                if e > worst:
                    worst, wname = e, spec.name
                if er > wrel:
                    wrel, wrname = er, spec.name
            e, er = H.errors(lo, ref)
            extra = ""
            if dtype in ("f32s", "f32x"):
                a, wm, sat = c.split_range()
                extra = f" | range act {a:.3g} wino {wm:.3g} sat {sat}"
            print(f"{dtype:5s} {name:7s} logits {e:.2e} / {er:.2e}   per-layer worst {worst:.2e} ({wname}) / {wrel:.2e} ({wrname}){extra}")
            c.close()
