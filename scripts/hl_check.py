#!/usr/bin/env python3
"""INFUR_DTYPE_F16_HL ("f16hl", conv_hl.hip) against a float64 evaluation: per layer and logits, synthetic and hostile parameters,
direct convolutions and every Winograd tile, at 320x240 and (with --full) 1920x1080; --time adds per-layer HIP-event times of a
1080p frame.  Run on an MI355X.    python scripts/hl_check.py [--full] [--time] [--modes f16hl,f32x]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostile as H  # noqa: E402
from infur_amd import weights as W  # noqa: E402
from infur_amd.processors import Context, FramePath, Model, ModelCmd  # noqa: E402
from oracle.infur_oracle import COracle, TorchModel  # noqa: E402

full = "--full" in sys.argv
modes = ["f16hl"]
if "--modes" in sys.argv:
    modes = sys.argv[sys.argv.index("--modes") + 1].split(",")
co = COracle()


def probe(tag, blob, frame_fn, sizes):
    ref64 = TorchModel(blob, float64=True)
    for (h, w) in sizes:
        fr = frame_fn(h, w)
        taps = {}
        ref, ref_aux = ref64.forward_lowres(co.pack_normalize(fr), taps=taps)
        ref, ref_aux = ref.numpy(), ref_aux.numpy()
        print(f"== {tag} {w}x{h}: |logits| max {np.abs(ref).max():.3g}", flush=True)
        for dtype in modes:
            for tile, name in ((-1, "direct"), (6, "F(6x6)"), (4, "F(4x4)")):
                c = Context(device=0, dtype=dtype, keep_activations=True, winograd_tile=max(tile, 0), winograd_min_cin=0xFFFFFFFF if tile < 0 else 0)
                m = Model(c).control(ModelCmd.LoadBlob(blob))
                out = []
                m.advance(fr, out)
                lo, la = m.lowres()
                worst, wname, wrel, wrname = 0.0, "", 0.0, ""
                lines = []
                for i, spec in enumerate(W.graph(50)):
                    r = taps[spec.name].numpy()
                    buf = np.empty(r.shape, np.float32)
                    cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
                    c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
                    e, er = H.errors(buf, r)
                    lines.append(f"      {i:2d} {spec.name:36s} {e:.2e} / {er:.2e}")
                    if e > worst:
                        worst, wname = e, spec.name
                    if er > wrel:
                        wrel, wrname = er, spec.name
                (e, er), (ea, era) = H.errors(lo, ref), H.errors(la, ref_aux)
                print(f"{dtype:6s} {name:7s} logits {max(e, ea):.2e} / {max(er, era):.2e}   per-layer worst {worst:.2e} ({wname}) / {wrel:.2e} ({wrname})", flush=True)
                if worst > 5e-3 or "--layers" in sys.argv:
                    print("\n".join(lines), flush=True)
                c.close()
                # the frame path proper: fused stem + pool, conv3 + downsample as one two-source launch (no per-layer read-back)
                c = Context(device=0, dtype=dtype, winograd_tile=max(tile, 0), winograd_min_cin=0xFFFFFFFF if tile < 0 else 0)
                m = Model(c).control(ModelCmd.LoadBlob(blob))
                FramePath(c, 0).advance(fr, 1.0)
                lo, la = m.lowres()
                (e, er), (ea, era) = H.errors(lo, ref), H.errors(la, ref_aux)
                print(f"{dtype:6s} {name:7s} fused frame path: logits {max(e, ea):.2e} / {max(er, era):.2e}", flush=True)
                c.close()


sizes = [(240, 320)] + ([(1080, 1920)] if full else [])
if "--time-only" not in sys.argv:
  probe("synthetic", W.synth_blob(), lambda h, w: W.synth_frame(h, w, index=3), sizes[:1])
  probe("hostile", H.hostile_blob(), lambda h, w: H.saturated_frame(h, w, index=2), sizes)

if "--time" in sys.argv or "--time-only" in sys.argv:
    blob = W.synth_blob()
    fr = W.synth_frame(1080, 1920, index=1)
    for dtype in modes:
        c = Context(device=0, dtype=dtype, profile=True)
        Model(c).control(ModelCmd.LoadBlob(blob))
        fp = FramePath(c, 0)
        for _ in range(4):
            fp.advance(fr, 1.0)
        recs = c.profile()
        tot = sum(r["ms"] for r in recs)
        print(f"== {dtype} 1920x1080 per-layer (one context): {tot:.3f} ms of kernels")
        for r in recs:
            tf = r["flops"] / r["ms"] / 1e9 if r["ms"] > 0 else 0.0
            gb = r["bytes"] / r["ms"] / 1e6 if r["ms"] > 0 else 0.0
            print(f"   {r['name']:44s} {r['kernel']:28s} {r['ms'] * 1e3:8.1f} us  {tf:8.1f} TFLOP/s  {gb:8.1f} GB/s")
        c.close()
