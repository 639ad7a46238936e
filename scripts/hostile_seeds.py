#!/usr/bin/env python3
"""the accuracy of the reduced-width modes as a DISTRIBUTION over hostile parameter sets, not one sample.
For every (seed, base_seed) of tests/hostile.py::SEEDS and two frames: logits of the HIP modes against a FLOAT64 evaluation of the
network (max-abs / max-abs ; worst per-element relative error over |ref| > 1e-2 max |ref|), and -- CPU side, same sets -- the
simulation of scripts/sim_hl_assign.py with EXACT products on three-byte tensors (the format's own floor) and with the kernel's
products (both hi bytes truncated, debiased), so that the kernel's excess over the format is visible.

    python scripts/hostile_seeds.py [w h] [modes, e.g. f16hl,f32x] [sim: 0/1]        # on an MI355X"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np

import hostile as H
from infur_amd.processors import Context, FramePath, Model, ModelCmd
from oracle.infur_oracle import COracle, TorchModel

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 540)
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f16hl", "f32x"]
do_sim = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
co = COracle()
if do_sim:
    import sim_hl_assign as S
worst = {}
print(f"# hostile sets x frames at {w}x{h}: max-abs/max-abs ; worst per-element (out and aux heads, the larger of the two)")
for seed, base in H.SEEDS:
    blob = H.hostile_blob(seed=seed, base_seed=base)
    ref_model = TorchModel(blob, float64=True)
    for index in (2, 7):
        fr = H.saturated_frame(h, w, index=index)
        chw = co.pack_normalize(fr)
        t0 = time.time()
        ref, ref_aux = (t.numpy() for t in ref_model.forward_lowres(chw))
        row = [f"seed {seed:#x} base {'lib' if base is None else hex(base)} frame {index}"]
        for dt in modes:
            c = Context(device=0, dtype=dt)
            m = Model(c).control(ModelCmd.LoadBlob(blob))
            FramePath(c).advance(fr, 1.0)
            lo, la = m.lowres()
            (e, r), (ea, ra) = H.errors(lo, ref), H.errors(la, ref_aux)
            c.close()
            row.append(f"{dt} {max(e, ea):.2e} ; {max(r, ra):.2e}")
            k = worst.setdefault(dt, [0.0, 0.0])
            k[0], k[1] = max(k[0], e, ea), max(k[1], r, ra)
        if do_sim:
            sim = S.Sim(blob)
            n = sim.n
            for tag, md in (("sim exact products on 3-byte tensors", "f32"), ("sim kernel products (x5ttd)", "x5ttd")):
                out, aux = sim.forward(chw, [md] * n, ["h5"] * n)
                (e, r), (ea, ra) = H.errors(out, ref), H.errors(aux, ref_aux)
                row.append(f"{tag} {max(e, ea):.2e} ; {max(r, ra):.2e}")
                k = worst.setdefault(tag, [0.0, 0.0])
                k[0], k[1] = max(k[0], e, ea), max(k[1], r, ra)
        print(" | ".join(row), f"({time.time() - t0:.0f} s)", flush=True)
print("# worst over all sets and frames")
for k, (a, b) in worst.items():
    print(f"{k:45s} {a:.2e} ; {b:.2e}")
