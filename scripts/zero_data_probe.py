#!/usr/bin/env python3
"""Same kernels, all-zero operands (weights and frame): if the big f16 GEMMs run markedly faster than on random data the
K loop is limited by the package's power/clock management, not by its instruction schedule (MI355X_MICROARCH.md, DVFS).
   python scripts/zero_data_probe.py [dtype] [depth] [w] [h]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 101
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
blob = bytearray(W.synth_blob(depth=depth))
n = int.from_bytes(blob[20:24], "little")
hdr = (32 + n * 80 + 63) & ~63
for label, zero in (("random", False), ("zeros", True), ("random", False), ("zeros", True)):
    b = bytes(blob[:hdr]) + bytes(len(blob) - hdr) if zero else bytes(blob)
    fr = np.zeros((h, w, 3), np.uint8) if zero else W.synth_frame(h, w)
    c = Context(device=0, dtype=dt, profile=True)
    Model(c).control(ModelCmd.LoadBlob(b))
    d_in = torch.from_numpy(fr).cuda()
    d_out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    fp = FramePath(c)
    c.L.infur_profile_enable(c.h, 0)
    for _ in range(4):
        fp.advance_dev(d_in.data_ptr(), w, h, 1.0, d_out.data_ptr(), d_out.numel())
    c.L.infur_profile_enable(c.h, 1)
    fp.advance_dev(d_in.data_ptr(), w, h, 1.0, d_out.data_ptr(), d_out.numel())
    c.synchronize()
    recs = {r["name"]: r for r in c.profile()}
    tot = sum(r["ms"] for r in recs.values())
    pick = ["classifier.0", "backbone.layer4.1.conv2", "backbone.layer3.5.conv2", "backbone.layer4.1.conv1", "backbone.layer3.5.conv3"]
    print(label, f"frame {tot:.2f} ms", " ".join(f"{k.split('.')[-2]}.{k.split('.')[-1]}={recs[k]['flops'] / recs[k]['ms'] / 1e9:.0f}TF" for k in pick if k in recs), flush=True)
    c.close()
