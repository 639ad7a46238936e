#!/usr/bin/env python3
"""HBM-resident frames/s for the workloads x arithmetic modes of LAB_NOTES.md section 5 (two contexts per GPU, as bench.py).
   python scripts/rate_table.py            (on an MI355X; prints a markdown table)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from infur_amd import weights as W

a = types.SimpleNamespace(contexts_per_gpu=2, no_aux=False, scale_mode=0)
rows = [("1080p FCN-ResNet50, scale 1.0 (BASELINE configs[1])", 50, 1920, 1080, 1.0, 48),
        ("1080p -> 960x540, scale 0.5 (configs[2]'s per-frame work)", 50, 1920, 1080, 0.5, 96),
        ("640x480 FCN-ResNet50 (configs[0]'s frame size)", 50, 640, 480, 1.0, 128),
        ("1080p FCN-ResNet101", 101, 1920, 1080, 1.0, 32),
        ("4K FCN-ResNet50", 50, 3840, 2160, 1.0, 16),
        ("4K FCN-ResNet101 (configs[4] is the f16 column)", 101, 3840, 2160, 1.0, 12)]
blobs, qblobs = {}, {}
print("| workload | f32 (MFMA f32) | f32s (split) | f32x | f16 | int8 (quantised model) |\n|---|---|---|---|---|---|")
for name, depth, w, h, scale, n in rows:
    blob = blobs.setdefault(depth, W.synth_blob(depth=depth))
    d_in = [torch.from_numpy(W.synth_frame(h, w, index=i)).cuda() for i in range(4)]
    oh, ow = (int(h * scale), int(w * scale))
    d_out = [torch.empty((oh, ow, 4), dtype=torch.uint8, device="cuda") for _ in d_in]
    cells = []
    for dt in ("f32", "f32s", "f32x", "f16", "i8"):
        if dt == "i8":  # the QOperator int8 form of the same network (another model file: its own arithmetic)
            from infur_amd import quantize

            qb = qblobs.setdefault(depth, quantize.synth_qblob(depth=depth))
            fps, _ = bench.resident_rate(a, 0, "f32", qb, d_in, d_out, w, h, scale, n)
        else:
            fps, _ = bench.resident_rate(a, 0, dt, blob, d_in, d_out, w, h, scale, n)
        cells.append(f"{fps:.1f}")
    print(f"| {name} | " + " | ".join(cells) + " |", flush=True)
