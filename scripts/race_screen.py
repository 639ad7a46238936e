#!/usr/bin/env python3
"""Race screen for the kernels whose LDS traffic is ordered by hand (LDS-DMA with counted vmcnt + raw barriers:
conv_igemm `dma` / `dmai`, conv1x1_areg): the same frame N times on K contexts AT THE SAME TIME (so that timing varies),
every run's stride-8 logits hashed -- one distinct hash per (dtype, size) or the schedule has a race.
    python scripts/race_screen.py [runs [dtype]]        (run on an MI355X; dtype restricts the jobs, e.g. with INFUR_CONV_CFG=19 to
                                                          force conv3x3_halo.hip onto every 3x3 of the f16 mode)"""
import hashlib
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from infur_amd import processors as P
from infur_amd import weights as W

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
JOBS = [("f16", 101, 3840, 2160), ("f16", 50, 1920, 1080), ("f16", 50, 961, 541), ("f32s", 50, 1920, 1080), ("f32x", 50, 1920, 1080), ("f32", 50, 1920, 1080),
        # round 5: conv_hl.hip -- a ring of three (two) LDS images filled by DMA under a counted vmcnt, residual loads kept in flight
        # across the last K steps; INFUR_CONV_CFG = 11 / 0 / 6 / 5 / 12 / 13 / 14 forces each form onto every layer, 15 conv_hl_areg.hip (per-wave DMA bookkeeping,
        # residual slot reused in place) onto every expansion
        ("f16hl", 50, 1920, 1080), ("f16hl", 50, 961, 541), ("f16hl", 101, 3840, 2160),
        # the quantised model: i8 `dma` / `dmai` tiles and conv1x1_q8 (hand-counted vmcnt over DMA pieces, residual loads and stores);
        # 2160p so that the tuner also takes conv1x1_q8 for the expansions, 1080p for the database's choices
        ("i8", 50, 1920, 1080), ("i8", 50, 3840, 2160)]
ONLY = sys.argv[2] if len(sys.argv) > 2 else None
bad = 0
for dtype, depth, w, h in JOBS:
    if ONLY and dtype != ONLY:
        continue
    if dtype == "i8":
        from infur_amd import quantize

        blob = quantize.synth_qblob(depth=depth)
    else:
        blob = W.synth_blob(depth=depth)
    fr = W.synth_frame(h, w, index=7)
    K = 3
    hashes = [[] for _ in range(K)]

    def work(k):
        c = P.Context(device=0, dtype="f32" if dtype == "i8" else dtype)
        m = P.Model(c).control(P.ModelCmd.LoadBlob(blob))
        fp = P.FramePath(c)
        n = max(6, RUNS // (4 if w > 2000 else 1))
        for _ in range(n):
            rgba, _ = fp.advance(fr, 1.0)
            lo, la = m.lowres()
            hashes[k].append(hashlib.sha1(lo.tobytes() + la.tobytes() + np.ascontiguousarray(rgba).tobytes()).hexdigest())
        c.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    allh = [x for hs in hashes for x in hs]
    distinct = len(set(allh))
    bad += distinct != 1
    print(f"{dtype} R{depth} {w}x{h}: {len(allh)} runs on {K} concurrent contexts, {distinct} distinct result(s)")
print("RACE SCREEN", "CLEAN" if not bad else "FAILED")
sys.exit(1 if bad else 0)
