#!/usr/bin/env python3
"""CPU simulation (torch) of candidate arithmetic modes for the f16-rate path: logit error against f32 of
  f16                  : x ~ f16, w ~ f16, one product                       (the shipped f16 mode)
  f32s                 : hi/lo f16 pairs, ah*bh + ah*bl + al*bh              (the shipped split mode)
  f16+fp8              : ah*bh on the f16 MFMA, the two cross terms with e4m3 operands (per-tensor power-of-two scales)
  f16+fp8 (w only)     : ah*bh + fp8(ah)*fp8(bl)
fp8 MFMA runs at twice the f16 rate, so f16+fp8 costs 1 + 2 * 0.5 = 2 f16-MFMA units per product against 3 for f32s.
    python scripts/sim_fp8_cross_terms.py            (CPU only; a few minutes)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from infur_amd import weights as W
from oracle.infur_oracle import COracle, TorchModel

blob = W.synth_blob()
co = COracle()
F = torch.nn.functional
orig = F.conv2d


def split16(x):
    hi = x.half().float()
    return hi, (x - hi).half().float()


def q8(x):  # e4m3 with a per-tensor power-of-two scale (max |x| -> (128, 256])
    m = x.abs().max().item()
    if m == 0:
        return x
    s = 2.0 ** (8 - math.ceil(math.log2(m)))
    return (x * s).to(torch.float8_e4m3fn).float() / s


def make_conv(mode):
    def conv2d(x, w, b=None, **kw):
        if x.shape[1] == 3:  # the stem keeps its exact pre-processing
            return orig(x, w, b, **kw)
        xh, xl = split16(x)
        wh, wl = split16(w)
        y = orig(xh, wh, b, **kw)
        if mode == "f32s":
            y = y + orig(xh, wl, None, **kw) + orig(xl, wh, None, **kw)
        elif mode == "f16+fp8":
            y = y + orig(q8(xh), q8(wl), None, **kw) + orig(q8(xl), q8(wh), None, **kw)
        elif mode == "f16+fp8 (w only)":
            y = y + orig(q8(xh), q8(wl), None, **kw)
        return y

    return conv2d


for (w, h) in ((320, 240), (960, 540)):
    chw = co.pack_normalize(W.synth_frame(h, w, index=3))
    tm = TorchModel(blob)
    ref = tm.forward_lowres(chw)[0].numpy()
    for mode in ("f16", "f32s", "f16+fp8", "f16+fp8 (w only)"):
        tm.torch.nn.functional.conv2d = make_conv(mode)
        try:
            got = tm.forward_lowres(chw)[0].numpy()
        finally:
            tm.torch.nn.functional.conv2d = orig
        print(f"{w}x{h} {mode:18s} rel err {np.abs(got - ref).max() / np.abs(ref).max():.2e}")
