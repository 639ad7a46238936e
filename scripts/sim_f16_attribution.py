#!/usr/bin/env python3
"""VERDICT r3 item 1, step one: WHERE does the f16 mode's 2e-3 logit error come from?  CPU simulation (torch, f32
accumulation = what the f16 MFMA does: f16 products are exact in f32) of the FCN-ResNet50 forward with the roundings of
the f16 mode switched on ONE AT A TIME, and switched off one at a time, on the synthetic and the hostile parameter set,
against the FLOAT64 evaluation of the network.  Lab tooling, not product: it imports the oracle as the reference.

Rounding sites of the f16 mode (infur_amd/csrc: T = _Float16):
  w        weights stored as f16                                              ("split" = f16 hi + lo pair, 2 MFMAs, ~22 bits)
  branch   conv1 / conv2 outputs stored as f16 (the inputs of conv2 / conv3)
  trunk    the residual trunk stored as f16: pooled stem output, every conv3 + identity + ReLU output -- both its use as
           the NEXT residual and its use as the GEMM operand of conv1 / downsample / the head convs
  tread    only the trunk's use as a GEMM operand is f16 (trunk kept f32 in HBM, rounded while it is staged);
           "split" = the operand is an f16 hi + lo pair (2 MFMAs)
  head     the 3x3 head convs' outputs (inputs of the classifiers) stored as f16

    python scripts/sim_f16_attribution.py [h w]        (CPU only; ~3 min at 240x320)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import hostile as H
from infur_amd import weights as W
from oracle.infur_oracle import COracle, TorchModel

F = torch.nn.functional


def r16(x):
    return x.half().float()


def split(x):
    hi = r16(x)
    return hi, r16(x - hi)


class Sim:
    """knobs: w in {f32,f16,split}; branch, trunk, head in {f32,f16}; tread in {f32,f16,split} (only read when trunk == f32)"""

    def __init__(self, blob):
        meta, tensors = W.unpack_blob(blob)
        self.specs = W.graph(meta["depth"], meta["num_classes"], meta["aux"])
        self.params = [(torch.from_numpy(np.array(w)), torch.from_numpy(np.array(b))) for _, w, b in tensors]

    def conv(self, x, i, xmode, wmode):
        s = self.specs[i]
        w, b = self.params[i]
        kw = dict(stride=s.stride, padding=s.pad, dilation=s.dil)
        if s.role == "stem":
            return F.conv2d(x, w, b, **kw)  # exact f32 stem in every mode
        xs = {"f32": (x,), "f16": (r16(x),), "split": split(x)}[xmode]
        ws = {"f32": (w,), "f16": (r16(w),), "split": split(w)}[wmode]
        y = F.conv2d(xs[0], ws[0], b, **kw)
        for xi, xv in enumerate(xs):
            for wi, wv in enumerate(ws):
                if (xi, wi) != (0, 0) and xi + wi < 2:  # lo x lo dropped, as the split kernels do
                    y = y + F.conv2d(xv, wv, None, **kw)
        return y

    def forward(self, chw, w="f32", branch="f32", trunk="f32", tread="f32", head="f32"):
        st = lambda t, m: r16(t) if m == "f16" else t  # noqa: E731
        tr = "f16" if trunk == "f16" else tread  # how a conv reads the trunk
        specs = self.specs
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(chw, np.float32))[None]
            x = F.relu(self.conv(x, 0, "f32", "f32"))
            x = st(F.max_pool2d(x, 3, 2, 1), trunk)
            i, l3 = 1, None
            while specs[i].role == "conv1":
                has_down = specs[i + 3].role == "down"
                t = st(F.relu(self.conv(x, i, tr, w)), branch)
                t = st(F.relu(self.conv(t, i + 1, "f16" if branch == "f16" else "f32", w)), branch)
                idt = self.conv(x, i + 3, tr, w) if has_down else x
                y = st(F.relu(self.conv(t, i + 2, "f16" if branch == "f16" else "f32", w) + idt), trunk)
                name = specs[i].name
                i += 4 if has_down else 3
                x = y
                if name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
                    l3 = x
            hm = "f16" if head == "f16" else "f32"
            h = st(F.relu(self.conv(x, i, tr, w)), head)
            out = self.conv(h, i + 1, hm, w)[0]
            a = st(F.relu(self.conv(l3, i + 2, tr, w)), head)
            aux = self.conv(a, i + 3, hm, w)[0]
        return out.numpy(), aux.numpy()


ALL16 = dict(w="f16", branch="f16", trunk="f16", head="f16")
CASES = [
    ("all f32 (oracle-grade)", {}),
    ("f16 mode as shipped", ALL16),
    ("ONLY w f16", dict(w="f16")),
    ("ONLY branch f16", dict(branch="f16")),
    ("ONLY trunk f16 (store+read)", dict(trunk="f16")),
    ("ONLY trunk READ f16 (f32 store)", dict(tread="f16")),
    ("ONLY head f16", dict(head="f16")),
    ("all f16 BUT w split", dict(ALL16, w="split")),
    ("all f16 BUT w f32", dict(ALL16, w="f32")),
    ("all f16 BUT branch f32", dict(ALL16, branch="f32")),
    ("all f16 BUT trunk f32-stored (read f16)", dict(ALL16, trunk="f32", tread="f16")),
    ("all f16 BUT trunk f32, read split", dict(ALL16, trunk="f32", tread="split")),
    ("all f16 BUT head f32", dict(ALL16, head="f32")),
    # candidates for the compliant f16-rate mode
    ("CAND A: f32 trunk (read f16) + w split", dict(ALL16, trunk="f32", tread="f16", w="split")),
    ("CAND B: f32 trunk read split + w f16", dict(ALL16, trunk="f32", tread="split")),
    ("CAND C: f32 trunk read split + w split", dict(ALL16, trunk="f32", tread="split", w="split")),
    ("CAND D: f32 trunk (read f16) + head f32", dict(ALL16, trunk="f32", tread="f16", head="f32")),
]


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 320)
    co = COracle()
    sets = (("synthetic", W.synth_blob(), W.synth_frame(h, w, index=3)), ("hostile", H.hostile_blob(), H.saturated_frame(h, w)))
    rows = {}
    for sname, blob, fr in sets:
        chw = co.pack_normalize(fr)
        ref, ref_aux = (t.numpy() for t in TorchModel(blob, float64=True).forward_lowres(chw))
        sim = Sim(blob)
        for cname, kw in CASES:
            out, aux = sim.forward(chw, **kw)
            (e, r), (ea, ra) = H.errors(out, ref), H.errors(aux, ref_aux)
            rows.setdefault(cname, []).append((max(e, ea), max(r, ra)))
            print(f"{sname:9s} {w}x{h} {cname:42s} max-abs/max-abs {max(e, ea):.2e}   per-element {max(r, ra):.2e}", flush=True)
    print("\n| arithmetic | synthetic: max-abs / per-element | hostile: max-abs / per-element |\n|---|---|---|")
    for cname, _ in CASES:
        (a, b), (c, d) = rows[cname]
        print(f"| {cname} | {a:.1e} / {b:.1e} | {c:.1e} / {d:.1e} |")


if __name__ == "__main__":
    main()
