#!/usr/bin/env python3
"""WHICH arithmetic / tensor format does each of the 57 convolutions need for a mode at (about) two
f16-MFMA units per product that stays inside north_star's 1e-3 (max-abs) AND 1e-2 per element on the hostile parameter set?
Extends scripts/sim_hi_lo8.py from tensor classes to a PER-LAYER assignment (torch CPU, f32 accumulation, direct convolutions,
FLOAT64 reference; lab tooling: it imports the oracle as the reference).

  tensor formats    f32 (4 B), f16 (2 B), h5 (3 B: f16 hi + e5m2 of (x - hi) * 2^11)
  products          f16    hi*hi                                                         1 unit
                    x5     hi*hi (f16) + e5(a_hi) e5(w_lo) + e5(a_lo) e5(w_hi)  (bf8 MFMA)   2 units, every bf8 operand rounded (RNE)
                    x5t    the activation's hi byte is the TOP BYTE of its f16 (one v_perm per 4 values, no conversion)
                    x5td   x5t + the truncation's mean (hi / trunc(hi) ~ 1 + d) folded into the static w_lo plane
                    x5ttd  x5td + the weight's hi byte truncated too (no w_hi8 plane in LDS), its mean folded into the a_lo plane
                    x25    a_hi * (w_hi + w_lo) on the f16 MFMA (two of them) + e5(a_lo) e5(w_hi)      2.5 units
                    f32s   three f16 MFMAs                                                        3 units
  wopt              w_hi chosen among the two f16 neighbours of w so that w - w_hi is closest to an e5m2 value (free: load time)

Experiments: (1) uniform assignments; (2) one layer at a time promoted to f32s from the uniform 2-unit mode: where the error is
made; (3) one tensor at a time demoted to f16: which tensors could stay 2-byte; (4) greedy: promote the layer that buys most until
both bars hold with 2x margin on max-abs.

    python scripts/sim_hl_assign.py [h w] [experiments, e.g. 1,2]"""
import math
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
LO = 2048.0  # scale of the lo planes


def r16(x):
    return x.clamp(-65504, 65504).half().float()


def e5(x):
    return x.clamp(-57344, 57344).to(torch.float8_e5m2).float()


def trunc5(x16):
    b = x16.half().view(torch.int16) & torch.tensor(-256, dtype=torch.int16)
    return b.view(torch.float16).float()


def p2scale(m, top):
    return 2.0 ** math.floor(math.log2(top / m)) if m > 0 else 1.0


def store(x, fmt, lo_gain=1.0):
    if fmt == "f32":
        return x
    hi = r16(x)
    if fmt == "f16":
        return hi
    if fmt == "h5":
        return hi + e5((x - hi) * LO) / LO
    raise ValueError(fmt)


# mean of hi / trunc5(hi) - 1 over f16 mantissas (uniform): the multiplicative debias of a truncated hi byte
def trunc_gain():
    m = torch.arange(1024, dtype=torch.float32)
    full = 1.0 + m / 1024.0
    tr = 1.0 + torch.floor(m / 256.0) / 4.0
    return float((full / tr).mean())


DEBIAS = trunc_gain()


def split_w(w, wopt):
    """w (already scaled) -> f16 hi, f32 remainder"""
    wh = r16(w)
    if not wopt:
        return wh, w - wh
    # the other f16 neighbour on the side of w
    bits = wh.half().view(torch.int16).to(torch.int32)
    # neighbour towards w: sign-magnitude f16 -> +-1 on the magnitude
    mag = bits & 0x7FFF
    sgn = bits & ~0x7FFF
    toward_larger_mag = ((w.abs() > wh.abs())).to(torch.int32)
    mag2 = torch.where(toward_larger_mag.bool(), mag + 1, torch.clamp(mag - 1, min=0))
    alt = (sgn | mag2).to(torch.int16).view(torch.float16).float()
    alt = torch.where(torch.isfinite(alt), alt, wh)
    err0 = (w - wh - e5((w - wh) * LO) / LO).abs()
    err1 = (w - alt - e5((w - alt) * LO) / LO).abs()
    pick = err1 < err0
    wh2 = torch.where(pick, alt, wh)
    return wh2, w - wh2


def product(x, xfmt, w, b, mode, kw, wopt=False):
    """conv2d(x, w) + b with the matrix-core arithmetic `mode`.  x is the value the consumer sees (already through `store`);
    for h5 tensors the lo plane is what the producer stored, so a_lo is exact e5m2 already."""
    if mode == "f32":
        return F.conv2d(x, w, b, **kw)
    ws = p2scale(w.abs().max().item(), 16383.0)
    w = w * ws
    xh = r16(x)
    xl = x - xh
    wh, wl = split_w(w, wopt and mode != "f16")
    y = F.conv2d(xh, wh, None, **kw)
    if mode == "f16":
        pass
    elif mode == "f32s":
        y = y + F.conv2d(xh, r16(wl), None, **kw) + F.conv2d(r16(xl), wh, None, **kw)
    elif mode == "x25":
        y = y + F.conv2d(xh, r16(wl), None, **kw) + F.conv2d(e5(xl * LO), e5(w), None, **kw) / LO
    elif mode == "x5":
        y = y + (F.conv2d(e5(xh), e5(wl * LO), None, **kw) + F.conv2d(e5(xl * LO), e5(w), None, **kw)) / LO
    elif mode == "x5t":
        y = y + (F.conv2d(trunc5(xh), e5(wl * LO), None, **kw) + F.conv2d(e5(xl * LO), e5(w), None, **kw)) / LO
    elif mode == "x5td":
        y = y + (F.conv2d(trunc5(xh), e5(wl * LO * DEBIAS), None, **kw) + F.conv2d(e5(xl * LO), e5(w), None, **kw)) / LO
    elif mode == "x5ttd":
        y = y + (F.conv2d(trunc5(xh), e5(wl * LO * DEBIAS), None, **kw) + F.conv2d(e5(xl * LO * DEBIAS), trunc5(wh), None, **kw)) / LO
    else:
        raise ValueError(mode)
    y = y / ws
    return y if b is None else y + b.view(1, -1, 1, 1)


UNITS = {"f32": 16.0, "f16": 1.0, "x5": 2.0, "x5t": 2.0, "x5td": 2.0, "x5ttd": 2.0, "x25": 2.5, "f32s": 3.0}


class Sim:
    def __init__(self, blob):
        meta, tensors = W.unpack_blob(blob)
        self.specs = W.graph(meta["depth"], meta["num_classes"], meta["aux"])
        self.params = [(torch.from_numpy(np.array(w)), torch.from_numpy(np.array(b))) for _, w, b in tensors]
        self.n = len(self.specs)

    def flops(self, h, w):
        """direct-conv MACs per layer at h x w (relative weights for the unit count)"""
        out = []
        for s in self.specs:
            out.append(float(np.prod(self.params[len(out)][0].shape)))
        return out

    def forward(self, chw, modes, fmts, wopt=False):
        """modes[i]: product of conv i; fmts[i]: storage format of conv i's OUTPUT tensor (after its epilogue: for conv3 the
        block output, i.e. the trunk); fmts[0] = the pooled stem output"""
        specs = self.specs

        def conv(x, i):
            s = specs[i]
            w, b = self.params[i]
            kw = dict(stride=s.stride, padding=s.pad, dilation=s.dil)
            return F.conv2d(x, w, b, **kw) if s.role == "stem" else product(x, None, w, b, modes[i], kw, wopt)

        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(chw, np.float32))[None]
            x = store(F.max_pool2d(F.relu(conv(x, 0)), 3, 2, 1), fmts[0])
            i, l3 = 1, None
            while specs[i].role == "conv1":
                has_down = specs[i + 3].role == "down"
                t = store(F.relu(conv(x, i)), fmts[i])
                t = store(F.relu(conv(t, i + 1)), fmts[i + 1])
                idt = conv(x, i + 3) if has_down else x
                y = store(F.relu(conv(t, i + 2) + idt), fmts[i + 2])
                name = specs[i].name
                i += 4 if has_down else 3
                x = y
                if name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
                    l3 = x
            out = conv(store(F.relu(conv(x, i)), fmts[i]), i + 1)[0]
            aux = conv(store(F.relu(conv(l3, i + 2)), fmts[i + 2]), i + 3)[0]
        return out.numpy(), aux.numpy()


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 320)
    exps = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else {"1", "2", "3"}
    co = COracle()
    sets = []
    for sname, blob, fr in (("synthetic", W.synth_blob(), W.synth_frame(h, w, index=3)), ("hostile", H.hostile_blob(), H.saturated_frame(h, w))):
        chw = co.pack_normalize(fr)
        ref, ref_aux = (t.numpy() for t in TorchModel(blob, float64=True).forward_lowres(chw))
        sets.append((sname, Sim(blob), chw, ref, ref_aux))
    n = sets[0][1].n
    names = [s.name for s in sets[0][1].specs]

    def run(modes, fmts, wopt=False):
        res = []
        for sname, sim, chw, ref, ref_aux in sets:
            out, aux = sim.forward(chw, modes, fmts, wopt)
            (e, r), (ea, ra) = H.errors(out, ref), H.errors(aux, ref_aux)
            res.append((max(e, ea), max(r, ra)))
        return res

    def show(tag, res):
        print(f"{tag:58s} synthetic {res[0][0]:.2e} / {res[0][1]:.2e}   hostile {res[1][0]:.2e} / {res[1][1]:.2e}", flush=True)

    if "1" in exps:
        print(f"# (1) uniform assignments, {w}x{h}: max-abs/max-abs / per-element   (debias factor {DEBIAS:.4f})")
        for mode in ("x5", "x5t", "x5td", "x5ttd", "x25", "f32s"):
            for fmt in ("h5", "f32"):
                show(f"{fmt} / {mode}", run([mode] * n, [fmt] * n))
        show("h5 / x5 wopt", run(["x5"] * n, ["h5"] * n, True))
        show("h5 / x5td wopt", run(["x5td"] * n, ["h5"] * n, True))
        show("h5 / x5ttd wopt", run(["x5ttd"] * n, ["h5"] * n, True))
        # classifiers in three units (their cost is nil)
        m = ["x5td"] * n
        for i, nm in enumerate(names):
            if nm.endswith("classifier.4"):
                m[i] = "f32s"
        show("h5 / x5td, classifier.4 convs f32s", run(m, ["h5"] * n))
    base_mode = "x5td"
    if "2" in exps:
        print(f"# (2) one layer promoted to f32s from uniform h5 / {base_mode}")
        base = run([base_mode] * n, ["h5"] * n)
        show("base", base)
        for i in range(1, n):
            m = [base_mode] * n
            m[i] = "f32s"
            show(f"  {i:2d} {names[i]} -> f32s", run(m, ["h5"] * n))
    if "3" in exps:
        print(f"# (3) one tensor demoted to f16 from uniform h5 / {base_mode}")
        for i in range(0, n):
            f = ["h5"] * n
            f[i] = "f16"
            show(f"  {i:2d} {names[i]} output -> f16", run([base_mode] * n, f))
    if "4" in exps:
        print("# (4) classes of tensors in f16")
        roles = [s.role for s in sets[0][1].specs]
        for cls in ("conv1", "conv2"):
            f = ["f16" if roles[i] == cls else "h5" for i in range(n)]
            show(f"  all {cls} outputs f16", run([base_mode] * n, f))
        f = ["f16" if roles[i] in ("conv1", "conv2") else "h5" for i in range(n)]
        show("  all branch tensors f16", run([base_mode] * n, f))


if __name__ == "__main__":
    main()
