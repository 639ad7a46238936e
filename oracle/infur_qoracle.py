"""CPU oracle for QUANTISED models (test infrastructure -- never part of the product path).

The reference's own tests load the QOperator int8 form of the network, `fcn-resnet50-12-int8.onnx`
(infur-test-gen/build.rs:88-93, infur/src/predict_onnx.rs:357-381), and ONNX Runtime executes it inside `session.run`
(predict_onnx.rs:138).  Neither the file nor ONNX Runtime exists in this image (parity unpinned against ORT, as for the
float model); what IS exactly definable is the integer arithmetic of the operators, restated here from their ONNX /
com.microsoft definitions:

    QuantizeLinear    q = sat_u8(round(x / s) + zp)                                   (onnx: QuantizeLinear-10)
    QLinearConv       acc = sum (x - x_zp) * (w - w_zp) + bias  (int32, exact)          (onnx: QLinearConv-10)
                      y = sat_u8(round(float(acc) * (x_s * w_s[o] / y_s)) + y_zp)
    QLinearAdd        c = sat_u8(round((a - a_zp) * (a_s / c_s) + (b - b_zp) * (b_s / c_s)) + c_zp)   (com.microsoft)
    MaxPool on u8, DequantizeLinear x = (q - zp) * s, Resize(linear, pytorch_half_pixel) on the dequantised logits

with round = round-half-to-even and every floating-point step ONE IEEE f32 operation in the order written (the order
ONNX Runtime's MLAS requantisation uses: int32 -> f32, one multiply, nearbyint, + zero point, saturate).  The integer
convolution is evaluated exactly (float64 conv2d on integer-valued operands: |acc| < 2^53).

`quantise_model` makes a quantised model out of the seeded synthetic float one (static quantisation as ONNX Runtime's /
Neural Compressor's tools do it: u8 activations with per-tensor scale and zero point from calibration ranges, s8 weights
with per-output-channel scales and zero point 0, int32 bias in units of x_s * w_s[o]; a tensor that follows a ReLU is
calibrated from 0, so its zero point is 0 and the ReLU is the clamp).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from infur_amd import weights as W

f32 = np.float32


def _sat_u8(v: np.ndarray) -> np.ndarray:
    return np.clip(v, 0, 255).astype(np.uint8)


def quantize_linear(x: np.ndarray, scale: float, zp: int) -> np.ndarray:
    return _sat_u8(np.rint(x.astype(f32) / f32(scale)) + f32(zp))


def requantize(acc: np.ndarray, mult: np.ndarray, y_zp: int) -> np.ndarray:
    """acc int [C,H,W], mult f32 [C]: sat_u8(round(f32(acc) * mult) + zp)"""
    t = acc.astype(f32) * mult.astype(f32)[:, None, None]
    return _sat_u8(np.rint(t) + f32(y_zp))


def qlinear_add(a: np.ndarray, b: np.ndarray, p: W.QAdd) -> np.ndarray:
    ra, rb = f32(p.a_scale) / f32(p.c_scale), f32(p.b_scale) / f32(p.c_scale)
    ta = (a.astype(np.int32) - int(p.a_zp)).astype(f32) * ra
    tb = (b.astype(np.int32) - int(p.b_zp)).astype(f32) * rb
    return _sat_u8(np.rint(ta + tb) + f32(p.c_zp))


def conv_mult(c: W.QConv) -> np.ndarray:
    return (f32(c.x_scale) * c.w_scale.astype(f32)) / f32(c.y_scale)


def qconv(x_u8: np.ndarray, c: W.QConv, spec: W.ConvSpec) -> np.ndarray:
    """exact integer convolution [Cin,H,W] u8 -> int64 [Cout,OH,OW] (padding = x_zp, i.e. contributes nothing)"""
    import torch

    xs = torch.from_numpy(x_u8.astype(np.float64) - float(c.x_zp))[None]
    w = torch.from_numpy(c.w.astype(np.float64))
    y = torch.nn.functional.conv2d(xs, w, None, stride=spec.stride, padding=spec.pad, dilation=spec.dil)[0].numpy()
    acc = np.rint(y).astype(np.int64) + c.bias.astype(np.int64)[:, None, None]
    assert np.abs(acc).max() < 2**31
    return acc


def maxpool_u8(x: np.ndarray) -> np.ndarray:
    import torch

    return torch.nn.functional.max_pool2d(torch.from_numpy(x.astype(np.float32))[None], 3, 2, 1)[0].numpy().astype(np.uint8)


def qforward(blob: bytes, chw: np.ndarray, taps: Optional[Dict[str, np.ndarray]] = None) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """normalised image [3,h,w] f32 (the reference's pre-proc, predict_onnx.rs:126-137) -> dequantised output-stride-8 logits
    (out [K,lh,lw] f32, aux or None).  `taps`: filled with every conv's u8 output (conv3: AFTER the block's QLinearAdd, as the
    fused HIP launch produces it; downsample: its own u8 output)."""
    meta, convs, adds = W.unpack_qblob(blob)
    specs = W.graph(meta["depth"], meta["num_classes"], meta["aux"])
    assert [c.name for c in convs] == [s.name for s in specs]
    it = iter(zip(specs, convs))
    add_it = iter(adds)

    def conv(x):
        s, c = next(it)
        return requantize(qconv(x, c, s), conv_mult(c), c.y_zp), s, c

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    x = quantize_linear(chw, convs[0].x_scale, convs[0].x_zp)
    tap("input", x)
    x, s, _ = conv(x)
    tap(s.name, x)
    x = maxpool_u8(x)
    i, l3 = 1, None
    while specs[i].role == "conv1":
        has_down = specs[i + 3].role == "down"
        t, s1, _ = conv(x)
        tap(s1.name, t)
        t, s2, _ = conv(t)
        tap(s2.name, t)
        y3, s3, _ = conv(t)
        idt = x
        if has_down:
            sd, cd = next(it)
            idt = requantize(qconv(x, cd, sd), conv_mult(cd), cd.y_zp)
            tap(sd.name, idt)
        x = qlinear_add(y3, idt, next(add_it))
        tap(s3.name, x)
        i += 4 if has_down else 3
        if s3.name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
            l3 = x

    def head(feat):
        h, sh, _ = conv(feat)
        tap(sh.name, h)
        q, sq, cq = conv(h)
        tap(sq.name, q)
        return (q.astype(np.int32) - int(cq.y_zp)).astype(f32) * f32(cq.y_scale)

    out = head(x)
    aux = head(l3) if meta["aux"] else None
    return out, aux


# --------------------------------------------------------------------------- #
# static quantisation of the synthetic float model
# --------------------------------------------------------------------------- #
def _act_params(lo: float, hi: float) -> Tuple[float, int]:
    lo, hi = min(0.0, float(lo)), max(0.0, float(hi))
    scale = max((hi - lo) / 255.0, 1e-8)
    zp = int(np.clip(np.rint(-lo / scale), 0, 255))
    return float(f32(scale)), zp


def quantise_model(float_blob: bytes, calib_chw: List[np.ndarray]) -> bytes:
    """INFURW01 float blob + calibration inputs (normalised [3,h,w] f32) -> INFURQ01 quantised blob"""
    import torch

    F = torch.nn.functional
    meta, tensors = W.unpack_blob(float_blob)
    specs = W.graph(meta["depth"], meta["num_classes"], meta["aux"])
    params = [(torch.from_numpy(np.array(w)), torch.from_numpy(np.array(b))) for _, w, b in tensors]
    rng: Dict[str, List[float]] = {}

    def see(name, t):
        lo, hi = float(t.min()), float(t.max())
        r = rng.setdefault(name, [lo, hi])
        r[0], r[1] = min(r[0], lo), max(r[1], hi)

    with torch.no_grad():
        for chw in calib_chw:
            x = torch.from_numpy(np.ascontiguousarray(chw, np.float32))[None]
            see("input", x)
            it = iter(zip(specs, params))

            def conv(x, relu):
                s, (w, b) = next(it)
                y = F.conv2d(x, w, b, stride=s.stride, padding=s.pad, dilation=s.dil)
                if relu:
                    y = F.relu(y)
                see(s.name, y)
                return y, s

            x, _ = conv(x, True)
            x = F.max_pool2d(x, 3, 2, 1)
            i, l3, blk = 1, None, 0
            while specs[i].role == "conv1":
                has_down = specs[i + 3].role == "down"
                t, _ = conv(x, True)
                t, _ = conv(t, True)
                y3, s3 = conv(t, False)  # the QLinearConv of conv3 has no ReLU: the Add follows
                idt = x
                if has_down:
                    idt, _ = conv(x, False)
                x = F.relu(y3 + idt)
                see(f"add{blk}", x)
                blk += 1
                i += 4 if has_down else 3
                if s3.name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
                    l3 = x
            h, _ = conv(x, True)
            conv(h, False)
            if meta["aux"]:
                a, _ = conv(l3, True)
                conv(a, False)

    act = {k: _act_params(*v) for k, v in rng.items()}
    convs: List[W.QConv] = []
    adds: List[W.QAdd] = []
    # which tensor feeds each conv: walk the graph again, names only
    src_of: Dict[str, str] = {}
    cur, i, blk, l3n = "backbone.conv1", 1, 0, None
    src_of["backbone.conv1"] = "input"
    while specs[i].role == "conv1":
        has_down = specs[i + 3].role == "down"
        src_of[specs[i].name] = cur
        src_of[specs[i + 1].name] = specs[i].name
        src_of[specs[i + 2].name] = specs[i + 1].name
        if has_down:
            src_of[specs[i + 3].name] = cur
        a_p, c_p = act[specs[i + 2].name], act[f"add{blk}"]
        b_p = act[specs[i + 3].name] if has_down else act[cur]
        adds.append(W.QAdd(a_p[0], a_p[1], b_p[0], b_p[1], c_p[0], c_p[1]))
        act[f"blockout{blk}"] = c_p
        name3 = specs[i + 2].name
        cur = f"add{blk}"
        blk += 1
        i += 4 if has_down else 3
        if name3.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
            l3n = cur
    src_of[specs[i].name] = cur
    src_of[specs[i + 1].name] = specs[i].name
    if meta["aux"]:
        src_of[specs[i + 2].name] = l3n
        src_of[specs[i + 3].name] = specs[i + 2].name
    for s, (_, w, b) in zip(specs, tensors):
        xs, xz = act[src_of[s.name]]
        ys, yz = act[s.name]
        w = np.asarray(w, np.float64)
        amax = np.abs(w).reshape(s.cout, -1).max(1)
        ws = np.maximum(amax / 127.0, 1e-12).astype(f32)
        wq = np.clip(np.rint(w / ws.astype(np.float64)[:, None, None, None]), -127, 127).astype(np.int8)
        bq = np.rint(np.asarray(b, np.float64) / (np.float64(f32(xs)) * ws.astype(np.float64))).astype(np.int64)
        bq = np.clip(bq, -2**31 + 1, 2**31 - 1).astype(np.int32)
        convs.append(W.QConv(s.name, wq, ws, bq, xs, xz, ys, yz))
    return W.pack_qblob(convs, adds, meta["depth"], meta["num_classes"], meta["aux"])


def synth_qblob(depth: int = 50, calib: int = 3, size: Tuple[int, int] = (96, 128)) -> bytes:
    """the seeded synthetic model, statically quantised on `calib` synthetic frames"""
    from oracle.infur_oracle import COracle

    co = COracle()
    frames = [co.pack_normalize(W.synth_frame(size[0], size[1], index=100 + k)) for k in range(calib)]
    return quantise_model(W.synth_blob(depth=depth), frames)
