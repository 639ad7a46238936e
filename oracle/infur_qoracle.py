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
    MaxPool on u8, DequantizeLinear x = (q - zp) * s, Resize(linear, pytorch_half_pixel) on the dequantised logits -- or, for files that
    keep Resize on the u8 tensor (QLinearConv -> Resize -> DequantizeLinear): the float interpolation of the codes truncated to u8
    (onnxruntime's UpsampleBilinear<uint8_t>: static_cast), then DequantizeLinear

with round = round-half-to-even and every floating-point step ONE IEEE f32 operation in the order written (the order
ONNX Runtime's MLAS requantisation uses: int32 -> f32, one multiply, nearbyint, + zero point, saturate).  The integer
convolution is evaluated exactly (float64 conv2d on integer-valued operands: |acc| < 2^53).

The quantised model itself comes from infur_amd/quantize.py (static quantisation of the seeded synthetic float model).
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


def qforward_codes(blob: bytes, chw: np.ndarray):
    """models whose file resizes the u8 logits before DequantizeLinear (INFURQ01 flag bit 0): the heads' u8 codes as f32
    [K,lh,lw], and each head's (zero point, scale)"""
    meta, convs, _ = W.unpack_qblob(blob)
    lo, aux = qforward(blob, chw)
    heads = [c for c in convs if c.name.endswith("classifier.4")]
    out = []
    for logits, c in zip((lo, aux), heads):
        if logits is None:
            out.append(None)
            continue
        codes = np.rint(logits.astype(np.float64) / np.float64(f32(c.y_scale))) + c.y_zp  # exact: logits = (q - zp) * scale in f32
        assert ((codes >= 0) & (codes <= 255)).all() and (((codes - c.y_zp).astype(f32) * f32(c.y_scale)) == logits).all()
        out.append(codes.astype(f32))
    return out, [(int(c.y_zp), float(c.y_scale)) for c in heads]


def resize_u8_then_dequantise(codes: np.ndarray, zp: int, scale: float, h: int, w: int, upsample) -> np.ndarray:
    """ONNX Runtime's UpsampleBilinear<uint8_t> -- the float interpolation (`upsample`: the oracle's bilinear, the same expression
    as for float tensors), static_cast to uint8_t = truncation -- followed by DequantizeLinear"""
    up = np.trunc(upsample(codes, h, w)).astype(f32)
    return ((up - f32(zp)) * f32(scale)).astype(f32)


def synth_qblob(depth: int = 50) -> bytes:
    from infur_amd import quantize

    return quantize.synth_qblob(depth=depth)
