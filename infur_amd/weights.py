"""FCN-ResNet graph description, synthetic weights and the flat weight blob.

The reference loads ``fcn-resnet50-12*.onnx`` through ONNX Runtime
(infur/src/predict_onnx.rs:288-309; file fetched by infur-test-gen/build.rs:88-93).
Neither the model file nor pretrained weights exist in this environment, so the
hot path is exercised with deterministic synthetic weights of the same
architecture (torchvision ``fcn_resnet50`` / ``fcn_resnet101``, 21 classes,
``replace_stride_with_dilation=[False, True, True]``, BN folded into the convs).

Blob layout ("INFURW01", little endian) -- consumed by ``infur_model_load_blob``
(include/infur_hip.h), by the C oracle (oracle/infur_oracle.c) and by the torch
oracle (oracle/infur_oracle.py):

    0   char[8]  magic "INFURW01"
    8   u32      depth (50 | 101)
    12  u32      num_classes
    16  u32      has_aux
    20  u32      n_convs
    24  u32      input kind: 0 = Float image input (RGB planes, torchvision normalisation: predict_onnx.rs:126-137),
                 1 = Uint8 image input (the frame's bytes as they are, BGR kept: predict_onnx.rs:116-122,296-301)
    28  u32      reserved
    32  n_convs x 80-byte entries: char name[40]; u32 cout, cin, kh, kw;
                 u64 w_off; u64 b_off; u8 reserved[8]
    ... f32 data, 64-byte aligned: weights OIHW (BN already folded), bias [cout]

Convs appear in graph order: backbone.conv1; per layer/block conv1, conv2, conv3
(+ downsample.0 in block 0); classifier.0, classifier.4; aux_classifier.0,
aux_classifier.4.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict, Iterator, List, Tuple

import numpy as np

MAGIC = b"INFURW01"
HDR = 32
ENTRY = 80
NUM_CLASSES = 21
DEFAULT_SEED = 0x1F0A2026

LAYER_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


@dataclass(frozen=True)
class ConvSpec:
    """One (BN-folded) convolution of the graph, in execution order."""

    name: str
    cout: int
    cin: int
    k: int
    stride: int
    pad: int
    dil: int
    relu: bool
    has_bn: bool  # synthetic generator folds a BN into it
    role: str  # stem | conv1 | conv2 | conv3 | down | head3 | cls | aux3 | auxcls


def graph(depth: int = 50, num_classes: int = NUM_CLASSES, aux: bool = True) -> List[ConvSpec]:
    """Conv list of torchvision's fcn_resnet{depth} (Bottleneck v1.5, output stride 8)."""
    blocks = LAYER_BLOCKS[depth]
    out: List[ConvSpec] = [ConvSpec("backbone.conv1", 64, 3, 7, 2, 3, 1, True, True, "stem")]
    inplanes, dilation = 64, 1
    for li, nb in enumerate(blocks):
        planes = 64 << li
        stride = 1 if li == 0 else 2
        prev_dil = dilation
        if li >= 2:  # replace_stride_with_dilation
            dilation *= stride
            stride = 1
        for b in range(nb):
            bs = stride if b == 0 else 1
            bd = prev_dil if b == 0 else dilation
            p = f"backbone.layer{li + 1}.{b}"
            out.append(ConvSpec(f"{p}.conv1", planes, inplanes, 1, 1, 0, 1, True, True, "conv1"))
            out.append(ConvSpec(f"{p}.conv2", planes, planes, 3, bs, bd, bd, True, True, "conv2"))
            out.append(ConvSpec(f"{p}.conv3", planes * 4, planes, 1, 1, 0, 1, True, True, "conv3"))
            if b == 0:
                out.append(
                    ConvSpec(f"{p}.downsample.0", planes * 4, inplanes, 1, bs, 0, 1, False, True, "down")
                )
            inplanes = planes * 4
    out.append(ConvSpec("classifier.0", 512, 2048, 3, 1, 1, 1, True, True, "head3"))
    out.append(ConvSpec("classifier.4", num_classes, 512, 1, 1, 0, 1, False, False, "cls"))
    if aux:
        out.append(ConvSpec("aux_classifier.0", 256, 1024, 3, 1, 1, 1, True, True, "aux3"))
        out.append(ConvSpec("aux_classifier.4", num_classes, 256, 1, 1, 0, 1, False, False, "auxcls"))
    return out


def conv_out(n: int, k: int, s: int, p: int, d: int) -> int:
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def lowres_dims(h: int, w: int) -> Tuple[int, int]:
    """Output-stride-8 feature map size for an h x w input (stem, maxpool, layer2 stride)."""
    for k, s, p in ((7, 2, 3), (3, 2, 1), (3, 2, 1)):
        h, w = conv_out(h, k, s, p, 1), conv_out(w, k, s, p, 1)
    return h, w


def plan(h: int, w: int, depth: int = 50, aux: bool = True) -> List[Tuple[ConvSpec, int, int, int, int]]:
    """Execution plan: (spec, in_h, in_w, out_h, out_w) for every conv at an h x w network input."""
    out = []
    cur = (h, w)
    blk_in = cur
    l3 = cur
    for c in graph(depth, aux=aux):
        if c.role == "conv1":
            blk_in = cur
        ih, iw = blk_in if c.role == "down" else (l3 if c.role == "aux3" else cur)
        oh, ow = conv_out(ih, c.k, c.stride, c.pad, c.dil), conv_out(iw, c.k, c.stride, c.pad, c.dil)
        out.append((c, ih, iw, oh, ow))
        if c.role != "down":
            cur = (oh, ow)
        if c.role == "stem":  # maxpool 3x3/2 pad 1 follows
            cur = (conv_out(oh, 3, 2, 1, 1), conv_out(ow, 3, 2, 1, 1))
        if c.name.startswith("backbone.layer3.") and c.role == "conv3":
            l3 = cur
    return out


def conv_flops(h: int, w: int, depth: int = 50, aux: bool = True) -> Dict[str, object]:
    """Algorithmic FLOPs (2 x MAC) of the conv layers for an h x w network input."""
    per = [(c.name, 2.0 * oh * ow * c.cout * c.cin * c.k * c.k, c.k) for c, _, _, oh, ow in plan(h, w, depth, aux)]
    return {
        "total": sum(f for _, f, _ in per),
        "conv3x3": sum(f for _, f, k in per if k == 3),
        "per_conv": [(n, f) for n, f, _ in per],
    }


# --------------------------------------------------------------------------- #
# counter-based PRNG (splitmix64 finaliser over seed + index) -> U[0,1) f32
# --------------------------------------------------------------------------- #
_M64 = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 output function on uint64 counters."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """n floats in [0,1) from the counter (seed, stream, index): 24 random bits each."""
    base = (seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & _M64
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(base)
    bits = splitmix64(ctr) >> np.uint64(40)
    return bits.astype(np.float32) * np.float32(1.0 / (1 << 24))


def synth_frame(h: int, w: int, index: int = 0, seed: int = DEFAULT_SEED) -> np.ndarray:
    """Deterministic packed-BGR u8 frame [h,w,3]: PRNG noise blended 50/50 with an x/y gradient."""
    n = h * w * 3
    noise = (uniform01(seed ^ 0x5EED, 1000 + index, n) * 256.0).astype(np.uint8).reshape(h, w, 3)
    yy = (np.arange(h, dtype=np.uint32)[:, None] * 255 // max(h - 1, 1)).astype(np.uint16)
    xx = (np.arange(w, dtype=np.uint32)[None, :] * 255 // max(w - 1, 1)).astype(np.uint16)
    grad = np.stack(
        [np.broadcast_to(xx, (h, w)), np.broadcast_to(yy, (h, w)), (np.broadcast_to(xx, (h, w)) + yy) // 2],
        axis=-1,
    ).astype(np.uint16)
    shift = (index * 37) & 0xFF
    grad = (grad + shift) & 0xFF
    return ((noise.astype(np.uint16) + grad) // 2).astype(np.uint8)


def synth_tensors(
    depth: int = 50, num_classes: int = NUM_CLASSES, aux: bool = True, seed: int = DEFAULT_SEED
) -> Iterator[Tuple[ConvSpec, np.ndarray, np.ndarray]]:
    """Yield (spec, folded weight OIHW f32, folded bias f32) with seeded synthetic values.

    conv W ~ U(-a, a), a = sqrt(6 / fan_in); BN gamma in [0.5,1.5], beta/mean in
    [-0.1,0.1], var in [0.5,1.5], eps 1e-5, folded in float64 then rounded to f32:
    W' = W*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps).  The residual
    branch's last conv (conv3) gets gamma scaled by 0.25 so activations stay O(1)
    through 16/33 blocks; classifier biases are distinct so exact ties are rare.
    """
    for idx, c in enumerate(graph(depth, num_classes, aux)):
        fan_in = c.cin * c.k * c.k
        a = np.sqrt(6.0 / fan_in)
        n = c.cout * fan_in
        w = ((uniform01(seed, 4 * idx + 0, n).astype(np.float64) * 2.0 - 1.0) * a).reshape(
            c.cout, c.cin, c.k, c.k
        )
        if c.has_bn:
            u = uniform01(seed, 4 * idx + 1, 4 * c.cout).astype(np.float64).reshape(4, c.cout)
            gamma = 0.5 + u[0]
            beta = (u[1] - 0.5) * 0.2
            mean = (u[2] - 0.5) * 0.2
            var = 0.5 + u[3]
            if c.role == "conv3":
                gamma = gamma * 0.25
            s = gamma / np.sqrt(var + 1e-5)
            w = w * s[:, None, None, None]
            b = beta - mean * s
        else:
            u = uniform01(seed, 4 * idx + 1, c.cout).astype(np.float64)
            b = (u - 0.5) * 0.5 + 0.01 * np.arange(c.cout)
            w = w * 0.05  # keep max logits around 1: alpha bytes cover (0,255) and saturate
        yield c, w.astype(np.float32), b.astype(np.float32)


def pack_blob(tensors, depth: int, num_classes: int, aux: bool, input_u8: bool = False) -> bytes:
    """Serialise (spec-or-name, W OIHW f32, b f32) triples into the INFURW01 blob."""
    items = []
    for spec, w, b in tensors:
        name = spec if isinstance(spec, str) else spec.name
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        assert w.ndim == 4 and b.shape == (w.shape[0],), name
        items.append((name, w, b))
    n = len(items)
    off = HDR + n * ENTRY
    off = (off + 63) & ~63
    table = bytearray()
    chunks = []
    for name, w, b in items:
        w_off = off
        off = (off + w.nbytes + 63) & ~63
        b_off = off
        off = (off + b.nbytes + 63) & ~63
        nb = name.encode()
        assert len(nb) < 40, name
        table += nb.ljust(40, b"\0")
        table += struct.pack("<4I2Q8x", w.shape[0], w.shape[1], w.shape[2], w.shape[3], w_off, b_off)
        chunks.append((w_off, w, b_off, b))
    buf = bytearray(off)
    buf[0:HDR] = MAGIC + struct.pack("<6I", depth, num_classes, 1 if aux else 0, n, 1 if input_u8 else 0, 0)
    buf[HDR : HDR + len(table)] = table
    for w_off, w, b_off, b in chunks:
        buf[w_off : w_off + w.nbytes] = w.tobytes()
        buf[b_off : b_off + b.nbytes] = b.tobytes()
    return bytes(buf)


def synth_blob(depth: int = 50, num_classes: int = NUM_CLASSES, aux: bool = True, seed: int = DEFAULT_SEED,
               input_u8: bool = False) -> bytes:
    """``input_u8``: a model that declares a Uint8 image input -- the stem sees the raw BGR bytes (0..255), so its
    synthetic weights are scaled by 1/64 to keep the activations in the range the Float-input model has."""
    if not input_u8:
        return pack_blob(synth_tensors(depth, num_classes, aux, seed), depth, num_classes, aux)
    ts = [(c, w * np.float32(1.0 / 64.0) if c.name == "backbone.conv1" else w, b) for c, w, b in synth_tensors(depth, num_classes, aux, seed)]
    return pack_blob(ts, depth, num_classes, aux, input_u8=True)


def unpack_blob(blob: bytes):
    """Parse a blob -> (meta dict, [(name, W OIHW f32 view, b f32 view), ...])."""
    if len(blob) < HDR or blob[:8] != MAGIC:
        raise ValueError("not an INFURW01 weight blob")
    depth, ncls, aux, n, kind, _ = struct.unpack_from("<6I", blob, 8)
    out = []
    for i in range(n):
        e = HDR + i * ENTRY
        name = blob[e : e + 40].split(b"\0", 1)[0].decode()
        cout, cin, kh, kw, w_off, b_off = struct.unpack_from("<4I2Q", blob, e + 40)
        w = np.frombuffer(blob, dtype=np.float32, count=cout * cin * kh * kw, offset=w_off).reshape(cout, cin, kh, kw)
        b = np.frombuffer(blob, dtype=np.float32, count=cout, offset=b_off)
        out.append((name, w, b))
    return {"depth": depth, "num_classes": ncls, "aux": bool(aux), "n_convs": n, "input_u8": kind == 1}, out


# --------------------------------------------------------------------------- #
# quantised models: the "INFURQ01" blob
# --------------------------------------------------------------------------- #
# The reference's own tests load `fcn-resnet50-12-int8.onnx` (infur-test-gen/build.rs:88-93; predict_onnx.rs:357-381): the
# QOperator form of the same network -- QuantizeLinear on the image, QLinearConv (u8 activations, s8 weights, i32 bias,
# per-output-channel weight scales) for every convolution with the ReLU folded into the clamp at the zero point, a u8
# MaxPool, QLinearAdd (com.microsoft) for the residual sums, DequantizeLinear in front of the two Resize nodes.  This blob
# carries exactly what those nodes carry (little endian):
#
#     0   char[8]  magic "INFURQ01"
#     8   u32 depth, u32 num_classes, u32 has_aux, u32 n_convs, u32 n_adds, u32 flags (bit 0: the file's Resize runs on the u8
#         logits and DequantizeLinear follows it -- QLinearConv -> Resize -> DequantizeLinear -- instead of the other way round)
#     32  n_convs x 96-byte entries: char name[40]; u32 cout, cin, kh, kw; f32 x_scale; i32 x_zp; f32 y_scale; i32 y_zp;
#                                    u64 w_off (s8 OIHW); u64 ws_off (f32 [cout] weight scales); u64 b_off (i32 [cout])
#     ..  n_adds x 24-byte entries:  f32 a_scale; i32 a_zp; f32 b_scale; i32 b_zp; f32 c_scale; i32 c_zp
#                                    (A = the block's conv3 output, B = the identity / downsample branch, C = the block output)
#     ..  data, 64-byte aligned
# conv 0's (x_scale, x_zp) quantise the normalised image; the (y_scale, y_zp) of classifier.4 / aux_classifier.4 dequantise
# the logits.  Arithmetic (ONNX operator definitions; every step is one IEEE f32 operation, round = to nearest even):
#     QuantizeLinear   q = sat_u8(round(x / s) + zp)
#     QLinearConv      acc = sum (x - x_zp) * w + b   (i32, exact);   y = sat_u8(round(f32(acc) * M[o]) + y_zp),
#                      M[o] = (x_scale * w_scale[o]) / y_scale;  padding contributes x_zp, i.e. nothing
#     QLinearAdd       c = sat_u8(round(f32(a - a_zp) * (a_s / c_s) + f32(b - b_zp) * (b_s / c_s)) + c_zp)
#     DequantizeLinear x = f32(q - zp) * s
QMAGIC = b"INFURQ01"
QENTRY, QADD = 96, 24


@dataclass
class QConv:
    name: str
    w: np.ndarray        # s8 [cout, cin, kh, kw]
    w_scale: np.ndarray  # f32 [cout]
    bias: np.ndarray     # i32 [cout]
    x_scale: float
    x_zp: int
    y_scale: float
    y_zp: int


@dataclass
class QAdd:
    a_scale: float
    a_zp: int
    b_scale: float
    b_zp: int
    c_scale: float
    c_zp: int


def pack_qblob(convs: List[QConv], adds: List[QAdd], depth: int, num_classes: int, aux: bool, resize_u8: bool = False) -> bytes:
    n, na = len(convs), len(adds)
    off = (HDR + n * QENTRY + na * QADD + 63) & ~63
    table, chunks = bytearray(), []
    for c in convs:
        w = np.ascontiguousarray(c.w, dtype=np.int8)
        ws = np.ascontiguousarray(c.w_scale, dtype=np.float32)
        b = np.ascontiguousarray(c.bias, dtype=np.int32)
        assert w.ndim == 4 and ws.shape == (w.shape[0],) and b.shape == (w.shape[0],), c.name
        offs = []
        for arr in (w, ws, b):
            offs.append(off)
            chunks.append((off, arr))
            off = (off + arr.nbytes + 63) & ~63
        nb = c.name.encode()
        assert len(nb) < 40
        table += nb.ljust(40, b"\0") + struct.pack("<4IfifI3Q", *w.shape, np.float32(c.x_scale), int(c.x_zp), np.float32(c.y_scale),
                                                   int(c.y_zp) & 0xFFFFFFFF, *offs)
    for a in adds:
        table += struct.pack("<fififi", np.float32(a.a_scale), int(a.a_zp), np.float32(a.b_scale), int(a.b_zp), np.float32(a.c_scale), int(a.c_zp))
    buf = bytearray(off)
    buf[0:HDR] = QMAGIC + struct.pack("<6I", depth, num_classes, 1 if aux else 0, n, na, 1 if resize_u8 else 0)
    buf[HDR:HDR + len(table)] = table
    for o, arr in chunks:
        buf[o:o + arr.nbytes] = arr.tobytes()
    return bytes(buf)


def unpack_qblob(blob: bytes):
    if len(blob) < HDR or blob[:8] != QMAGIC:
        raise ValueError("not an INFURQ01 quantised weight blob")
    depth, ncls, aux, n, na, flags = struct.unpack_from("<6I", blob, 8)
    convs, adds = [], []
    for i in range(n):
        e = HDR + i * QENTRY
        name = blob[e:e + 40].split(b"\0", 1)[0].decode()
        cout, cin, kh, kw, xs, xz, ys, yz, w_off, ws_off, b_off = struct.unpack_from("<4Ififi3Q", blob, e + 40)
        convs.append(QConv(name, np.frombuffer(blob, np.int8, cout * cin * kh * kw, w_off).reshape(cout, cin, kh, kw),
                           np.frombuffer(blob, np.float32, cout, ws_off), np.frombuffer(blob, np.int32, cout, b_off), xs, xz, ys, yz))
    for i in range(na):
        adds.append(QAdd(*struct.unpack_from("<fififi", blob, HDR + n * QENTRY + i * QADD)))
    meta = {"depth": depth, "num_classes": ncls, "aux": bool(aux), "n_convs": n, "n_adds": na}
    if flags & 1:
        meta["resize_u8"] = True
    return meta, convs, adds
