"""Test helper: a minimal hand-rolled ONNX (protobuf wire format) WRITER.

The build image has neither the `onnx` package nor any model file, so tests fabricate
`fcn_resnet50`-shaped ModelProtos from the synthetic weights to exercise the library's reader
(infur_amd/csrc/onnx_reader.cpp).  Field numbers follow onnx.proto3."""
import struct

import numpy as np


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wt: int) -> bytes:
    return _varint((field << 3) | wt)


def f_varint(field, v):
    return _key(field, 0) + _varint(v)


def f_bytes(field, b: bytes):
    return _key(field, 2) + _varint(len(b)) + b


def f_str(field, s: str):
    return f_bytes(field, s.encode())


def f_float(field, x: float):
    return _key(field, 5) + struct.pack("<f", x)


def tensor(name: str, arr: np.ndarray, raw=True, packed_dims=True, dtype=1) -> bytes:
    arr = np.ascontiguousarray(arr)
    if packed_dims:
        body = f_bytes(1, b"".join(_varint(d) for d in arr.shape))
    else:
        body = b"".join(f_varint(1, d) for d in arr.shape)
    body += f_varint(2, dtype)
    if raw:
        body += f_bytes(9, arr.tobytes())
    else:
        body += f_bytes(4, arr.astype(np.float32).tobytes())  # packed float_data
    return body + f_str(8, name)


def attr_ints(name, vals):
    return f_str(1, name) + b"".join(f_varint(8, v) for v in vals) + f_varint(20, 7)


def attr_int(name, v):
    return f_str(1, name) + f_varint(3, v) + f_varint(20, 2)


def attr_float(name, v):
    return f_str(1, name) + f_float(2, v) + f_varint(20, 1)


def node(op, inputs, outputs, attrs=()):
    b = b"".join(f_str(1, i) for i in inputs) + b"".join(f_str(2, o) for o in outputs) + f_str(4, op)
    return b + b"".join(f_bytes(5, a) for a in attrs)


def value_info(name, elem_type, dims):
    shape = b""
    for d in dims:
        shape += f_bytes(1, f_varint(1, d) if isinstance(d, int) else f_str(2, d))
    tt = f_varint(1, elem_type) + f_bytes(2, shape)
    return f_str(1, name) + f_bytes(2, f_bytes(1, tt))


def fcn_model(tensors, specs, *, unfold_bn=False, raw=True, packed_dims=True, inits_as_inputs=False,
              input_type=1, input_dims=("N", 3, "H", "W"), conv_op="Conv", drop_last=0, rng=None):
    """tensors: [(name, W OIHW f32, b f32)] in graph order; specs: infur_amd.weights.graph(...).
    unfold_bn: emit Conv(no bias) + BatchNormalization whose folding reproduces (W, b)."""
    rng = rng or np.random.default_rng(0)
    g = b""
    inits = []
    prev = "input"
    items = list(zip(specs, tensors))
    if drop_last:
        items = items[:-drop_last]
    bn_params = {}
    for i, (s, (name, w, b)) in enumerate(items):
        attrs = [attr_ints("dilations", [s.dil, s.dil]), attr_int("group", 1), attr_ints("kernel_shape", [s.k, s.k]),
                 attr_ints("pads", [s.pad] * 4), attr_ints("strides", [s.stride, s.stride])]
        wn, bn_, out = f"{name}.weight", f"{name}.bias", f"conv_{i}"
        if unfold_bn and s.has_bn:
            gamma = (0.5 + rng.random(s.cout)).astype(np.float32)
            var = (0.5 + rng.random(s.cout)).astype(np.float32)
            mean = ((rng.random(s.cout) - 0.5) * 0.2).astype(np.float32)
            beta = ((rng.random(s.cout) - 0.5) * 0.2).astype(np.float32)
            w_raw = (rng.standard_normal(w.shape) * 0.05).astype(np.float32)
            bn_params[name] = (w_raw, gamma, beta, mean, var)
            inits.append(tensor(wn, w_raw, raw, packed_dims))
            g += f_bytes(1, node(conv_op, [prev, wn], [out], attrs))
            names = [f"{name}.bn.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
            for nm, arr in zip(names, (gamma, beta, mean, var)):
                inits.append(tensor(nm, arr, raw, packed_dims))
            g += f_bytes(1, node("BatchNormalization", [out] + names, [out + "_bn"], [attr_float("epsilon", 1e-5)]))
            out = out + "_bn"
        else:
            inits.append(tensor(wn, w, raw, packed_dims))
            inits.append(tensor(bn_, b, raw, packed_dims))
            g += f_bytes(1, node(conv_op, [prev, wn, bn_], [out], attrs))
        if s.relu:
            g += f_bytes(1, node("Relu", [out], [out + "_relu"]))
            out = out + "_relu"
        prev = out
    g += f_str(2, "torch-jit-export")
    g += b"".join(f_bytes(5, t) for t in inits)
    g += f_bytes(11, value_info("input", input_type, input_dims))
    if inits_as_inputs:  # IR < 4 exporters list every initializer as a graph input too
        g += f_bytes(11, value_info("backbone.conv1.weight", 1, (64, 3, 7, 7)))
    g += f_bytes(12, value_info("out", 1, ("N", 21, "H", "W")))
    g += f_bytes(12, value_info("aux", 1, ("N", 21, "H", "W")))
    model = f_varint(1, 6) + f_str(2, "pytorch") + f_bytes(7, g) + f_bytes(8, f_str(1, "") + f_varint(2, 12))
    return model, bn_params
