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


def attr_str(name, v: str):
    return f_str(1, name) + f_bytes(4, v.encode()) + f_varint(20, 3)


def attr_tensor(name, t: bytes):
    return f_str(1, name) + f_bytes(5, t) + f_varint(20, 4)


def tensor_i64(name: str, vals) -> bytes:
    arr = np.asarray(vals, np.int64)
    return f_bytes(1, b"".join(_varint(d) for d in arr.shape)) + f_varint(2, 7) + f_bytes(9, arr.tobytes()) + f_str(8, name)


def tensor_external(name: str, shape, location="weights.bin") -> bytes:
    body = f_bytes(1, b"".join(_varint(d) for d in shape)) + f_varint(2, 1) + f_str(8, name)
    body += f_bytes(13, f_str(1, "location") + f_str(2, location))  # external_data entry
    return body + f_varint(14, 1)  # data_location = EXTERNAL


def fcn_model(tensors, specs, *, unfold_bn=False, raw=True, packed_dims=True, inits_as_inputs=False,
              input_type=1, input_dims=("N", 3, "H", "W"), conv_op="Conv", drop_last=0, rng=None,
              order="topo", dropout=False, identities=False, resize_subgraph=True, resize_mode="linear",
              coord_mode="pytorch_half_pixel", swap_outputs=False, external=None, missing_relu=None, opset=12,
              const_nodes=False, extra_conv=False, front=()):
    """A torchvision-shaped fcn_resnet ModelProto: stem -> Relu -> MaxPool -> bottlenecks (conv1/conv2/conv3 [+ downsample],
    Add, Relu) -> classifier (Conv, Relu, [Dropout], Conv, Resize) and the aux head off layer3, as torch.onnx.export
    writes it (dynamic H/W: Shape -> Gather -> Unsqueeze -> Concat feeding Resize's `sizes`).

    tensors: [(name, W OIHW f32, b f32)] in INFURW01 order; specs: infur_amd.weights.graph(...).
    unfold_bn: Conv(no bias) + BatchNormalization whose folding reproduces (W, b).
    front: nodes between the image input and the stem, in order -- "cast" (Cast to FLOAT), "cast_double" (Cast to DOUBLE),
           "transpose" (perm 0,3,1,2), "transpose_bad" (perm 0,2,3,1), "mul" (an in-graph scaling) -- what a Uint8 and / or
           NHWC input carries (predict_onnx.rs:116-122).
    order: "topo" (export order) | "ds_first" (every downsample conv serialised BEFORE its block's conv1..conv3) |
           "shuffled" (random node order: slot assignment must come from the edges, not from positions).
    """
    rng = rng or np.random.default_rng(0)
    items = list(zip(specs, tensors))
    if drop_last:
        items = items[:-drop_last]
    by_name = {s.name: (s, t) for s, t in items}
    inits, nodes, bn_params = [], [], {}
    uid = [0]

    def fresh(prefix):
        uid[0] += 1
        return f"{prefix}_{uid[0]}"

    def emit(op, ins, outs, attrs=()):
        nodes.append(f_bytes(1, node(op, ins, outs, attrs)))

    def passthrough(t):
        if identities and rng.random() < 0.3:
            o = fresh("id")
            emit("Identity", [t], [o])
            return o
        return t

    def conv(name, x):
        if name not in by_name:
            return None
        s, (_, w, b) = by_name[name]
        attrs = [attr_ints("dilations", [s.dil, s.dil]), attr_int("group", 1), attr_ints("kernel_shape", [s.k, s.k]),
                 attr_ints("pads", [s.pad] * 4), attr_ints("strides", [s.stride, s.stride])]
        wn, bn_, out = f"{name}.weight", f"{name}.bias", fresh("conv")
        if external == name:
            inits.append(tensor_external(wn, w.shape))
            inits.append(tensor(bn_, b, raw, packed_dims))
            emit(conv_op, [x, wn, bn_], [out], attrs)
        elif unfold_bn and s.has_bn:
            gamma = (0.5 + rng.random(s.cout)).astype(np.float32)
            var = (0.5 + rng.random(s.cout)).astype(np.float32)
            mean = ((rng.random(s.cout) - 0.5) * 0.2).astype(np.float32)
            beta = ((rng.random(s.cout) - 0.5) * 0.2).astype(np.float32)
            w_raw = (rng.standard_normal(w.shape) * 0.05).astype(np.float32)
            bn_params[name] = (w_raw, gamma, beta, mean, var)
            inits.append(tensor(wn, w_raw, raw, packed_dims))
            emit(conv_op, [x, wn], [out], attrs)
            names = [f"{name}.bn.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
            for nm, arr in zip(names, (gamma, beta, mean, var)):
                inits.append(tensor(nm, arr, raw, packed_dims))
            emit("BatchNormalization", [out] + names, [out + "_bn"], [attr_float("epsilon", 1e-5)])
            out = out + "_bn"
        else:
            if const_nodes and name == "backbone.layer1.0.conv1":  # weights as a Constant node instead of an initializer
                emit("Constant", [], [wn], [attr_tensor("value", tensor(wn, w, raw, packed_dims))])
            else:
                inits.append(tensor(wn, w, raw, packed_dims))
            inits.append(tensor(bn_, b, raw, packed_dims))
            emit(conv_op, [x, wn, bn_], [out], attrs)
        return out

    def relu(x, tag):
        if missing_relu == tag:
            return x
        o = fresh("relu")
        emit("Relu", [x], [o])
        return passthrough(o)

    # ---- backbone ----
    x = "input"
    for kind in front:
        o = fresh("front")
        if kind == "cast":
            emit("Cast", [x], [o], [attr_int("to", 1)])
        elif kind == "cast_double":
            emit("Cast", [x], [o], [attr_int("to", 11)])
        elif kind == "transpose":
            emit("Transpose", [x], [o], [attr_ints("perm", [0, 3, 1, 2])])
        elif kind == "transpose_bad":
            emit("Transpose", [x], [o], [attr_ints("perm", [0, 2, 3, 1])])
        elif kind == "mul":
            inits.append(tensor("front.scale", np.asarray([1.0 / 255.0], np.float32)))
            emit("Mul", [x, "front.scale"], [o])
        else:
            raise ValueError(kind)
        x = o
    x = relu(conv("backbone.conv1", x), "stem")
    o = fresh("pool")
    emit("MaxPool", [x], [o], [attr_ints("kernel_shape", [3, 3]), attr_ints("pads", [1, 1, 1, 1]), attr_ints("strides", [2, 2]),
                                attr_int("ceil_mode", 0)])
    x = o
    l3 = None
    blocks = sorted({s.name.rsplit(".", 1)[0] for s, _ in items if s.role == "conv1"}, key=lambda p: [int(v) if v.isdigit() else v for v in p.replace("layer", "layer.").split(".")])
    for p in blocks:
        start = len(nodes)
        ds_nodes = (0, 0)
        t = relu(conv(p + ".conv1", x), p + ".conv1")
        t = relu(conv(p + ".conv2", t), p + ".conv2")
        t = conv(p + ".conv3", t)
        idt = x
        if (p + ".downsample.0") in by_name:
            d0 = len(nodes)
            idt = conv(p + ".downsample.0", x)
            ds_nodes = (d0, len(nodes))
        if order == "ds_first" and ds_nodes[1] > ds_nodes[0]:
            blk = nodes[start:]
            ds = blk[ds_nodes[0] - start:ds_nodes[1] - start]
            rest = blk[:ds_nodes[0] - start] + blk[ds_nodes[1] - start:]
            nodes[start:] = ds + rest
        if t is None:
            break
        a = fresh("add")
        emit("Add", [t, idt] if rng.random() < 0.5 else [idt, t], [a])
        x = relu(a, p + ".out")
        if p.startswith("backbone.layer3."):
            l3 = x

    def head(prefix, feat, out_name):
        h = conv(prefix + ".0", feat)
        if h is None:
            return
        h = relu(h, prefix + ".0")
        if dropout:
            d, m = fresh("drop"), fresh("mask")
            emit("Dropout", [h], [d, m])
            h = d
        lo = conv(prefix + ".4", h)
        if lo is None:
            return
        if resize_subgraph:  # sizes = concat(shape(lo)[:2], shape(input)[2:]) built from Shape/Gather/Unsqueeze/Concat
            shp, sl = fresh("shape"), fresh("slice")
            emit("Shape", [lo], [shp])
            for nm, v in ((sl + "_s", [0]), (sl + "_e", [2]), (sl + "_a", [0])):
                emit("Constant", [], [nm], [attr_tensor("value", tensor_i64(nm, v))])
            emit("Slice", [shp, sl + "_s", sl + "_e", sl + "_a"], [sl])
            dims = []
            for ax in (2, 3):
                ishp, gi, g, u = fresh("ishape"), fresh("gidx"), fresh("gather"), fresh("unsq")
                emit("Shape", ["input"], [ishp])
                emit("Constant", [], [gi], [attr_tensor("value", tensor_i64(gi, ax))])
                emit("Gather", [ishp, gi], [g], [attr_int("axis", 0)])
                emit("Unsqueeze", [g], [u], [attr_ints("axes", [0])])
                dims.append(u)
            cc, cast, sizes = fresh("concat"), fresh("cast"), fresh("sizes")
            emit("Concat", dims, [cc], [attr_int("axis", 0)])
            emit("Cast", [cc], [cast], [attr_int("to", 7)])
            emit("Concat", [sl, cast], [sizes], [attr_int("axis", 0)])
            roi, scales = fresh("roi"), fresh("scales")
            emit("Constant", [], [roi], [attr_tensor("value", tensor(roi, np.zeros((0,), np.float32)))])
            emit("Constant", [], [scales], [attr_tensor("value", tensor(scales, np.zeros((0,), np.float32)))])
            rin = [lo, roi, scales, sizes]
        else:
            inits.append(tensor(out_name + ".scales", np.array([1, 1, 8, 8], np.float32)))
            rin = [lo, "", out_name + ".scales"]
        attrs = [attr_str("mode", resize_mode)]
        if coord_mode is not None:
            attrs.append(attr_str("coordinate_transformation_mode", coord_mode))
        emit("Resize" if opset >= 10 else "Upsample", rin, [out_name], attrs)

    onames = ("aux", "out") if swap_outputs else ("out", "aux")
    head("classifier", x, onames[0])
    head("aux_classifier", l3, onames[1])
    if extra_conv:  # a conv that is not part of FCN-ResNet hanging off the graph
        w = np.zeros((4, 21, 1, 1), np.float32)
        inits.append(tensor("extra.weight", w))
        emit("Conv", ["out", "extra.weight"], ["extra_out"], [attr_ints("kernel_shape", [1, 1])])

    if order == "shuffled":
        perm = rng.permutation(len(nodes))
        nodes = [nodes[i] for i in perm]
    g = b"".join(nodes)
    g += f_str(2, "torch-jit-export")
    g += b"".join(f_bytes(5, t) for t in inits)
    g += f_bytes(11, value_info("input", input_type, input_dims))
    if inits_as_inputs:  # IR < 4 exporters list every initializer as a graph input too
        g = f_bytes(11, value_info("backbone.conv1.weight", 1, (64, 3, 7, 7))) + g
        g += f_bytes(11, value_info("classifier.4.bias", 1, (21,)))
    g += f_bytes(12, value_info("out", 1, ("N", 21, "H", "W")))
    g += f_bytes(12, value_info("aux", 1, ("N", 21, "H", "W")))
    model = f_varint(1, 6) + f_str(2, "pytorch") + f_bytes(7, g) + f_bytes(8, f_str(1, "") + f_varint(2, opset))
    return model, bn_params


def fcn_qmodel(convs, adds, specs, *, rng=None, order="topo", per_tensor_scale=(), relu_after=(), w_zp=0, w_dtype=3, swap_add=None,
               input_type=1, coord_mode="pytorch_half_pixel", drop_last=0, stem_scale=None, dq_scale=None, no_bias=(), vector_wzp=False,
               pad_zp_conv=None, extra_qconv=False, shift_weights=True, resize_u8=False, resize_subgraph=False, qdq=False, qdq_share_dq=False, float_add=False,
               qdq_bias_scale_factor=None, qdq_bias_zp=None):
    """The QOperator int8 form of the same network, as ONNX Runtime's static quantisation writes it (the shape of
    `fcn-resnet50-12-int8.onnx`, the file the reference's tests load: predict_onnx.rs:357-381):
    QuantizeLinear -> QLinearConv (ReLU folded) -> MaxPool (u8) -> bottlenecks (QLinearConv x3 [+ downsample], com.microsoft
    QLinearAdd) -> heads (QLinearConv, QLinearConv, DequantizeLinear, Resize).

    convs / adds: infur_amd.weights.QConv / QAdd lists in blob order; specs: infur_amd.weights.graph(...).
    per_tensor_scale: conv names whose w_scale is written as a scalar (all channels must then share it);
    relu_after: conv names followed by an explicit (redundant) Relu on the u8 tensor; swap_add: True / False / None (random)
    operand order of QLinearAdd; the remaining switches produce the files the reader must REJECT."""
    rng = rng or np.random.default_rng(0)
    items = list(zip(specs, convs))
    if drop_last:
        items = items[:-drop_last]
    by_name = {s.name: (s, c) for s, c in items}
    inits, nodes = [], []
    uid = [0]

    def fresh(prefix):
        uid[0] += 1
        return f"{prefix}_{uid[0]}"

    def emit(op, ins, outs, attrs=(), domain=None):
        b = node(op, ins, outs, attrs)
        if domain:
            b += f_str(7, domain)
        nodes.append(f_bytes(1, b))

    def scalar_f(name, v):
        inits.append(tensor(name, np.asarray(v, np.float32).reshape(())))
        return name

    def scalar_u8(name, v):
        inits.append(tensor(name, np.asarray(v, np.uint8).reshape(()), dtype=2))
        return name

    qparams, shared_dq = {}, {}

    def dequant(q, sc, zp):
        """QDQ form: the float view of a quantised tensor -- one DequantizeLinear per consumer, or a shared one"""
        if qdq_share_dq and q in shared_dq:
            return shared_dq[q]
        f = fresh("dq")
        emit("DequantizeLinear", [q, sc, zp], [f])
        shared_dq[q] = f
        return f

    def qconv(name, x):
        if name not in by_name:
            return None
        s, c = by_name[name]
        xs, xz = scalar_f(name + ".x_scale", stem_scale if (stem_scale and s.role == "stem") else c.x_scale), scalar_u8(name + ".x_zp", pad_zp_conv[1] if pad_zp_conv and pad_zp_conv[0] == name else c.x_zp)
        ys, yz = scalar_f(name + ".y_scale", c.y_scale), scalar_u8(name + ".y_zp", c.y_zp)
        # the file stores w + w_zp (UINT8 or INT8): the reader re-centres it to the blob's s8 weights with zero point 0
        w = np.ascontiguousarray(c.w, np.int8)
        stored = w.astype(np.int16) + (w_zp if shift_weights else 0)
        if w_dtype == 2:
            inits.append(tensor(name + ".weight", np.clip(stored, 0, 255).astype(np.uint8), dtype=2))
        else:
            inits.append(tensor(name + ".weight", np.clip(stored, -128, 127).astype(np.int8), dtype=3))
        if name in per_tensor_scale:
            inits.append(tensor(name + ".w_scale", np.asarray(c.w_scale[0], np.float32).reshape(())))
        else:
            inits.append(tensor(name + ".w_scale", np.asarray(c.w_scale, np.float32)))
        zp = np.full((s.cout,) if vector_wzp else (), w_zp, np.uint8 if w_dtype == 2 else np.int8)
        inits.append(tensor(name + ".w_zp", zp, dtype=2 if w_dtype == 2 else 3))
        ins = [x, xs, xz, name + ".weight", name + ".w_scale", name + ".w_zp", ys, yz]
        if name not in no_bias:
            inits.append(tensor(name + ".bias", np.asarray(c.bias, np.int32), dtype=6))
            ins.append(name + ".bias")
        out = fresh("qconv")
        cattrs = [attr_ints("dilations", [s.dil, s.dil]), attr_int("group", 1), attr_ints("kernel_shape", [s.k, s.k]),
                  attr_ints("pads", [s.pad] * 4), attr_ints("strides", [s.stride, s.stride])]
        if qdq:  # DQ(x), DQ(w), DQ(b) -> Conv -> [Relu ->] Q: what ONNX Runtime fuses back into QLinearConv at session creation
            xf = dequant(x, xs, xz)
            wf = fresh("w_dq")
            emit("DequantizeLinear", [name + ".weight", name + ".w_scale", name + ".w_zp"], [wf], [attr_int("axis", 0)])
            cin = [xf, wf]
            if name not in no_bias:
                bsc = (np.float32(c.x_scale) * np.asarray(c.w_scale, np.float32)).astype(np.float32)
                if qdq_bias_scale_factor and name == qdq_bias_scale_factor[0]:  # a file the reader must reject
                    bsc = (bsc * np.float32(qdq_bias_scale_factor[1])).astype(np.float32)
                inits.append(tensor(name + ".b_scale", bsc))
                bf = fresh("b_dq")
                bins = [name + ".bias", name + ".b_scale"]
                if qdq_bias_zp and name == qdq_bias_zp[0]:
                    inits.append(tensor(name + ".b_zp", np.full(s.cout, qdq_bias_zp[1], np.int32), dtype=6))
                    bins.append(name + ".b_zp")
                emit("DequantizeLinear", bins, [bf], [attr_int("axis", 0)])
                cin.append(bf)
            y = fresh("conv")
            emit("Conv", cin, [y], cattrs)
            if (c.y_zp == 0 and s.relu) or name in relu_after:
                r = fresh("relu")
                emit("Relu", [y], [r])
                y = r
            emit("QuantizeLinear", [y, ys, yz], [out])
            qparams[out] = (ys, yz)
            return out
        emit("QLinearConv", ins, [out], cattrs)
        if name in relu_after:
            r = fresh("relu")
            emit("Relu", [out], [r])
            out = r
        return out

    stem_c = by_name["backbone.conv1"][1]
    q = fresh("q")
    emit("QuantizeLinear", ["input", scalar_f("input.scale", stem_c.x_scale), scalar_u8("input.zp", stem_c.x_zp)], [q])
    x = qconv("backbone.conv1", q)
    o = fresh("pool")
    pattrs = [attr_ints("kernel_shape", [3, 3]), attr_ints("pads", [1, 1, 1, 1]), attr_ints("strides", [2, 2]), attr_int("ceil_mode", 0)]
    if qdq:  # DQ -> MaxPool -> Q with the same parameters
        pf = fresh("poolf")
        emit("MaxPool", [dequant(x, "backbone.conv1.y_scale", "backbone.conv1.y_zp")], [pf], pattrs)
        emit("QuantizeLinear", [pf, "backbone.conv1.y_scale", "backbone.conv1.y_zp"], [o])
    else:
        emit("MaxPool", [x], [o], pattrs)
    x = o
    l3 = None
    blocks = sorted({s.name.rsplit(".", 1)[0] for s, _ in items if s.role == "conv1"}, key=lambda p: [int(v) if v.isdigit() else v for v in p.replace("layer", "layer.").split(".")])
    add_it = iter(adds)
    for p in blocks:
        start = len(nodes)
        t = qconv(p + ".conv1", x)
        t = qconv(p + ".conv2", t)
        t = qconv(p + ".conv3", t)
        idt, ds_nodes = x, (0, 0)
        if (p + ".downsample.0") in by_name:
            d0 = len(nodes)
            idt = qconv(p + ".downsample.0", x)
            ds_nodes = (d0, len(nodes))
        if order == "ds_first" and ds_nodes[1] > ds_nodes[0]:
            blk = nodes[start:]
            nodes[start:] = blk[ds_nodes[0] - start:ds_nodes[1] - start] + blk[:ds_nodes[0] - start] + blk[ds_nodes[1] - start:]
        if t is None:
            break
        a = next(add_it)
        A = [t, scalar_f(p + ".add.a_scale", a.a_scale), scalar_u8(p + ".add.a_zp", a.a_zp)]
        B = [idt, scalar_f(p + ".add.b_scale", a.b_scale), scalar_u8(p + ".add.b_zp", a.b_zp)]
        sw = rng.random() < 0.5 if swap_add is None else swap_add
        o = fresh("qadd")
        if qdq or float_add:  # DQ(a), DQ(b) -> Add -> [Relu ->] Q  (float_add: between QLinearConv nodes, pre-QLinearAdd quantisers)
            fa, fb = dequant(*A), dequant(*B)
            y = fresh("add")
            emit("Add", [fb, fa] if sw else [fa, fb], [y])
            if a.c_zp == 0:
                r = fresh("relu")
                emit("Relu", [y], [r])
                y = r
            emit("QuantizeLinear", [y, scalar_f(p + ".add.c_scale", a.c_scale), scalar_u8(p + ".add.c_zp", a.c_zp)], [o])
        else:
            emit("QLinearAdd", (B + A if sw else A + B) + [scalar_f(p + ".add.c_scale", a.c_scale), scalar_u8(p + ".add.c_zp", a.c_zp)], [o], domain="com.microsoft")
        x = o
        if p.startswith("backbone.layer3."):
            l3 = x

    def head(prefix, feat, out_name):
        h = qconv(prefix + ".0", feat)
        lo = qconv(prefix + ".4", h) if h is not None else None
        if lo is None:
            return
        c = by_name[prefix + ".4"][1]
        dq = fresh("dq")
        attrs = [attr_str("mode", "linear")]
        if coord_mode is not None:
            attrs.append(attr_str("coordinate_transformation_mode", coord_mode))
        dq_in = [scalar_f(prefix + ".dq.scale", dq_scale or c.y_scale), scalar_u8(prefix + ".dq.zp", c.y_zp)]
        rs = resize_u8 if isinstance(resize_u8, bool) else resize_u8[0 if prefix == "classifier" else 1]

        def resize_inputs(x):
            if not resize_subgraph:
                inits.append(tensor(out_name + ".scales", np.array([1, 1, 8, 8], np.float32)))
                return [x, "", out_name + ".scales"]
            # the exporter's dynamic-size form, kept in float by the quantiser: sizes = concat(shape(x)[:2], shape(input)[2:])
            shp, sl = fresh("shape"), fresh("slice")
            emit("Shape", [x], [shp])
            for nm, v in ((sl + "_s", [0]), (sl + "_e", [2]), (sl + "_a", [0])):
                emit("Constant", [], [nm], [attr_tensor("value", tensor_i64(nm, v))])
            emit("Slice", [shp, sl + "_s", sl + "_e", sl + "_a"], [sl])
            dims = []
            for ax in (2, 3):
                ishp, gi, gg, u = fresh("ishape"), fresh("gidx"), fresh("gather"), fresh("unsq")
                emit("Shape", ["input"], [ishp])
                emit("Constant", [], [gi], [attr_tensor("value", tensor_i64(gi, ax))])
                emit("Gather", [ishp, gi], [gg], [attr_int("axis", 0)])
                emit("Unsqueeze", [gg], [u], [attr_ints("axes", [0])])
                dims.append(u)
            cc, cast, sizes = fresh("concat"), fresh("cast"), fresh("sizes")
            emit("Concat", dims, [cc], [attr_int("axis", 0)])
            emit("Cast", [cc], [cast], [attr_int("to", 7)])
            emit("Concat", [sl, cast], [sizes], [attr_int("axis", 0)])
            roi, scales = fresh("roi"), fresh("scales")
            emit("Constant", [], [roi], [attr_tensor("value", tensor(roi, np.zeros((0,), np.float32)))])
            emit("Constant", [], [scales], [attr_tensor("value", tensor(scales, np.zeros((0,), np.float32)))])
            return [x, roi, scales, sizes]

        if rs and qdq:  # DQ -> Resize -> Q (same parameters) -> DQ -> output
            rf, rq = fresh("resizef"), fresh("resizeq")
            emit("Resize", resize_inputs(dequant(lo, *dq_in)), [rf], attrs)
            emit("QuantizeLinear", [rf] + dq_in, [rq])
            emit("DequantizeLinear", [rq] + dq_in, [out_name])
        elif rs:  # onnxruntime's QOperator quantiser: Resize stays on the u8 tensor, DequantizeLinear comes last
            emit("Resize", resize_inputs(lo), [dq], attrs)
            emit("DequantizeLinear", [dq] + dq_in, [out_name])
        else:
            emit("DequantizeLinear", [lo] + dq_in, [dq])
            emit("Resize", resize_inputs(dq), [out_name], attrs)

    head("classifier", x, "out")
    head("aux_classifier", l3, "aux")
    if extra_qconv:
        c = by_name["classifier.4"][1]
        inits.append(tensor("extra.weight", np.zeros((4, 21, 1, 1), np.int8), dtype=3))
        inits.append(tensor("extra.w_scale", np.ones((4,), np.float32)))
        inits.append(tensor("extra.w_zp", np.zeros((), np.int8), dtype=3))
        emit("QLinearConv", [l3, "classifier.4.y_scale", "classifier.4.y_zp", "extra.weight", "extra.w_scale", "extra.w_zp", "classifier.4.y_scale", "classifier.4.y_zp"],
             ["extra_out"], [attr_ints("kernel_shape", [1, 1])])
    if order == "shuffled":
        perm = rng.permutation(len(nodes))
        nodes = [nodes[i] for i in perm]
    ncls = by_name["classifier.4"][0].cout if "classifier.4" in by_name else 21
    g = b"".join(nodes) + f_str(2, "onnxruntime-quantized")
    g += b"".join(f_bytes(5, t) for t in inits)
    g += f_bytes(11, value_info("input", input_type, ("N", 3, "H", "W")))
    g += f_bytes(12, value_info("out", 1, ("N", ncls, "H", "W")))
    g += f_bytes(12, value_info("aux", 1, ("N", ncls, "H", "W")))
    return (f_varint(1, 6) + f_str(2, "onnx.quantize") + f_bytes(7, g) + f_bytes(8, f_str(1, "") + f_varint(2, 12)) +
            f_bytes(8, f_str(1, "com.microsoft") + f_varint(2, 1)))
