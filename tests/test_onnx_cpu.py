"""ModelCmd::Load("*.onnx") support: the host-only ONNX -> INFURW01 converter (no GPU needed).
Model files are fabricated by tests/onnx_writer.py (no onnx package / zoo file in the image)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import onnx_writer as OW  # noqa: E402

from infur_amd import _lib
from infur_amd import weights as W


def convert(lib, model: bytes):
    blob, n = C.c_void_p(None), C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = lib.infur_onnx_to_blob(model, len(model), C.byref(blob), C.byref(n), err, 512)
    if rc != 0:
        return rc, err.value.decode(), None
    out = C.string_at(blob, n.value)
    lib.infur_buffer_free(blob)
    return 0, "", out


@pytest.fixture(scope="module")
def tensors50(blob50):
    return W.unpack_blob(blob50)[1]


@pytest.mark.parametrize("raw,packed,as_inputs", [(True, True, False), (False, False, True)])
def test_folded_model_roundtrips_bit_exact(lib, blob50, tensors50, raw, packed, as_inputs):
    model, _ = OW.fcn_model(tensors50, W.graph(50), raw=raw, packed_dims=packed, inits_as_inputs=as_inputs)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    meta, got = W.unpack_blob(out)
    assert meta == {"depth": 50, "num_classes": 21, "aux": True, "n_convs": 57, "input_u8": False}
    for (n0, w0, b0), (n1, w1, b1) in zip(tensors50, got):
        assert n0 == n1 and (w0.view(np.uint32) == w1.view(np.uint32)).all() and (b0.view(np.uint32) == b1.view(np.uint32)).all()


def test_unfolded_batchnorm_is_folded(lib, tensors50):
    """Exporters that keep BatchNormalization nodes: W' = W*g/sqrt(v+eps), b' = (0-mean)*g/sqrt(v+eps)+beta."""
    specs = W.graph(50)
    model, bn = OW.fcn_model(tensors50, specs, unfold_bn=True)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    _, got = W.unpack_blob(out)
    for s, (name, w1, b1), (_, w0, b0) in zip(specs, got, tensors50):
        if name in bn:
            w_raw, gamma, beta, mean, var = bn[name]
            sc = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + np.float64(np.float32(1e-5)))  # eps is an f32 attribute
            assert (w1 == (w_raw.astype(np.float64) * sc[:, None, None, None]).astype(np.float32)).all(), name
            assert (b1 == ((0.0 - mean.astype(np.float64)) * sc + beta).astype(np.float32)).all(), name
        else:
            assert (w1 == w0).all() and (b1 == b0).all()


def test_input_checks_carry_the_reference_messages(lib, tensors50):
    """infer_img_pre_proc, predict_onnx.rs:223-265."""
    g = W.graph(50)
    cases = [
        (dict(input_dims=("N", 4, "H", "W")), "couldn't locate model's color input by dimension length 3"),
        (dict(input_dims=(3, "H", "W")), "only 4 dimensions supported got 3"),
        (dict(input_dims=("N", "H", 3, "W")), "color dimension only at NCHW or NHWC but not in position 2 supported"),
        (dict(input_type=7), "only Float (f32) and Uint8 (u8) input supported"),
        # Uint8 / NHWC inputs are valid for the reference (and load: test_uint8_and_nhwc_inputs); the graph must carry them
        (dict(input_type=2), "a Uint8 image input must reach the stem convolution through one Cast to FLOAT"),
        (dict(input_type=2, front=("cast_double",)), "must produce FLOAT"),
        (dict(input_type=2, front=("cast", "mul")), "expected exactly one Conv after the image input"),
        (dict(input_dims=("N", "H", "W", 3)), "an NHWC image input must reach the stem convolution through one Transpose"),
        (dict(input_dims=("N", "H", "W", 3), front=("transpose_bad",)), "perm = [0,3,1,2]"),
        (dict(front=("cast",)), "expected exactly one Conv after the image input"),   # a Float input needs no Cast
        (dict(conv_op="ConvInteger"), "dynamically quantised model"),            # (the static QOperator form loads: test_onnx_quant_cpu.py)
        (dict(conv_op="QLinearConv"), "expected exactly one QuantizeLinear"),    # float tensors under quantised operators
        (dict(drop_last=3), "Conv nodes"),
    ]
    for kw, msg in cases:
        model, _ = OW.fcn_model(tensors50, g, **kw)
        rc, err, _ = convert(lib, model)
        assert rc == _lib.E_MODEL_FORMAT and msg in err, (kw, err)
    rc, err, _ = convert(lib, b"\x08\x06garbage")
    assert rc == _lib.E_MODEL_FORMAT


@pytest.mark.parametrize("kw,u8", [
    (dict(input_type=2, front=("cast",)), True),                                                   # Uint8 NCHW
    (dict(input_type=2, input_dims=("N", "H", "W", 3), front=("transpose", "cast")), True),         # Uint8 NHWC
    (dict(input_type=2, input_dims=("N", "H", "W", 3), front=("cast", "transpose")), True),         # ... Cast first
    (dict(input_dims=("N", "H", "W", 3), front=("transpose",)), False),                             # Float NHWC
    (dict(input_type=2, input_dims=(1, 240, 320, 3), front=("transpose", "cast"), identities=True, order="shuffled"), True),
])
def test_uint8_and_nhwc_inputs(lib, tensors50, kw, u8):
    """infer_img_pre_proc accepts NCHW / NHWC x Float / Uint8 (predict_onnx.rs:223-265).  The layout is the file's own
    business (a Transpose in front of the stem); the element type decides what the session is fed -- normalised RGB
    floats or the frame's BGR bytes (predict_onnx.rs:114-139) -- and travels in the blob header."""
    model, _ = OW.fcn_model(tensors50, W.graph(50), **kw)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    meta, got = W.unpack_blob(out)
    assert meta == {"depth": 50, "num_classes": 21, "aux": True, "n_convs": 57, "input_u8": u8}
    for (n0, w0, b0), (n1, w1, b1) in zip(tensors50, got):
        assert n0 == n1 and (w0.view(np.uint32) == w1.view(np.uint32)).all() and (b0.view(np.uint32) == b1.view(np.uint32)).all()


def test_wrong_conv_attributes_are_rejected(lib, tensors50):
    import dataclasses

    g = W.graph(50)
    bad = list(g)
    bad[6] = dataclasses.replace(bad[6], dil=3, pad=3)  # layer1.1.conv2 with a wrong dilation
    model, _ = OW.fcn_model(tensors50, bad)
    rc, err, _ = convert(lib, model)
    assert rc == _lib.E_MODEL_FORMAT and "layer1.1.conv2" in err


# ---- round 2: slots come from the graph's edges, and real exporter output has more than Conv/Relu ----
def _same_blob(a: bytes, b: bytes):
    (ma, ta), (mb, tb) = W.unpack_blob(a), W.unpack_blob(b)
    assert ma == mb
    for (n0, w0, b0), (n1, w1, b1) in zip(ta, tb):
        assert n0 == n1 and (w0.view(np.uint32) == w1.view(np.uint32)).all() and (b0.view(np.uint32) == b1.view(np.uint32)).all(), n0


@pytest.mark.parametrize("kw", [
    dict(order="ds_first"),                       # downsample serialised before conv1..conv3 (layer1.0: same shape as conv3)
    dict(order="shuffled"),                       # any node order
    dict(dropout=True, identities=True),          # Dropout / Identity nodes between layers
    dict(resize_subgraph=False),                  # constant `scales` instead of the Shape/Gather/Concat size arithmetic
    dict(coord_mode="half_pixel"),
    dict(coord_mode=None),                        # attribute absent: half_pixel is the opset-11 default
    dict(const_nodes=True, inits_as_inputs=True),  # weights in a Constant node; initializers listed as graph inputs
    dict(order="shuffled", unfold_bn=False, dropout=True, identities=True, raw=False, packed_dims=False),
])
def test_slots_follow_the_topology_not_the_node_order(lib, blob50, tensors50, kw):
    """conv3 and downsample.0 of layer1.0 are both [256,64,1,1] s1 p0 -- a file that serialises them in
    the other order must still put each weight set in its own slot."""
    model, _ = OW.fcn_model(tensors50, W.graph(50), rng=np.random.default_rng(7), **kw)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    _same_blob(out, blob50)


def test_unfolded_bn_in_any_order(lib, tensors50):
    specs = W.graph(50)
    m0, _ = OW.fcn_model(tensors50, specs, unfold_bn=True, rng=np.random.default_rng(3))
    m1, _ = OW.fcn_model(tensors50, specs, unfold_bn=True, rng=np.random.default_rng(3), order="ds_first")
    (rc0, e0, b0), (rc1, e1, b1) = convert(lib, m0), convert(lib, m1)
    assert rc0 == 0 and rc1 == 0, (e0, e1)
    _same_blob(b0, b1)


def test_model_without_aux_head_and_resnet101_shape(lib):
    t = W.unpack_blob(W.synth_blob(aux=False))[1]
    model, _ = OW.fcn_model(t, W.graph(50, aux=False), order="shuffled")
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    assert W.unpack_blob(out)[0] == {"depth": 50, "num_classes": 21, "aux": False, "n_convs": 55, "input_u8": False}


def test_graphs_that_are_not_fcn_resnet_are_rejected(lib, tensors50):
    g = W.graph(50)
    cases = [
        (dict(resize_mode="nearest"), "Resize mode must be linear"),
        (dict(coord_mode="align_corners"), "coordinate_transformation_mode 'align_corners'"),
        (dict(coord_mode="asymmetric"), "coordinate_transformation_mode"),
        (dict(opset=10), "opset 10"),
        (dict(opset=9), "Upsample"),
        (dict(swap_outputs=True), "graph output #0"),
        (dict(missing_relu="backbone.layer2.1.conv2"), "backbone.layer2.1"),
        (dict(missing_relu="backbone.layer3.0.out"), "Relu"),
        (dict(missing_relu="stem"), "Relu"),
        (dict(missing_relu="classifier.0"), "classifier"),
        (dict(external="backbone.layer4.0.conv2"), "external_data"),
        (dict(extra_conv=True), "Conv nodes"),
    ]
    for kw, msg in cases:
        model, _ = OW.fcn_model(tensors50, g, **kw)
        rc, err, _ = convert(lib, model)
        assert rc == _lib.E_MODEL_FORMAT and msg in err, (kw, err)


def test_hostile_tensor_dims_do_not_reach_an_allocation(lib):
    """negative / overflowing dims in an initializer are a format error, not a bad_alloc through the C ABI."""
    dims = [(-1, 3, 7, 7), (2 ** 40, 2 ** 40, 7, 7), (0, 3, 7, 7)]
    for d in dims:
        body = OW.f_bytes(1, b"".join(OW._varint(x) for x in d)) + OW.f_varint(2, 1) + OW.f_bytes(9, b"\0" * 16) + OW.f_str(8, "w")
        nodes = OW.f_bytes(1, OW.node("Conv", ["input", "w"], ["y"], [OW.attr_ints("kernel_shape", [7, 7])]))
        g = nodes + OW.f_bytes(5, body) + OW.f_bytes(11, OW.value_info("input", 1, ("N", 3, "H", "W"))) + OW.f_bytes(12, OW.value_info("y", 1, ("N", 64, "H", "W")))
        model = OW.f_varint(1, 6) + OW.f_bytes(7, g) + OW.f_bytes(8, OW.f_str(1, "") + OW.f_varint(2, 12))
        rc, err, _ = convert(lib, model)
        assert rc == _lib.E_MODEL_FORMAT, (d, err)
    # truncated length prefixes / random bytes after a valid header
    rng = np.random.default_rng(5)
    for _ in range(50):
        junk = b"\x08\x06" + bytes(rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8))
        assert convert(lib, junk)[0] == _lib.E_MODEL_FORMAT


def test_identity_cycle_is_a_format_error_not_a_crash(lib):
    """A (non-DAG) file whose Identity nodes feed each other must be rejected without unbounded recursion."""
    n1 = OW.f_bytes(1, OW.node("Identity", ["input"], ["a"]))
    n2 = OW.f_bytes(1, OW.node("Identity", ["a"], ["b"]))
    n3 = OW.f_bytes(1, OW.node("Identity", ["b"], ["a"]))
    g = n1 + n2 + n3 + OW.f_bytes(11, OW.value_info("input", 1, ("N", 3, "H", "W"))) + OW.f_bytes(12, OW.value_info("b", 1, ("N", 21, "H", "W")))
    model = OW.f_varint(1, 6) + OW.f_bytes(7, g) + OW.f_bytes(8, OW.f_str(1, "") + OW.f_varint(2, 12))
    assert convert(lib, model)[0] == _lib.E_MODEL_FORMAT
