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
    assert meta == {"depth": 50, "num_classes": 21, "aux": True, "n_convs": 57}
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
        (dict(input_type=2), "NCHW Float segmentation models only"),          # u8 models: valid for the reference, not here
        (dict(input_dims=("N", "H", "W", 3)), "NCHW Float segmentation models only"),
        (dict(conv_op="QLinearConv"), "quantised model"),                        # the int8 zoo file of infur-test-gen
        (dict(drop_last=3), "Conv nodes"),
    ]
    for kw, msg in cases:
        model, _ = OW.fcn_model(tensors50, g, **kw)
        rc, err, _ = convert(lib, model)
        assert rc == _lib.E_MODEL_FORMAT and msg in err, (kw, err)
    rc, err, _ = convert(lib, b"\x08\x06garbage")
    assert rc == _lib.E_MODEL_FORMAT


def test_wrong_conv_attributes_are_rejected(lib, tensors50):
    import dataclasses

    g = W.graph(50)
    bad = list(g)
    bad[6] = dataclasses.replace(bad[6], dil=3, pad=3)  # layer1.1.conv2 with a wrong dilation
    model, _ = OW.fcn_model(tensors50, bad)
    rc, err, _ = convert(lib, model)
    assert rc == _lib.E_MODEL_FORMAT and "layer1.1.conv2" in err
