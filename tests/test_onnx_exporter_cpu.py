"""The ONNX reader on REAL exporter output, and the oracle against torch's own module graph (no GPU needed).

`fcn-resnet50-12.onnx` -- the file the reference downloads and loads (infur-test-gen/build.rs:88-93,
predict_onnx.rs:288-309) -- is torchvision's `fcn_resnet50` through `torch.onnx.export(opset_version=12)`.  Neither the
file, torchvision nor the `onnx` package is in the image, but PyTorch's exporter is: tests/tv_fcn.py restates the
torchvision modules with torchvision's names and drives the TorchScript exporter, whose C++ serialiser writes the
ModelProto (producer "pytorch", opset 12, BatchNorm folded by the exporter, the Shape/Gather/Slice/Concat size arithmetic
in front of Resize).  So the reader is tested on bytes it did not write, and the oracle's restatement of the
architecture is checked against nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d / F.interpolate wired the torchvision way.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

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


def assert_folded(got, folded, tol=1e-6):
    """exporter folds BatchNorm in f32; the reference fold is float64 -> f32: a few ulp apart, never more"""
    assert len(got) == len(folded)
    for (name, w1, b1), (spec, w0, b0) in zip(got, folded):
        assert name == spec.name and w1.shape == w0.shape, name
        assert np.abs(w1 - w0).max() <= tol * np.abs(w0).max(), name
        assert np.abs(b1 - b0).max() <= tol * max(np.abs(b0).max(), 1e-3), name


def test_exported_model_is_pytorch_output(exported50):
    _, _, onnx = exported50
    assert b"pytorch" in onnx[:64]  # ModelProto.producer_name, written by the exporter
    assert len(onnx) > 140e6  # 35 M f32 parameters as raw_data initializers


def test_reader_accepts_real_exporter_output(lib, exported50, blob50):
    _, folded, onnx = exported50
    rc, err, out = convert(lib, onnx)
    assert rc == 0, err
    meta, got = W.unpack_blob(out)
    assert meta == {"depth": 50, "num_classes": 21, "aux": True, "n_convs": 57, "input_u8": False}
    assert_folded(got, folded)
    # same PRNG streams as synth_blob: the module's folded parameters are the blob every other test runs on
    assert_folded(got, [(s, w, b) for s, (_, w, b) in zip(W.graph(50), W.unpack_blob(blob50)[1])])


@pytest.mark.parametrize("depth,aux,dynamic", [(50, False, True), (50, True, False), (101, True, True)])
def test_reader_on_exporter_variants(lib, depth, aux, dynamic):
    """no aux head (one graph output), static shapes (Resize sizes constant-folded by the exporter), ResNet-101"""
    import tv_fcn

    m, folded = tv_fcn.synth_fcn(depth, aux=aux)
    rc, err, out = convert(lib, tv_fcn.export_onnx(m, h=48, w=64, dynamic=dynamic))
    assert rc == 0, err
    meta, got = W.unpack_blob(out)
    assert meta == {"depth": depth, "num_classes": 21, "aux": aux, "n_convs": len(folded), "input_u8": False}
    assert_folded(got, folded)


@pytest.mark.parametrize("wh", [(96, 64), (161, 97)])
def test_oracle_matches_torch_module_graph(lib, oracle, exported50, wh):
    """pre-processed frame -> torch's module graph (eval, BatchNorm unfolded)  vs  the C oracle AND the functional torch
    oracle running the blob converted from the exported file: logits at stride 8 and at full size"""
    import torch

    from oracle.infur_oracle import COracle, TorchModel

    m, _, onnx = exported50
    rc, err, blob = convert(lib, onnx)
    assert rc == 0, err
    w, h = wh
    chw = oracle.pack_normalize(W.synth_frame(h, w, index=5))
    with torch.no_grad():
        want = m(torch.from_numpy(chw)[None])
    want = {k: v[0].numpy() for k, v in want.items()}
    co = COracle()
    assert co.model_load(blob) == 0
    ref = co.model_forward(chw, full=True, low=False)
    t_out, t_aux = TorchModel(blob).forward(chw)
    for name, got_c, got_t in (("out", ref["out"], t_out), ("aux", ref["aux"], t_aux)):
        scale = np.abs(want[name]).max()
        assert np.abs(got_c - want[name]).max() / scale < 2e-5, name  # measured 3e-6
        assert np.abs(got_t - want[name]).max() / scale < 2e-5, name
    # class map of the C oracle == class map of the module's logits outside near-ties
    top2 = np.sort(want["out"], axis=0)[-2:]
    decided = (top2[1] - top2[0]) > 1e-4 * np.abs(want["out"]).max()
    assert decided.mean() > 0.95 and (ref["out"].argmax(0)[decided] == want["out"].argmax(0)[decided]).all()


def test_uint8_nhwc_model_from_the_exporter(lib, oracle, exported50_u8):
    """A model with a Uint8 NHWC image input, written by PyTorch's exporter (Transpose + Cast in front of the stem): the
    reader accepts it and marks the blob; the oracle fed as the reference feeds such a model (predict_onnx.rs:114-122: the
    frame's bytes, BGR kept, no normalisation -> oracle_pack_u8) agrees with the module evaluated on the frame itself."""
    import torch

    from oracle.infur_oracle import COracle, TorchModel

    m, folded, onnx = exported50_u8
    rc, err, blob = convert(lib, onnx)
    assert rc == 0, err
    meta, got = W.unpack_blob(blob)
    assert meta == {"depth": 50, "num_classes": 21, "aux": True, "n_convs": 57, "input_u8": True}
    assert_folded(got, folded)
    assert_folded(got, [(s, w, b) for s, (_, w, b) in zip(W.graph(50), W.unpack_blob(W.synth_blob(input_u8=True))[1])])
    frame = W.synth_frame(97, 161, index=6)
    with torch.no_grad():
        want = m(torch.from_numpy(frame)[None])
    want = {k: v[0].numpy() for k, v in want.items()}
    chw = oracle.pack_u8(frame)
    assert (chw[0] == frame[..., 0]).all() and (chw[2] == frame[..., 2]).all()  # plane 0 = B: BGR kept
    co = COracle()
    assert co.model_load(blob) == 0
    ref = co.model_forward(chw, full=True, low=False)
    t_out, t_aux = TorchModel(blob).forward(chw)
    for name, got_c, got_t in (("out", ref["out"], t_out), ("aux", ref["aux"], t_aux)):
        scale = np.abs(want[name]).max()
        assert np.abs(got_c - want[name]).max() / scale < 2e-5, name
        assert np.abs(got_t - want[name]).max() / scale < 2e-5, name
