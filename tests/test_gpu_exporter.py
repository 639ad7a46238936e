"""HIP path loaded from REAL exporter output vs torch's own module graph.

The file the reference loads is torchvision's fcn_resnet50 through torch.onnx.export (opset 12).  tests/tv_fcn.py
produces exactly that kind of file with PyTorch's exporter (synthetic parameters, BatchNorm unfolded in the module and
folded by the exporter) and keeps the module: `ModelCmd::Load(path)` -> hand-written ONNX reader -> HIP forward is compared
with nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d / F.interpolate evaluating the same module on the CPU.  Tolerance:
north_star's 1e-3 relative on the logits, class map identical outside near-ties (tests/test_gpu_ort.py is the same test
against ONNX Runtime, for boxes that have it).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

pytestmark = pytest.mark.gpu
REL_TOL = {"f32": 1e-3, "f32s": 1e-3, "f32x": 1e-3, "f16hl": 1e-3, "f16": 5e-3}


@pytest.fixture(scope="module")
def onnx_path(exported50, tmp_path_factory):
    p = tmp_path_factory.mktemp("exported") / "fcn-resnet50-12.onnx"
    p.write_bytes(exported50[2])
    return str(p)


def module_logits(m, oracle, frame):
    import torch

    with torch.no_grad():
        r = m(torch.from_numpy(oracle.pack_normalize(frame))[None])
    return r["out"][0].numpy(), r["aux"][0].numpy()


@pytest.mark.parametrize("dtype,wh", [("f32", (320, 240)), ("f32", (161, 97)), ("f32", (640, 480)), ("f32s", (320, 240)), ("f32x", (320, 240)), ("f16hl", (320, 240)), ("f16", (320, 240)),
                                      ("f32", (1920, 1080)), ("f32s", (1920, 1080)), ("f32x", (1920, 1080)), ("f16hl", (1920, 1080))])  # the last three: BASELINE configs[1] at full size
def test_exported_file_through_hip_matches_torch_modules(exported50, onnx_path, oracle, dtype, wh):
    m = exported50[0]
    w, h = wh
    frame = W.synth_frame(h, w, index=9)
    want_out, want_aux = module_logits(m, oracle, frame)
    with Context(device=0, dtype=dtype) as c:
        model = Model(c).control(ModelCmd.Load(onnx_path))
        info = model.get_info()
        assert info.input_names == ["input"] and info.output_names == ["out", "aux"] and info.depth == 50
        got = []
        model.advance(frame, got)
        rgba, _ = FramePath(c).advance(frame, 1.0)
    assert got[0].shape == want_out.shape == (21, h, w)
    for g, r, name in ((got[0], want_out, "out"), (got[1], want_aux, "aux")):
        err = np.abs(g - r).max() / np.abs(r).max()
        print(f"{dtype} {w}x{h} {name}: rel err vs torch module graph {err:.2e}")
        assert err < REL_TOL[dtype], (name, err)
    top2 = np.sort(want_out, axis=0)[-2:]
    decided = (top2[1] - top2[0]) > REL_TOL[dtype] * np.abs(want_out).max()
    assert decided.mean() > 0.9
    assert (got[0].argmax(0)[decided] == want_out.argmax(0)[decided]).all()
    # the mask: ColorCode (decode_predict.rs:53-79) of the module's logits, wherever the class is decided and the winning
    # logit cannot cross an alpha-byte boundary within twice the error actually measured
    ref_rgba = oracle.colorcode(want_out)
    e = 2.0 * float(np.abs(got[0] - want_out).max())
    cmax = want_out.max(0).astype(np.float64)
    alpha = lambda v: np.clip(np.floor(v * 255.0), 0, 255)  # `(c_max * 255.0) as u8`: truncation, saturating
    stable = decided & (alpha(cmax - e) == alpha(cmax + e))
    assert stable.mean() > (0.5 if dtype in ("f16", "f32x", "f16hl") else 0.8)  # (f16: 1e-3 of error is half an alpha step)
    assert (rgba[stable] == ref_rgba[stable]).all()


@pytest.fixture(scope="module")
def onnx_path_u8(exported50_u8, tmp_path_factory):
    p = tmp_path_factory.mktemp("exported_u8") / "fcn-resnet50-u8-nhwc.onnx"
    p.write_bytes(exported50_u8[2])
    return str(p)


@pytest.mark.parametrize("dtype,wh", [("f32", (320, 240)), ("f32", (161, 97)), ("f32s", (320, 240)), ("f16", (320, 240)), ("f32", (1280, 720))])
def test_uint8_nhwc_model_through_hip_matches_torch_modules(exported50_u8, onnx_path_u8, dtype, wh):
    """A model that declares a Uint8 NHWC image input (predict_onnx.rs:255,296-301): the reference hands its session the
    frame's bytes, BGR kept, no normalisation (:114-122).  Exported by PyTorch's exporter, loaded through ModelCmd::Load,
    run by the HIP path (identity table in the stem, stem weights stored with the channel axis reversed) and compared with
    the module evaluated on the same u8 frame on the CPU."""
    import torch

    m = exported50_u8[0]
    w, h = wh
    frame = W.synth_frame(h, w, index=13)
    with torch.no_grad():
        r = m(torch.from_numpy(frame)[None])
    want_out, want_aux = r["out"][0].numpy(), r["aux"][0].numpy()
    with Context(device=0, dtype=dtype) as c:
        model = Model(c).control(ModelCmd.Load(onnx_path_u8))
        info = model.get_info()
        assert info.input_names == ["input"] and info.input0_dtype == "Uint8" and info.output_names == ["out", "aux"]
        got = []
        model.advance(frame, got)
    for g, rr, name in ((got[0], want_out, "out"), (got[1], want_aux, "aux")):
        err = np.abs(g - rr).max() / np.abs(rr).max()
        print(f"u8 model {dtype} {w}x{h} {name}: rel err vs torch module graph {err:.2e}")
        assert err < REL_TOL[dtype], (name, err)
    top2 = np.sort(want_out, axis=0)[-2:]
    decided = (top2[1] - top2[0]) > REL_TOL[dtype] * np.abs(want_out).max()
    assert decided.mean() > 0.9
    assert (got[0].argmax(0)[decided] == want_out.argmax(0)[decided]).all()


def test_uint8_blob_whole_path_and_group_replication(oracle):
    """INFURW01 blob with input kind 1 (Uint8): the fused frame path against the whole-path C oracle (which feeds the model
    oracle_pack_u8 for such a blob), a Float model loaded afterwards on the same context switches the stem table back,
    and infur_group_weights_broadcast carries the input kind to the other contexts."""
    from infur_amd.processors import Group
    from oracle.infur_oracle import COracle

    blob8 = W.synth_blob(input_u8=True)
    frame = W.synth_frame(96, 128, index=21)
    co = COracle()
    assert co.model_load(blob8) == 0
    ref = co.model_forward(co.pack_u8(frame), full=False)
    a, b = Context(device=0), Context(device=0)
    ma = Model(a).control(ModelCmd.LoadBlob(blob8))
    assert ma.get_info().input0_dtype == "Uint8"
    rgba, _ = FramePath(a).advance(frame, 1.0)
    lo, _ = ma.lowres()
    assert np.abs(lo - ref["out_low"]).max() / np.abs(ref["out_low"]).max() < 1e-3
    assert (rgba == oracle.colorcode(oracle.upsample_bilinear(lo, 96, 128))).all()
    with Group([a, b]) as g:
        g.weights_broadcast(0)
    mb = Model(b)
    assert mb.get_info().input0_dtype == "Uint8"
    rgba_b, _ = FramePath(b).advance(frame, 1.0)
    assert (rgba_b == rgba).all()
    # back to a Float model on the same context: normalised RGB planes again
    blob = W.synth_blob()
    ma.control(ModelCmd.LoadBlob(blob))
    assert ma.get_info().input0_dtype == "Float"
    assert co.model_load(blob) == 0
    ref = co.model_forward(co.pack_normalize(frame), full=False)
    FramePath(a).advance(frame, 1.0)
    lo, _ = ma.lowres()
    assert np.abs(lo - ref["out_low"]).max() / np.abs(ref["out_low"]).max() < 1e-3
    a.close()
    b.close()
