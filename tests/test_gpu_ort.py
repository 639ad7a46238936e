"""Parity against the reference's OWN runtime, where it exists.

The reference executes the network with ONNX Runtime on a model file it downloads at build time
(infur/src/predict_onnx.rs:288-293, infur-test-gen/build.rs:88-93); neither is in the build image, which is why
DESIGN.md lists the conv stack as "parity unpinned".  This test turns that pin green on any box that has both:

    INFUR_ONNX_MODEL=/path/to/fcn-resnet50-12.onnx INFUR_ONNX_MODEL_INT8=/path/to/fcn-resnet50-12-int8.onnx \
        python -m pytest tests/test_gpu_ort.py -m gpu

It loads the file through ``infur_model_load`` (the hand-written ONNX reader), runs ONNX Runtime on the CPU exactly
as the reference configures it (3 intra-op threads, ORT_ENABLE_EXTENDED), and requires the logits within 1e-3
relative (north_star's tolerance) and the argmax class map bit-exact outside near-ties.  It skips -- it must never
silently pass -- when the package or the file is missing.
"""
import os

import numpy as np
import pytest

from infur_amd import weights as W
from infur_amd.processors import Context, Model, ModelCmd

pytestmark = pytest.mark.gpu
REL_TOL = 1e-3


@pytest.fixture(scope="module")
def ort_session():
    ort = pytest.importorskip("onnxruntime", reason="onnxruntime is not installed in this image")
    path = os.environ.get("INFUR_ONNX_MODEL")
    if not path or not os.path.exists(path):
        pytest.skip("set INFUR_ONNX_MODEL to the zoo's fcn-resnet50-12.onnx to run the ONNX Runtime parity leg")
    so = ort.SessionOptions()
    so.intra_op_num_threads = 3  # predict_onnx.rs:292
    so.graph_optimization_level = ort.GraphOptimizationLevel.ORT_ENABLE_EXTENDED  # predict_onnx.rs:291
    return path, ort.InferenceSession(path, so, providers=["CPUExecutionProvider"])


def reference_preproc(frame: np.ndarray) -> np.ndarray:
    """predict_onnx.rs:103-137 in numpy f32: BGR->RGB, HWC->CHW, (v*1/255 - mean) * (1/std)."""
    mean = np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]
    std1 = (np.float32(1.0) / np.array([0.229, 0.224, 0.225], np.float32))[:, None, None]
    x = np.ascontiguousarray(frame[..., ::-1].transpose(2, 0, 1)).astype(np.float32) * np.float32(1.0) / np.float32(255.0)
    return ((x - mean) * std1)[None]


@pytest.mark.parametrize("wh", [(320, 240), (640, 480)])
def test_logits_and_class_map_match_onnxruntime(ort_session, wh):
    path, sess = ort_session
    w, h = wh
    frame = W.synth_frame(h, w, index=11)
    ref_out, ref_aux = sess.run(None, {sess.get_inputs()[0].name: reference_preproc(frame)})[:2]
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.Load(path))
        info = m.get_info()
        assert info.input_names == [sess.get_inputs()[0].name]
        assert info.output_names == [o.name for o in sess.get_outputs()][:2]
        got = []
        m.advance(frame, got)
    assert got[0].shape == ref_out[0].shape == (21, h, w)  # predict_onnx.rs:371-381 pins exactly this
    for g, r, name in ((got[0], ref_out[0], "out"), (got[1], ref_aux[0], "aux")):
        err = np.abs(g - r).max() / np.abs(r).max()
        print(f"{w}x{h} {name}: rel err vs onnxruntime {err:.2e}")
        assert err < REL_TOL, (name, err)
    # class map: bit-exact wherever ORT's own top-2 margin exceeds the logit tolerance
    ref = ref_out[0]
    top2 = np.sort(ref, axis=0)[-2:]
    decided = (top2[1] - top2[0]) > REL_TOL * np.abs(ref).max()
    assert decided.mean() > 0.9
    assert (got[0].argmax(0)[decided] == ref.argmax(0)[decided]).all()


@pytest.fixture(scope="module")
def ort_int8_session():
    ort = pytest.importorskip("onnxruntime", reason="onnxruntime is not installed in this image")
    path = os.environ.get("INFUR_ONNX_MODEL_INT8")
    if not path or not os.path.exists(path):
        pytest.skip("set INFUR_ONNX_MODEL_INT8 to the zoo's fcn-resnet50-12-int8.onnx to run the quantised ONNX Runtime parity leg")
    so = ort.SessionOptions()
    so.intra_op_num_threads = 3
    # the QOperator graph as written: no re-fusion that could re-associate a requantisation
    so.graph_optimization_level = ort.GraphOptimizationLevel.ORT_DISABLE_ALL
    return path, ort.InferenceSession(path, so, providers=["CPUExecutionProvider"])


def test_quantised_model_matches_onnxruntime_bit_for_bit(ort_int8_session):
    """The file the reference's OWN tests load (predict_onnx.rs:357-381).  Integer accumulation is exact and every requantisation
    is a fixed sequence of f32 operations, so ONNX Runtime's dequantised logits are reproducible bit for bit; the
    full-resolution outputs go through the float Resize, compared at the float tolerance."""
    path, sess = ort_int8_session
    w, h = 640, 480  # the clip size of infur-test-gen (BASELINE configs[0])
    frame = W.synth_frame(h, w, index=11)
    ref_out, ref_aux = sess.run(None, {sess.get_inputs()[0].name: reference_preproc(frame)})[:2]
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.Load(path))
        got = []
        m.advance(frame, got)
    assert got[0].shape == ref_out[0].shape == (21, h, w)
    for g, r, name in ((got[0], ref_out[0], "out"), (got[1], ref_aux[0], "aux")):
        exact = (g.view(np.uint32) == r.view(np.uint32)).mean()
        err = np.abs(g - r).max() / np.abs(r).max()
        print(f"int8 {w}x{h} {name}: {exact:.6f} of the values bit-identical to onnxruntime, max rel err {err:.2e}")
        assert err < 1e-5, (name, err)  # (the float Resize may differ in its last bit; a requantisation off by one count would be ~1e-2)
    assert (got[0].argmax(0) == ref_out[0].argmax(0)).mean() > 0.9999
