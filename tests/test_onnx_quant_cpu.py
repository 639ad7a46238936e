"""ModelCmd::Load of the QUANTISED model file (`fcn-resnet50-12-int8.onnx` is what the reference's tests load:
infur-test-gen/build.rs:88-93, predict_onnx.rs:357-381): the host-only QOperator ONNX -> INFURQ01 converter, the INFURQ01
container itself and the integer oracle's operator arithmetic.  No GPU.  Files are fabricated by tests/onnx_writer.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import onnx_writer as OW  # noqa: E402
from test_onnx_cpu import convert  # noqa: E402

from infur_amd import weights as W


def random_qmodel(depth=50, aux=True, ncls=21, seed=0, per_tensor=()):
    """structurally valid quantised parameters with arbitrary values (the converter moves bytes, it does not run them)"""
    rng = np.random.default_rng(seed)
    specs = W.graph(depth, ncls, aux)
    convs, adds = [], []
    for s in specs:
        w = rng.integers(-127, 128, (s.cout, s.cin, s.k, s.k), dtype=np.int8)
        ws = (rng.random(s.cout).astype(np.float32) + np.float32(0.5)) * np.float32(1e-3)
        if s.name in per_tensor:
            ws[:] = ws[0]
        x_zp = int(rng.integers(0, 256)) if (s.pad == 0 or s.role == "stem") else 0
        convs.append(W.QConv(s.name, w, ws, rng.integers(-(1 << 20), 1 << 20, s.cout, dtype=np.int32), float(np.float32(rng.random() + 0.01)), x_zp,
                             float(np.float32(rng.random() + 0.01)), int(rng.integers(0, 256))))
        if s.role == "conv3":
            adds.append(W.QAdd(convs[-1].y_scale, convs[-1].y_zp, float(np.float32(rng.random() + 0.01)), int(rng.integers(0, 256)),
                               float(np.float32(rng.random() + 0.01)), int(rng.integers(0, 256))))
    return specs, convs, adds


@pytest.fixture(scope="module")
def q50():
    return random_qmodel(per_tensor=("backbone.layer2.1.conv2", "classifier.4"))


def test_infurq01_container_round_trips():
    specs, convs, adds = random_qmodel(seed=3)
    blob = W.pack_qblob(convs, adds, 50, 21, True)
    meta, c2, a2 = W.unpack_qblob(blob)
    assert meta == {"depth": 50, "num_classes": 21, "aux": True, "n_convs": 57, "n_adds": 16}
    for a, b in zip(convs, c2):
        assert a.name == b.name and (a.w == b.w).all() and (a.w_scale == b.w_scale).all() and (a.bias == b.bias).all()
        assert (np.float32(a.x_scale), a.x_zp, np.float32(a.y_scale), a.y_zp) == (np.float32(b.x_scale), b.x_zp, np.float32(b.y_scale), b.y_zp)
    assert [tuple(np.float32(v) for v in vars(a).values()) for a in adds] == [tuple(np.float32(v) for v in vars(a).values()) for a in a2]
    assert W.pack_qblob(c2, a2, 50, 21, True) == blob
    with pytest.raises(ValueError):
        W.unpack_qblob(b"INFURW01" + blob[8:])


@pytest.mark.parametrize("order,swap", [("topo", False), ("shuffled", None), ("ds_first", True)])
def test_qoperator_model_converts_to_the_same_blob_bit_for_bit(lib, q50, order, swap):
    specs, convs, adds = q50
    model = OW.fcn_qmodel(convs, adds, specs, order=order, swap_add=swap, rng=np.random.default_rng(7),
                          per_tensor_scale=("backbone.layer2.1.conv2", "classifier.4"), relu_after=())
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 50, 21, True)


@pytest.mark.parametrize("w_dtype,w_zp,vec", [(2, 128, False), (2, 127, True), (3, 1, True)])
def test_weights_around_a_zero_point_are_recentred(lib, w_dtype, w_zp, vec):
    """UINT8 weights around 128 (what older onnxruntime quantisers write), or INT8 weights with a non-zero zero point: the
    reader stores w - w_zp as s8, the blob equals the one packed from the s8 weights"""
    specs, convs, adds = random_qmodel(seed=9)
    if w_zp != 128:  # keep w + w_zp inside the stored type's range
        for c in convs:
            c.w = np.clip(c.w, -126, 126)
    model = OW.fcn_qmodel(convs, adds, specs, w_dtype=w_dtype, w_zp=w_zp, vector_wzp=vec)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 50, 21, True)


def test_variants_no_aux_r101_redundant_relu_missing_bias(lib):
    specs, convs, adds = random_qmodel(depth=101, aux=False, ncls=7, seed=5)
    zero_zp = [c.name for c in convs if c.y_zp == 0]
    for c in convs[:6]:
        c.y_zp = 0
    adds[0].a_zp = convs[3].y_zp  # (layer1.0.conv3 is conv #3)
    convs[2].bias = np.zeros_like(convs[2].bias)
    model = OW.fcn_qmodel(convs, adds, specs, relu_after=tuple(c.name for c in convs[:6]), no_bias=(convs[2].name,), vector_wzp=True)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 101, 7, False), zero_zp
    meta, _, _ = W.unpack_qblob(out)
    assert meta["depth"] == 101 and not meta["aux"] and meta["n_convs"] == 106 and meta["n_adds"] == 33


@pytest.mark.parametrize("kwargs,msg", [
    (dict(w_zp=3, shift_weights=False), "does not fit 8 bits"),   # INT8 weights -127..127 with zero point 3: -130 exists
    (dict(w_dtype=2, w_zp=200, shift_weights=False), "does not fit 8 bits"),  # UINT8 weights 0..127 with zero point 200
    (dict(input_type=2), "only a Float"),
    (dict(coord_mode="align_corners"), "align_corners"),
    (dict(drop_last=1), "QLinearConv nodes"),
    (dict(stem_scale=0.123), "differ from the image's QuantizeLinear"),
    (dict(dq_scale=0.5), "DequantizeLinear must use"),
    (dict(pad_zp_conv=("backbone.layer1.0.conv2", 9)), "pads an input whose zero point"),
    (dict(relu_after=("backbone.layer3.0.conv3",)), "Relu follows"),
    (dict(extra_qconv=True), "QLinearConv nodes"),
])
def test_files_the_reader_must_reject(lib, q50, kwargs, msg):
    specs, convs, adds = q50
    if "relu_after" in kwargs:
        assert next(c for c in convs if c.name == "backbone.layer3.0.conv3").y_zp != 0
    rc, err, out = convert(lib, OW.fcn_qmodel(convs, adds, specs, **kwargs))
    assert rc != 0 and out is None and msg in err, err


def test_wrong_shapes_and_dynamic_quantisation_are_format_errors(lib, q50):
    specs, convs, adds = q50
    bad = list(convs)
    c = bad[6]
    assert c.name == "backbone.layer1.1.conv2"
    bad[6] = W.QConv(c.name, c.w[:, :, :1, :1].copy(), c.w_scale, c.bias, c.x_scale, c.x_zp, c.y_scale, c.y_zp)  # a 3x3 written as 1x1
    rc, err, _ = convert(lib, OW.fcn_qmodel(bad, adds, specs))
    assert rc != 0 and "is [64,64,1,1]" in err and "expected [64,64,3,3]" in err, err
    model = OW.fcn_qmodel(convs, adds, specs)
    rc, err, _ = convert(lib, model.replace(b"QLinearAdd", b"QLinearMul"))
    assert rc != 0 and "QLinearMul" in err, err
    # truncations of a valid file never crash and never succeed
    for cut in (len(model) // 3, len(model) - 100, 40):
        rc, err, out = convert(lib, model[:cut])
        assert rc != 0 and out is None


def test_operator_arithmetic_of_the_integer_oracle():
    """the oracle's requantisation against hand-computed values: round half to even, saturation, the f32 multiplier"""
    from oracle import infur_qoracle as Q

    assert Q.quantize_linear(np.array([0.5, 1.5, 2.5, -0.5, 300.0, -300.0], np.float32), 1.0, 0).tolist() == [0, 2, 2, 0, 255, 0]
    assert Q.quantize_linear(np.array([0.05, -0.05], np.float32), 0.1, 128).tolist() == [128, 128]  # 0.5 -> 0, -0.5 -> -0
    acc = np.array([[[5]], [[-5]], [[1000000]]], np.int64)
    assert Q.requantize(acc, np.array([0.5, 0.5, 0.5], np.float32), 10).ravel().tolist() == [12, 8, 255]  # 2.5 -> 2, -2.5 -> -2
    a = np.array([[[200]]], np.uint8)
    b = np.array([[[100]]], np.uint8)
    p = W.QAdd(0.5, 0, 0.25, 100, 0.125, 3)
    assert Q.qlinear_add(a, b, p).ravel().tolist() == [255]  # 200 * 4 + 0 + 3 saturates
    p = W.QAdd(0.5, 0, 0.25, 100, 1.0, 3)
    assert Q.qlinear_add(a, b, p).ravel().tolist() == [103]
    c = W.QConv("x", np.zeros((2, 1, 1, 1), np.int8), np.array([0.1, 0.3], np.float32), np.zeros(2, np.int32), 0.7, 0, 0.9, 0)
    m = Q.conv_mult(c)
    assert m.dtype == np.float32 and m.tolist() == [float(np.float32(np.float32(0.7) * np.float32(0.1)) / np.float32(0.9)),
                                                     float(np.float32(np.float32(0.7) * np.float32(0.3)) / np.float32(0.9))]


def test_tiny_quantised_forward_matches_a_direct_integer_evaluation():
    """qconv (float64 conv2d on integers) against an explicit integer loop, padding included"""
    from oracle import infur_qoracle as Q

    rng = np.random.default_rng(1)
    spec = next(x for x in W.graph(50) if x.role == "conv2" and x.stride == 2)
    x = rng.integers(0, 256, (spec.cin, 9, 11), dtype=np.uint8)
    c = W.QConv(spec.name, rng.integers(-127, 128, (spec.cout, spec.cin, 3, 3), dtype=np.int8), np.ones(spec.cout, np.float32),
                rng.integers(-1000, 1000, spec.cout, dtype=np.int32), 1.0, 0, 1.0, 0)
    acc = Q.qconv(x, c, spec)
    xp = np.pad(x.astype(np.int64), ((0, 0), (spec.pad, spec.pad), (spec.pad, spec.pad)))
    oh, ow = acc.shape[1:]
    for (o, y, xx) in [(0, 0, 0), (5, oh - 1, ow - 1), (spec.cout - 1, 2, 3)]:
        win = xp[:, y * 2:y * 2 + 3, xx * 2:xx * 2 + 3]
        assert acc[o, y, xx] == int((win * c.w[o].astype(np.int64)).sum()) + int(c.bias[o])


def test_hostile_quantised_model_is_valid_reaches_the_edges_and_survives_the_onnx_round_trip(lib):
    """tests/hostile_q.py (the parameter set of tests/test_gpu_quant.py's hostile case): structurally valid for the reader,
    and -- evaluated by the oracle on small frames -- it does produce exact .5 ties (thousands, in convolutions and residual sums
    alike), both saturation tails, and non-zero zero points on both sides of residual sums, down to the heads"""
    from hostile_q import hostile_qmodel
    from oracle import infur_qoracle as Q

    specs, convs, adds = hostile_qmodel(seed=0)
    blob = W.pack_qblob(convs, adds, 50, 21, True)
    rc, err, out = convert(lib, OW.fcn_qmodel(convs, adds, specs, order="shuffled", rng=np.random.default_rng(1)))
    assert rc == 0, err
    assert out == blob
    assert min(c.w.min() for c in convs) == -128 and sum(a.b_zp != 0 and a.c_zp != 0 for a in adds) >= 4
    assert all(c.x_zp == 0 for s, c in zip(specs, convs) if s.pad and s.role != "stem")
    st = hostile_qmodel.last_stats  # what the calibration frame met on its way through the model
    assert st["add_ties"] > 1000 and st["conv_ties"] > 1000 and st["sat_lo"] > 10000 and st["sat_hi"] > 10000, st
    # half-to-even is what the oracle does on an exact tie
    assert Q.requantize(np.array([[[1]], [[3]], [[-1]], [[5]]], np.int64), np.full(4, 0.5, np.float32), 10).ravel().tolist() == [10, 12, 10, 12]
    a = np.array([[[3, 5, 7]]], np.uint8)
    assert Q.qlinear_add(a, np.zeros_like(a), W.QAdd(1.0, 0, 1.0, 0, 2.0, 0)).ravel().tolist() == [2, 2, 4]  # 1.5, 2.5, 3.5
    rng = np.random.default_rng(0)
    taps = {}
    Q.qforward(blob, (rng.standard_normal((3, 24, 32)) * 1.2).astype(np.float32), taps)
    for name in ("backbone.layer1.0.conv1", "backbone.layer3.2.conv2", "classifier.0"):  # (conv outputs; a conv3 tap is the residual sum)
        x = taps[name]
        assert (x == 0).any() and (x == 255).any() and len(np.unique(x)) > 150, name


def test_quantize_cli_on_an_onnx_file_with_a_raw_clip(lib, tmp_path, blob50):
    """python -m infur_amd.quantize: float ONNX file + raw bgr24 calibration clip -> INFURQ01, identical to the library call on the
    same frames; the result is a structurally valid quantised blob (and loads: tests/test_gpu_quant.py uses the same quantiser)"""
    import subprocess

    from infur_amd import quantize

    _, tensors = W.unpack_blob(blob50)
    model, _ = OW.fcn_model(tensors, W.graph(50))
    (tmp_path / "fcn.onnx").write_bytes(model)
    frames = [W.synth_frame(72, 96, index=i) for i in range(2)]
    (tmp_path / "clip.bgr24").write_bytes(b"".join(f.tobytes() for f in frames) + b"\x00" * 100)  # (a truncated trailing frame)
    r = subprocess.run([sys.executable, "-m", "infur_amd.quantize", str(tmp_path / "fcn.onnx"), str(tmp_path / "q.qblob"), "--frames",
                        str(tmp_path / "clip.bgr24"), "--width", "96", "--height", "72", "--calib", "5"], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "calibrated on 2 frame(s)" in r.stdout, r.stdout + r.stderr
    out = (tmp_path / "q.qblob").read_bytes()
    assert out == quantize.quantise_model(blob50, [quantize.normalise(f) for f in frames])
    meta, convs, adds = W.unpack_qblob(out)
    assert meta["n_convs"] == 57 and meta["n_adds"] == 16 and all(c.w.dtype == np.int8 for c in convs)
    assert all(c.x_zp == 0 for s, c in zip(W.graph(50), convs) if s.pad and s.role != "stem")


def test_resize_before_dequantize_sets_the_blob_flag(lib, q50):
    """QLinearConv -> Resize (u8) -> DequantizeLinear, what onnxruntime's QOperator quantiser writes with Resize on its list: same
    tensors, flag bit 0 of the blob; the two heads must agree on the order"""
    specs, convs, adds = q50
    rc, err, out = convert(lib, OW.fcn_qmodel(convs, adds, specs, resize_u8=True, per_tensor_scale=("backbone.layer2.1.conv2", "classifier.4")))
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 50, 21, True, resize_u8=True)
    assert W.unpack_qblob(out)[0].get("resize_u8") is True
    assert W.unpack_qblob(W.pack_qblob(convs, adds, 50, 21, True))[0].get("resize_u8") is None
    rc, err, _ = convert(lib, OW.fcn_qmodel(convs, adds, specs, resize_u8=(True, False)))
    assert rc != 0 and "order Resize and DequantizeLinear differently" in err, err


def test_u8_resize_arithmetic_of_the_oracle():
    """interpolate the codes in float, truncate (static_cast<uint8_t>), dequantise"""
    from oracle import infur_qoracle as Q

    codes = np.array([[[10.0, 20.0], [30.0, 41.0]]], np.float32)

    def up(x, h, w):  # a stand-in interpolation: the four values averaged
        return np.full((x.shape[0], h, w), x.mean(), np.float32)

    got = Q.resize_u8_then_dequantise(codes, 5, 0.5, 2, 2, up)
    assert got.shape == (1, 2, 2) and (got == np.float32((25 - 5) * 0.5)).all()  # 25.25 -> 25


@pytest.mark.parametrize("resize_u8", [False, True])
def test_dynamic_size_resize_subgraph_is_looked_through(lib, q50, resize_u8):
    """the exporter's Shape -> Gather -> Unsqueeze -> Concat arithmetic around Resize stays in float in a quantised file: the
    Shape readers of the image input and of the logits are not data consumers"""
    specs, convs, adds = q50
    model = OW.fcn_qmodel(convs, adds, specs, resize_subgraph=True, resize_u8=resize_u8, order="shuffled", rng=np.random.default_rng(4),
                          per_tensor_scale=("backbone.layer2.1.conv2", "classifier.4"))
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 50, 21, True, resize_u8=resize_u8)


@pytest.mark.parametrize("kw", [dict(), dict(resize_u8=True, order="shuffled", qdq_share_dq=True), dict(resize_subgraph=True, swap_add=True),
                                dict(resize_u8=True, resize_subgraph=True)])
def test_qdq_format_is_fused_into_the_same_blob(lib, kw):
    """QDQ format (onnxruntime's default since 1.11): DequantizeLinear -> Conv / Add / MaxPool / Resize -> [Relu ->] QuantizeLinear groups,
    which ONNX Runtime fuses into the QLinear operators when the session is created -- the reader does the same fusion and must
    arrive at the bytes the QOperator file gives"""
    from hostile_q import hostile_qmodel

    specs, convs, adds = hostile_qmodel(seed=4)  # (tensor parameters consistent along the edges: a shared DequantizeLinear is legal)
    c = next(c for c in convs if c.name == "backbone.layer3.2.conv1")
    c.w_scale = np.full_like(c.w_scale, c.w_scale[0])
    model = OW.fcn_qmodel(convs, adds, specs, qdq=True, rng=np.random.default_rng(2), per_tensor_scale=("backbone.layer3.2.conv1",), **kw)
    rc, err, out = convert(lib, model)
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 50, 21, True, resize_u8=bool(kw.get("resize_u8")))


def test_qdq_graphs_that_cannot_be_fused_are_rejected(lib, q50):
    specs, convs, adds = q50
    model = OW.fcn_qmodel(convs, adds, specs, qdq=True)
    assert convert(lib, model)[0] == 0
    # a Relu in front of a QuantizeLinear whose zero point is not 0 is not the clamp
    name = next(c.name for s_, c in zip(specs, convs) if c.y_zp != 0 and s_.role == "conv3")
    rc, err, _ = convert(lib, OW.fcn_qmodel(convs, adds, specs, qdq=True, relu_after=(name,)))
    assert rc != 0 and "Relu in front of a QuantizeLinear whose zero point is not 0" in err, err
    # a bias DequantizeLinear that is not in units of x_scale * w_scale, or has a zero point: not the fused operator's function
    rc, err, _ = convert(lib, OW.fcn_qmodel(convs, adds, specs, qdq=True, qdq_bias_scale_factor=("backbone.layer2.1.conv1", 1.5)))
    assert rc != 0 and "scale is not x_scale * w_scale" in err, err
    rc, err, _ = convert(lib, OW.fcn_qmodel(convs, adds, specs, qdq=True, qdq_bias_zp=("backbone.layer2.1.conv1", 3)))
    assert rc != 0 and "zero point is not an all-zero INT32" in err, err
    assert convert(lib, OW.fcn_qmodel(convs, adds, specs, qdq=True, qdq_bias_zp=("backbone.layer2.1.conv1", 0)))[0] == 0
    # an operator group that is not closed by a QuantizeLinear / a float operator the fusion does not know
    rc, err, _ = convert(lib, model.replace(b"MaxPool", b"MaxPooX"))
    assert rc != 0 and err
    rc, err, _ = convert(lib, model.replace(b"\x03Add", b"\x03Sub"))
    assert rc != 0 and err
    for cut in (len(model) // 2, len(model) - 64):
        rc, err, out = convert(lib, model[:cut])
        assert rc != 0 and out is None


def test_qoperator_file_with_float_residual_sums(lib):
    """QLinearConv everywhere but the residual sums left as DequantizeLinear -> Add -> [Relu ->] QuantizeLinear (quantisers older than
    com.microsoft QLinearAdd): fused like ONNX Runtime fuses them, same blob"""
    from hostile_q import hostile_qmodel

    specs, convs, adds = hostile_qmodel(seed=6)
    rc, err, out = convert(lib, OW.fcn_qmodel(convs, adds, specs, float_add=True, order="ds_first"))
    assert rc == 0, err
    assert out == W.pack_qblob(convs, adds, 50, 21, True)


def test_integer_oracle_against_its_committed_golden_vectors():
    """tests/golden/int8_golden.json: hashes of the oracle's logits on two hostile models (pure numpy / exact integer arithmetic:
    portable).  A change of the oracle's arithmetic, of the blob format or of the generator shows up here before any GPU run."""
    import importlib.util
    import json

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_int8_golden", os.path.join(here, "golden", "make_int8_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "int8_golden.json")))
    assert mod.build() == want
