"""an INDEPENDENT second opinion on the integer oracle (oracle/infur_qoracle.py).  The float forward is
cross-checked three ways (C, torch functional, the exporter's module graph); the quantised forward -- whose bar is
bit-exactness -- was only checked against hand computations by the same author.  PyTorch's quantised CPU kernels (fbgemm,
qnnpack: written by other people, from the same operator definitions) exist in this image, so every operator of the oracle is
run against them on the HOSTILE parameter set of tests/hostile_q.py (zero points anywhere in 0..255, a third of the
multipliers exact powers of two = thousands of exact .5 ties, s8 weights down to -128, both clamps reached).  Where a torch
kernel differs from the ONNX operator text the difference is counted, explained and asserted -- not hidden:

  QuantizeLinear   oracle == torch.quantize_per_tensor on all four engines bit for bit (incl. exact ties) for power-of-two scales;
                   otherwise torch multiplies by the f32 reciprocal where ONNX divides: one count on < 1e-4 of the elements.
  QLinearConv      oracle == qnnpack's quantised conv2d bit for bit (int32 bias added in the integer domain, one f32
                   multiply, round half to even -- the ONNX text).  fbgemm keeps the bias in FLOAT and adds it after the
                   multiply: it differs from the ONNX text (and from qnnpack) by one count on a handful of outputs, every one
                   of them on (or within f32 rounding of) a .5 tie of the integer path, which the float detour moves to the
                   other side; with zero biases fbgemm is bit-identical too.
  QLinearAdd       torch.ops.quantized.add is a DIFFERENT operator text: dequantise, add, quantise --
                   round(((a - a_zp) a_s + (b - b_zp) b_s) / c_s) + c_zp, the formula onnxruntime's ContribOperators.md prints --
                   while onnxruntime's MLAS kernel (what the oracle restates) pre-divides: round((a - a_zp)(a_s / c_s) +
                   (b - b_zp)(b_s / c_s)) + c_zp.  fbgemm's kernel equals a numpy restatement of the former up to its fused dequantise (< 2e-3
                   of the elements, one count); the two
                   texts agree whenever the scale ratios are dyadic and differ by ONE count on <= 0.2 % of the elements
                   otherwise.  Which of the two onnxruntime's kernel computes stays "parity unpinned" (tests/test_gpu_ort.py is
                   the hook); the oracle's own expression is additionally checked against exact rational arithmetic here.
No GPU.  Reference call site: the model `infur/src/predict_onnx.rs:357-381` loads, executed at `:138`."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostile_q as HQ  # noqa: E402
from oracle import infur_qoracle as Q  # noqa: E402

from infur_amd import weights as W

f32 = np.float32
ENGINES = [e for e in ("fbgemm", "qnnpack", "onednn", "x86") if e in torch.backends.quantized.supported_engines]


@pytest.fixture(autouse=True)
def _restore_engine():
    eng = torch.backends.quantized.engine
    yield
    torch.backends.quantized.engine = eng


@pytest.fixture(scope="module")
def hostile():
    """the hostile model and the u8 input of each of its first ten convolutions on a calibration frame (oracle-evaluated)"""
    specs, convs, adds = HQ.hostile_qmodel(50, seed=3)
    rng = np.random.default_rng(5)
    chw = (rng.standard_normal((3, 48, 64)) * 1.1).astype(f32)
    taps = {}
    Q.qforward(W.pack_qblob(convs, adds, 50, 21, True), chw, taps)
    return specs, convs, adds, chw, taps


def torch_qconv(x_u8, c, spec, engine, zero_bias=False):
    import torch.ao.nn.quantized.functional as qF

    torch.backends.quantized.engine = engine
    qx = torch._make_per_tensor_quantized_tensor(torch.from_numpy(x_u8[None].copy()), float(f32(c.x_scale)), int(c.x_zp))
    qw = torch._make_per_channel_quantized_tensor(torch.from_numpy(c.w.astype(np.int8)), torch.from_numpy(c.w_scale.astype(np.float64)),
                                                  torch.zeros(spec.cout, dtype=torch.int64), 0)
    # torch takes the bias as float: b_i32 * (x_s * w_s[o]); the kernels turn it back into the operator's int32 (qnnpack) or keep it float (fbgemm)
    bias_i = np.zeros_like(c.bias) if zero_bias else c.bias
    bias = torch.from_numpy((bias_i.astype(np.float64) * (np.float64(f32(c.x_scale)) * c.w_scale.astype(np.float64))).astype(np.float32))
    y = qF.conv2d(qx, qw, bias, stride=spec.stride, padding=spec.pad, dilation=spec.dil, scale=float(f32(c.y_scale)), zero_point=int(c.y_zp),
                  dtype=torch.quint8)
    return y.int_repr().numpy()[0]


@pytest.mark.parametrize("engine", ENGINES)
def test_quantize_linear_equals_torch(engine):
    """ONNX QuantizeLinear divides (x / scale, what the oracle and onnxruntime do); torch multiplies by the f32 reciprocal.  The two
    are the same function for power-of-two scales (bit for bit here, ties included) and differ by one count on a few elements per
    100,000 otherwise -- exactly the elements where x * (1 / scale) and x / scale round to different sides of a .5: a numpy
    restatement of torch's expression reproduces torch on every element, so the difference is the operator text, not the oracle."""
    torch.backends.quantized.engine = engine
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(300_000) * 3).astype(f32)
    x[:2000] = (np.arange(2000) - 1000).astype(f32) * f32(0.0625)  # exact .5 ties at scale 0.125, and both clamps
    for scale, zp in ((0.125, 128), (1.0 / 64.0, 117), (2.0, 255), (0.0371, 3), (0.3, 0), (0.0173, 128)):
        want = Q.quantize_linear(x, scale, zp)
        got = torch.quantize_per_tensor(torch.from_numpy(x), float(f32(scale)), zp, torch.quint8).int_repr().numpy()
        assert want.min() == 0 or want.max() == 255  # (a clamp is reached)
        recip = np.clip(np.rint(x * (f32(1.0) / f32(scale))) + f32(zp), 0, 255).astype(np.uint8)
        assert (got == recip).all(), (engine, scale, zp)
        d = got.astype(np.int32) - want.astype(np.int32)
        if float(np.log2(scale)).is_integer():
            assert (d == 0).all(), (engine, scale, zp)
        else:
            assert np.abs(d).max() <= 1 and (d != 0).mean() < 1e-4, (engine, scale, zp, int((d != 0).sum()))


def _layer_inputs(hostile):
    specs, convs, adds, chw, taps = hostile
    x = {"backbone.conv1": taps["input"], "backbone.layer1.0.conv1": Q.maxpool_u8(taps["backbone.conv1"]),
         "backbone.layer1.0.conv2": taps["backbone.layer1.0.conv1"], "backbone.layer1.0.conv3": taps["backbone.layer1.0.conv2"],
         "backbone.layer1.0.downsample.0": Q.maxpool_u8(taps["backbone.conv1"]), "backbone.layer1.1.conv1": taps["backbone.layer1.0.conv3"],
         "backbone.layer1.1.conv2": taps["backbone.layer1.1.conv1"], "backbone.layer2.0.conv1": taps["backbone.layer1.2.conv3"],
         "backbone.layer2.0.conv2": taps["backbone.layer2.0.conv1"],  # stride 2
         "backbone.layer3.1.conv2": taps["backbone.layer3.1.conv1"],  # dilation 2
         "backbone.layer4.1.conv2": taps["backbone.layer4.1.conv1"]}  # dilation 4
    by = {s.name: (s, c) for s, c in zip(specs, convs)}
    return [(by[n][0], by[n][1], v) for n, v in x.items()]


def test_qlinearconv_equals_qnnpack_bit_for_bit(hostile):
    if "qnnpack" not in ENGINES:
        pytest.skip("no qnnpack engine in this torch build")
    n_out = n_ties = 0
    for spec, c, x in _layer_inputs(hostile):
        acc = Q.qconv(x, c, spec)
        want = Q.requantize(acc, Q.conv_mult(c), c.y_zp)
        got = torch_qconv(x, c, spec, "qnnpack")
        assert (want == got).all(), (spec.name, int((want != got).sum()))
        t = acc.astype(np.float64) * Q.conv_mult(c).astype(np.float64)[:, None, None]
        n_ties += int((np.abs(t - np.floor(t)) == 0.5).sum())
        n_out += want.size
        assert (want == 0).any() and (want == 255).any(), spec.name  # both clamps are exercised
    print(f"QLinearConv: {n_out} outputs of 11 hostile layers (strides 1/2, dilations 1/2/4, zero points 0..255) == qnnpack; {n_ties} exact .5 ties among them")
    assert n_ties > 300


def test_qlinearconv_vs_fbgemm_differs_only_where_its_float_bias_moves_a_tie(hostile):
    if "fbgemm" not in ENGINES:
        pytest.skip("no fbgemm engine in this torch build")
    n_out = n_diff = 0
    for spec, c, x in _layer_inputs(hostile):
        acc = Q.qconv(x, c, spec)
        want = Q.requantize(acc, Q.conv_mult(c), c.y_zp)
        got = torch_qconv(x, c, spec, "fbgemm")
        d = got.astype(np.int32) - want.astype(np.int32)
        bad = d != 0
        n_out += d.size
        n_diff += int(bad.sum())
        if bad.any():
            assert np.abs(d).max() == 1
            # every differing output sits ON or within f32 rounding of a .5 tie of the integer path (acc * mult evaluated in f64)
            t = acc.astype(np.float64) * Q.conv_mult(c).astype(np.float64)[:, None, None]
            assert (np.abs(np.abs(t - np.floor(t))[bad] - 0.5) < 2e-3).all(), spec.name
        # with zero biases there is no float detour: bit-identical, ties included
        c0 = W.QConv(c.name, c.w, c.w_scale, np.zeros_like(c.bias), c.x_scale, c.x_zp, c.y_scale, c.y_zp)
        assert (Q.requantize(Q.qconv(x, c0, spec), Q.conv_mult(c0), c0.y_zp) == torch_qconv(x, c, spec, "fbgemm", zero_bias=True)).all(), spec.name
    print(f"QLinearConv vs fbgemm: {n_diff} of {n_out} outputs differ (by one count, all on exact ties: fbgemm adds the bias in float)")
    assert n_diff < 1e-4 * n_out


def _doc_text_add(a, b, p):
    """dequantise, add, quantise in f32 -- ContribOperators.md's formula and what torch.ops.quantized.add computes"""
    fa = (a.astype(np.int32) - int(p.a_zp)).astype(f32) * f32(p.a_scale)
    fb = (b.astype(np.int32) - int(p.b_zp)).astype(f32) * f32(p.b_scale)
    return Q.quantize_linear(fa + fb, p.c_scale, p.c_zp)


def test_qlinearadd_two_operator_texts(hostile):
    specs, convs, adds, chw, taps = hostile
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (64, 40, 40)).astype(np.uint8)
    b = rng.integers(0, 256, (64, 40, 40)).astype(np.uint8)
    cases = list(adds) + [W.QAdd(0.5, 3, 0.25, 250, 1.0, 17), W.QAdd(2.0, 0, 0.5, 255, 4.0, 128), W.QAdd(0.37, 3, 1.21, 250, 0.77, 128)]
    worst = 0.0
    for p in cases:
        mlas = Q.qlinear_add(a, b, p)
        doc = _doc_text_add(a, b, p)
        if "fbgemm" in ENGINES:  # torch's kernel IS the documentation's expression
            torch.backends.quantized.engine = "fbgemm"
            qa = torch._make_per_tensor_quantized_tensor(torch.from_numpy(a[None].copy()), float(f32(p.a_scale)), int(p.a_zp))
            qb = torch._make_per_tensor_quantized_tensor(torch.from_numpy(b[None].copy()), float(f32(p.b_scale)), int(p.b_zp))
            got = torch.ops.quantized.add(qa, qb, float(f32(p.c_scale)), int(p.c_zp)).int_repr().numpy()[0]
            # (its vectorised dequantise is fmadd(x, scale, -zp * scale), one rounding fewer than (x - zp) * scale: up to ~1 element per
            #  1,000 lands on the other side of a tie -- still one count)
            dd = got.astype(np.int32) - doc.astype(np.int32)
            assert np.abs(dd).max() <= 1 and (dd != 0).mean() < 2e-3, (p, int((dd != 0).sum()))
            assert np.abs(got.astype(np.int32) - mlas.astype(np.int32)).max() <= 1
        d = mlas.astype(np.int32) - doc.astype(np.int32)
        assert np.abs(d).max() <= 1
        frac = float((d != 0).mean())
        worst = max(worst, frac)
        ra, rb = Fraction(float(f32(p.a_scale))) / Fraction(float(f32(p.c_scale))), Fraction(float(f32(p.b_scale))) / Fraction(float(f32(p.c_scale)))
        dyadic = all(r.denominator & (r.denominator - 1) == 0 and r.denominator <= 256 and r.numerator <= 256 for r in (ra, rb))
        if dyadic:  # both expressions are then exact in f32: the same integer-plus-half values, the same ties
            assert frac == 0.0, p
    print(f"QLinearAdd: MLAS order vs documentation / torch order differ on at most {worst:.3%} of the elements (one count) over {len(cases)} parameter sets")
    assert worst < 2e-3


def test_qlinearadd_of_the_oracle_against_exact_rationals():
    """the oracle's expression evaluated with exact rational arithmetic on the f32 values it uses: every f32 step must be the correctly
    rounded one (numpy's f32 ops are IEEE), so round-half-even of the f32 sum is reproduced exactly"""
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, 4000).astype(np.uint8)
    b = rng.integers(0, 256, 4000).astype(np.uint8)
    for p in (W.QAdd(0.37, 3, 1.21, 250, 0.77, 128), W.QAdd(0.5, 255, 0.25, 0, 1.0, 17), W.QAdd(1.5867987, 9, 1.1093067, 200, 1.5155053, 77)):
        got = Q.qlinear_add(a, b, p)
        ra, rb = f32(p.a_scale) / f32(p.c_scale), f32(p.b_scale) / f32(p.c_scale)
        for k in range(a.size):
            ta = f32(float(Fraction(int(a[k]) - p.a_zp) * Fraction(float(ra))))  # one rounding: the product
            tb = f32(float(Fraction(int(b[k]) - p.b_zp) * Fraction(float(rb))))
            s = f32(float(Fraction(float(ta)) + Fraction(float(tb))))              # one rounding: the sum
            fl = np.floor(np.float64(s))
            r = fl + (1.0 if (np.float64(s) - fl > 0.5 or (np.float64(s) - fl == 0.5 and int(fl) % 2 == 1)) else 0.0)
            assert got[k] == min(255, max(0, int(r) + p.c_zp)), (p, k)


def test_one_whole_bottleneck_through_torch_kernels(hostile):
    """layer1.1 (identity residual) chained through qnnpack's conv kernels and the oracle's QLinearAdd: the block's output equals the
    oracle's own chain bit for bit -- and through fbgemm's add (the other operator text) it differs by at most one count, rarely"""
    if "qnnpack" not in ENGINES:
        pytest.skip("no qnnpack engine in this torch build")
    specs, convs, adds, chw, taps = hostile
    by = {s.name: (s, c) for s, c in zip(specs, convs)}
    x = taps["backbone.layer1.0.conv3"]
    t = x
    for n in ("backbone.layer1.1.conv1", "backbone.layer1.1.conv2", "backbone.layer1.1.conv3"):
        t = torch_qconv(t, by[n][1], by[n][0], "qnnpack")
        if n.endswith(("conv1", "conv2")):
            assert (t == taps[n]).all(), n
    add = adds[1]
    assert (Q.qlinear_add(t, x, add) == taps["backbone.layer1.1.conv3"]).all()
    doc = _doc_text_add(t, x, add)
    d = doc.astype(np.int32) - taps["backbone.layer1.1.conv3"].astype(np.int32)
    print(f"bottleneck layer1.1 with the documentation-order add: {int((d != 0).sum())} of {d.size} bytes differ by one count")
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 2e-3
