"""The f32-split mode (INFUR_DTYPE_F32_SPLIT): f32 tensors, conv GEMMs on the f16 matrix cores with every
operand split into an f16 hi + lo pair (three MFMAs per product, f32 accumulation).  It has to stay an f32
path in every observable way: logits against the f32 oracle well inside north_star's 1e-3 bar
(SPLIT_TOL below is the measured level with a margin), the class map stable outside a 1e-4 band, the
post stage bit-exact given the logits, and all tile configurations bit-identical."""
import ctypes as C

import numpy as np
import pytest

from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

pytestmark = pytest.mark.gpu

SPLIT_TOL = 3e-5  # relative to the largest logit; the native f32 MFMA mode measures ~4e-6, f16 ~2e-3
# INFUR_DTYPE_F32_SPLIT_FP8 ("f32x"): hi*hi on the f16 MFMA, the cross terms hi*lo on the bf8 (e5m2) MX MFMA -- products exact
# to ~2^-13 WHATEVER the tensors' dynamic range (round 3 used e4m3 under static scales: 1.2-1.5e-4 on these friendly weights but
# 7.1e-4 / 5.1e-2 per element on the hostile set; e5m2: 2-3e-4 here, 1.1e-4 / 7e-3 there -- tests/test_gpu_hostile.py).  The
# F(6x6) is the default tile, F(4x4) the one with more room; north_star's bar is 1e-3.
FP8X_TOL = 5e-4


def rel_err(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("dtype,SPLIT_TOL", [("f32s", SPLIT_TOL), ("f32x", FP8X_TOL)])
@pytest.mark.parametrize("shape", [(48, 64), (270, 480), (540, 960)])
def test_split_logits_and_mask(oracle, blob50, shape, dtype, SPLIT_TOL):
    from oracle.infur_oracle import TorchModel

    tm = TorchModel(blob50)
    h, w = shape
    fr = W.synth_frame(h, w, index=3)
    c = Context(device=0, dtype=dtype)
    m = Model(c).control(ModelCmd.LoadBlob(blob50))
    rgba, _ = FramePath(c).advance(fr, 1.0)
    lo, la = m.lowres()
    tl, ta = tm.forward_lowres(oracle.pack_normalize(fr))
    e_out, e_aux = rel_err(lo, tl.numpy()), rel_err(la, ta.numpy())
    print(f"{dtype} R50 {w}x{h}: logits rel err out={e_out:.2e} aux={e_aux:.2e}")
    assert e_out < SPLIT_TOL and e_aux < SPLIT_TOL
    # post stage bit-exact given the logits this mode produced
    assert (rgba == oracle.colorcode(oracle.upsample_bilinear(lo, h, w))).all()
    # class map against the oracle's own logits: differences only where the top-2 gap is inside the band
    ref = oracle.upsample_bilinear(tl.numpy(), h, w)
    kr, _ = oracle.argmax(ref)
    kg, _ = oracle.argmax(oracle.upsample_bilinear(lo, h, w))
    srt = np.sort(np.maximum(ref, 0.0), axis=0)
    gap = srt[-1] - srt[-2]
    bad = kr != kg
    print(f"   class map differs on {bad.mean():.5%} of pixels")
    assert not (bad & (gap >= SPLIT_TOL * np.abs(ref).max())).any()
    assert bad.mean() < (1e-3 if dtype == "f32s" else 1e-2)
    c.close()


def test_split_per_layer(oracle, blob50):
    """every conv of FCN-ResNet50 against the torch-CPU restatement, layer by layer"""
    from oracle.infur_oracle import TorchModel

    tm = TorchModel(blob50)
    c = Context(device=0, keep_activations=True, dtype="f32s")
    m = Model(c).control(ModelCmd.LoadBlob(blob50))
    fr = W.synth_frame(72, 104, index=5)
    out = []
    m.advance(fr, out)
    taps = {}
    tm.forward_lowres(oracle.pack_normalize(fr), taps=taps)
    worst = 0.0
    for i, spec in enumerate(W.graph(50)):
        ref = taps[spec.name].numpy()
        got = np.empty(ref.shape, np.float32)
        cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        c.check(c.L.infur_debug_read_activation(c.h, i, got.ctypes.data, got.size, C.byref(cc), C.byref(hh), C.byref(ww)))
        e = rel_err(got, ref)
        worst = max(worst, e)
        assert e < SPLIT_TOL, (spec.name, e)
    print(f"f32s per-layer worst rel err {worst:.2e}")
    c.close()


def test_split_matches_native_f32_mode(blob50):
    """same frame through the native f32 MFMA mode and the split mode"""
    fr = W.synth_frame(360, 640, index=9)
    lows = {}
    for dt in ("f32", "f32s"):
        c = Context(device=0, dtype=dt)
        m = Model(c).control(ModelCmd.LoadBlob(blob50))
        FramePath(c).advance(fr, 1.0)
        lows[dt] = m.lowres()[0].copy()
        c.close()
    e = rel_err(lows["f32s"], lows["f32"].astype(np.float64))
    print(f"f32s vs f32 MFMA: {e:.2e}")
    assert e < SPLIT_TOL


def test_split_small_and_large_values(oracle):
    """values far from 1.0: operand magnitudes 1e-6 .. 1e3 (f16 hi/lo under- and overflow edges)
    through one 1x1 conv layer via the raw GEMM is not reachable from the ABI, so scale the weights of a
    whole model instead: logits scale linearly in the classifier weights."""
    for scale in (2.0 ** -12, 2.0 ** 10):
        tensors = [(spec, w * np.float32(scale), b * np.float32(scale)) if spec.name == "classifier.4" else (spec, w, b)
                   for spec, w, b in W.synth_tensors(depth=50)]
        assert any(spec.name == "classifier.4" for spec, _, _ in tensors)
        blob = W.pack_blob(tensors, 50, W.NUM_CLASSES, True)
        from oracle.infur_oracle import TorchModel

        tm = TorchModel(blob)
        fr = W.synth_frame(64, 96, index=2)
        c = Context(device=0, dtype="f32s")
        m = Model(c).control(ModelCmd.LoadBlob(blob))
        FramePath(c).advance(fr, 1.0)
        lo, _ = m.lowres()
        tl, _ = tm.forward_lowres(oracle.pack_normalize(fr))
        e = rel_err(lo, tl.numpy())
        print(f"classifier weights x{scale:g}: rel err {e:.2e}")
        assert e < SPLIT_TOL
        c.close()


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f32s", SPLIT_TOL), ("f16", 5e-3)])
def test_fused_downsample_matches_unfused(blob50, dtype, tol):
    """conv3 + downsample as one two-source GEMM (default) against the two-launch form with a residual tensor"""
    fr = W.synth_frame(200, 328, index=4)
    lows = {}
    for fuse in (True, False):
        c = Context(device=0, dtype=dtype, fuse_downsample=fuse)
        m = Model(c).control(ModelCmd.LoadBlob(blob50))
        FramePath(c).advance(fr, 1.0)
        lows[fuse] = [x.copy() for x in m.lowres()]
        c.close()
    for a, b in zip(lows[True], lows[False]):
        e = rel_err(a, b.astype(np.float64))
        print(f"{dtype}: fused vs unfused {e:.2e}")
        assert e < tol


def test_split_activation_range(oracle):
    """The static activation scale (2^2; 2^-3 on Winograd-domain inputs) keeps full precision while the scaled value stays inside the f16
    pair's range: 100x larger activations than the synthetic model's are still f32-grade; 10^4 x larger ones
    saturate (MODE.FP16_OVFL clamps the conversions) -- finite logits, no inf/NaN poisoning."""
    from oracle.infur_oracle import TorchModel

    fr = W.synth_frame(64, 96, index=2)
    for gain, accurate in ((100.0, True), (1.0e4, False)):
        # scaling the stem's weights and bias scales every activation up to the first BN-free add by `gain`
        # (ReLU and max-pool are positively homogeneous; biases downstream are not, which is fine)
        tensors = [(s, w * np.float32(gain), b * np.float32(gain)) if s.name == "backbone.conv1" else (s, w, b)
                   for s, w, b in W.synth_tensors(depth=50)]
        blob = W.pack_blob(tensors, 50, W.NUM_CLASSES, True)
        c = Context(device=0, dtype="f32s")
        m = Model(c).control(ModelCmd.LoadBlob(blob))
        FramePath(c).advance(fr, 1.0)
        lo, _ = m.lowres()
        assert np.isfinite(lo).all()
        act, wino, saturated = c.split_range()  # the range monitor sees it
        print(f"activations x{gain:g}: max |activation| {act:.3g}, max |Winograd input| {wino:.3g}, saturated {saturated}")
        assert saturated == (not accurate) and act > 0 and wino > 0
        if accurate:
            tl, _ = TorchModel(blob).forward_lowres(oracle.pack_normalize(fr))
            e = rel_err(lo, tl.numpy())
            print(f"activations x{gain:g}: rel err {e:.2e}")
            assert e < SPLIT_TOL
        c.close()


@pytest.mark.parametrize("wh", [(1, 1), (7, 5), (33, 17), (130, 66), (257, 129)])
def test_split_tiny_and_ragged_frames(oracle, blob50, wh):
    """Degenerate sizes in the split mode: 1-pixel feature maps, ragged GEMM tiles in the two-source and the
    residual-prefetch forms, partial Winograd tiles."""
    from oracle.infur_oracle import TorchModel

    w, h = wh
    fr = W.synth_frame(h, w, index=w + h)
    c = Context(device=0, dtype="f32s")
    m = Model(c).control(ModelCmd.LoadBlob(blob50))
    rgba, _ = FramePath(c).advance(fr, 1.0)
    lo, la = m.lowres()
    tl, ta = TorchModel(blob50).forward_lowres(oracle.pack_normalize(fr))
    assert lo.shape == tuple(tl.shape)
    assert rel_err(lo, tl.numpy()) < SPLIT_TOL and rel_err(la, ta.numpy()) < SPLIT_TOL
    assert (rgba == oracle.colorcode(oracle.upsample_bilinear(lo, h, w))).all()
    c.close()


def test_split_resnet101_and_stream(oracle):
    """FCN-ResNet101 in the split mode, through the streaming entry points (depth-2 ring) and the batch call"""
    from oracle.infur_oracle import TorchModel

    from infur_amd.app import StreamPath

    blob = W.synth_blob(depth=101)
    tm = TorchModel(blob)
    c = Context(device=0, dtype="f32s")
    m = Model(c).control(ModelCmd.LoadBlob(blob))
    frames = [W.synth_frame(96, 160, index=i) for i in range(3)]
    ref = [oracle.colorcode(oracle.upsample_bilinear(tm.forward_lowres(oracle.pack_normalize(f))[0].numpy(), 96, 160)) for f in frames]
    masks = FramePath(c).advance_batch(frames, 1.0)
    for rgba, want in zip(masks, ref):
        assert rgba.shape == want.shape
        assert (rgba != want).any(axis=-1).mean() < 2e-3  # near-tie pixels only
    lo, _ = m.lowres()
    assert rel_err(lo, tm.forward_lowres(oracle.pack_normalize(frames[-1]))[0].numpy()) < SPLIT_TOL
    sp = StreamPath(c, depth=2)
    outs = list(sp.run(enumerate(frames), 1.0))
    assert [fid for fid, _ in outs] == [0, 1, 2]
    for (_, rgba), got in zip(outs, masks):
        assert (rgba == got).all()  # the ring and the batch call run the same path
    sp.close()
    c.close()


def test_split_range_only_in_split_mode(blob50):
    from infur_amd.processors import InfurError

    c = Context(device=0, dtype="f32")
    Model(c).control(ModelCmd.LoadBlob(blob50))
    FramePath(c).advance(W.synth_frame(48, 64), 1.0)
    with pytest.raises(InfurError):
        c.split_range()
    c.close()
    c = Context(device=0, dtype="f32s")
    Model(c).control(ModelCmd.LoadBlob(blob50))
    FramePath(c).advance(W.synth_frame(48, 64), 1.0)
    act, wino, saturated = c.split_range()
    assert 0 < act < 1e3 and 0 < wino < 1e5 and not saturated
    c.close()


def test_fp8_cross_terms_without_winograd(oracle, blob50):
    """f32x with direct 3x3 convs: the product error alone (no Winograd output transform amplifying it)"""
    from oracle.infur_oracle import TorchModel

    tm = TorchModel(blob50)
    fr = W.synth_frame(270, 480, index=3)
    tl, ta = tm.forward_lowres(oracle.pack_normalize(fr))
    errs = {}
    for name, kw in (("F(6x6)", {}), ("F(4x4)", {"winograd_tile": 4}), ("direct", {"winograd_min_cin": 0xFFFFFFFF})):
        c = Context(device=0, dtype="f32x", **kw)
        m = Model(c).control(ModelCmd.LoadBlob(blob50))
        FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        errs[name] = max(rel_err(lo, tl.numpy()), rel_err(la, ta.numpy()))
        c.close()
    print("f32x logits rel err:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["direct"] < 3e-4 and errs["F(4x4)"] < FP8X_TOL and errs["F(6x6)"] < FP8X_TOL
