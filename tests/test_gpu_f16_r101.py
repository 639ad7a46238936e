"""BASELINE configs[4] (4K frame, FCN-ResNet101, f16 operands on the CDNA4 f16 MFMA) and the f16
mode in general.  f16 is a reduced-precision mode: the oracle stays f32 and the tolerance is
F16_TOL below (measured ~1.5e-3 .. 2e-3 relative on the logits; the f32 mode's bar is 1e-3)."""
import ctypes as C

import numpy as np
import pytest

from infur_amd import weights as W
from infur_amd.processors import ColorCode, Context, FramePath, Model, ModelCmd, Slot

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3
F16_TOL = 5e-3


def rel_err(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def blob101():
    return W.synth_blob(depth=101)


def near_tie_fraction(oracle, got_low, ref_low, h, w, tol):
    """Fraction of pixels whose class differs although the oracle's top-2 gap exceeds tol."""
    ref = oracle.upsample_bilinear(ref_low, h, w)
    kr, _ = oracle.argmax(ref)
    kg, _ = oracle.argmax(oracle.upsample_bilinear(got_low, h, w))
    srt = np.sort(np.maximum(ref, 0.0), axis=0)
    gap = srt[-1] - srt[-2]
    bad = kr != kg
    return float(bad.mean()), float((bad & (gap >= tol)).mean())


def test_f16_mode_r50(oracle, blob50, golden):
    from oracle.infur_oracle import TorchModel

    tm = TorchModel(blob50)
    c = Context(device=0, dtype="f16")
    m = Model(c).control(ModelCmd.LoadBlob(blob50))
    for fr in (golden["bgr_64x48"], W.synth_frame(540, 960, index=1)):
        h, w = fr.shape[:2]
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        tl, ta = tm.forward_lowres(oracle.pack_normalize(fr))
        e_out, e_aux = rel_err(lo, tl.numpy()), rel_err(la, ta.numpy())
        print(f"f16 R50 {w}x{h}: logits rel err out={e_out:.2e} aux={e_aux:.2e}")
        assert e_out < F16_TOL and e_aux < F16_TOL
        # the post stage is still bit-exact given the (f32) low-res logits
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(lo, h, w))).all()
        diff, unexplained = near_tie_fraction(oracle, lo, tl.numpy(), h, w, F16_TOL * np.abs(tl.numpy()).max())
        print(f"   class map differs on {diff:.4%} of pixels, {unexplained:.4%} outside the tolerance band")
        assert unexplained == 0.0 and diff < 0.005
    # Model::advance full-resolution outputs are f32 in both modes
    out = []
    m.advance(golden["bgr_64x48"], out)
    assert out[0].dtype == np.float32 and rel_err(out[0], golden["out_64x48"]) < F16_TOL
    c.close()


@pytest.mark.parametrize("dtype,tol", [("f32", F32_TOL), ("f16", F16_TOL)])
def test_resnet101_per_layer(oracle, blob101, dtype, tol):
    """108 convs of FCN-ResNet101 against the torch-CPU restatement, layer by layer."""
    from oracle.infur_oracle import TorchModel

    tm = TorchModel(blob101)
    c = Context(device=0, keep_activations=True, dtype=dtype)
    m = Model(c).control(ModelCmd.LoadBlob(blob101))
    info = m.get_info()
    assert info.depth == 101 and info.num_classes == 21
    fr = W.synth_frame(72, 104, index=5)
    out = []
    m.advance(fr, out)
    taps = {}
    tl, ta = tm.forward_lowres(oracle.pack_normalize(fr), taps=taps)
    worst = 0.0
    specs = W.graph(101)
    assert len(specs) == 108
    for i, spec in enumerate(specs):
        ref = taps[spec.name].numpy()
        buf = np.empty(ref.shape, np.float32)
        cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
        e = rel_err(buf, ref)
        worst = max(worst, e)
        assert e < tol, (spec.name, e)
    print(f"R101 {dtype}: worst per-layer rel err {worst:.2e}")
    lo, la = m.lowres()
    assert rel_err(lo, tl.numpy()) < tol and rel_err(la, ta.numpy()) < tol
    c.close()


def test_config5_4k_resnet101_f16(oracle, blob101):
    """BASELINE configs[4]: one 3840x2160 frame, FCN-ResNet101, f16 MFMA, scale 1.0."""
    import time

    from oracle.infur_oracle import TorchModel

    w, h = 3840, 2160
    fr = W.synth_frame(h, w, index=3)
    c = Context(device=0, dtype="f16")
    m = Model(c).control(ModelCmd.LoadBlob(blob101))
    fp = FramePath(c)
    rgba, _ = fp.advance(fr, 1.0)  # warm-up (allocations)
    t0 = time.perf_counter()
    rgba2, _ = fp.advance(fr, 1.0)
    dt = time.perf_counter() - t0
    assert rgba.shape == (h, w, 4) and (rgba == rgba2).all()  # deterministic
    lo, la = m.lowres()
    assert lo.shape == (21, 270, 480)
    assert (rgba == oracle.colorcode(oracle.upsample_bilinear(lo, h, w))).all()  # fused post bit-exact
    print(f"4K R101 f16: {dt * 1e3:.1f} ms per frame from host buffers ({W.conv_flops(h, w, 101)['total'] / 1e12:.2f} TFLOP)")
    import torch

    torch.set_num_threads(32)
    tl, ta = TorchModel(blob101).forward_lowres(oracle.pack_normalize(fr))
    e_out, e_aux = rel_err(lo, tl.numpy()), rel_err(la, ta.numpy())
    print(f"4K R101 f16: logits rel err vs f32 CPU oracle out={e_out:.2e} aux={e_aux:.2e}")
    assert e_out < F16_TOL and e_aux < F16_TOL
    diff, unexplained = near_tie_fraction(oracle, lo, tl.numpy(), h, w, F16_TOL * np.abs(tl.numpy()).max())
    print(f"   class map differs on {diff:.4%} of pixels, {unexplained:.4%} outside the tolerance band")
    assert unexplained == 0.0 and diff < 0.005
    c.close()
