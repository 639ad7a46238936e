"""CPU tests: the oracle against the reference's own known answers and the committed goldens."""
import hashlib

import numpy as np
import pytest

from infur_amd import weights as W


# ---- reference KATs (data in tests/golden/reference_kats.json, file:line cited there) ----
def test_color_2(oracle, kats):
    k = kats["color_2"]
    r, g, b, a = k["unmultiplied_rgba"]
    assert (oracle.color_code(k["klass"], k["alpha"]) == oracle.from_rgba_unmultiplied(r, g, b, a)).all()
    # (0.5 * 255) as u8 == 127 and class 2 -> palette[2]
    assert oracle.color_code(2, 0.5)[3] == 127
    assert kats["palette_rgb"][2] == [r, g, b]


def test_decode_0to1(oracle, kats):
    k = kats["decode_0to1"]
    hm = np.linspace(0.0, 1.0, k["linspace"][2], dtype=np.float32).reshape(k["shape"])
    img = oracle.colorcode(hm)
    assert img.shape == (k["height"], k["width"], 4)
    conf = 0
    for p in img.reshape(-1, 4):
        assert (p == oracle.color_code(k["klass"], np.float32(p[3]) / np.float32(255.0))).all()
        assert conf <= p[3], "expected monotonically rising confidence/alpha"
        conf = p[3]
    assert conf == k["last_alpha"]
    kl, _ = oracle.argmax(hm)
    assert (kl == k["klass"]).all() and kats["palette_rgb"][k["klass"] % 20] == k["palette_rgb"]


def test_scale_kats(oracle, kats):
    from oracle.infur_oracle import E_INVALID_SCALE, E_ZERO_SIZE_IN, E_ZERO_SIZE_OUT

    k = kats["scale_from_size0"]
    assert oracle.scale_validate(k["factor"]) == 0
    assert oracle.scale_out_dims(k["w"], k["h"], k["factor"])[0] == E_ZERO_SIZE_IN
    k = kats["scale_to_size0"]
    assert oracle.scale_validate(k["factor"]) == 0
    assert oracle.scale_out_dims(k["w"], k["h"], k["factor"])[0] == E_ZERO_SIZE_OUT
    for f in kats["valid_scale_rejects"]["factors"]:
        assert oracle.scale_validate(f) == E_INVALID_SCALE
    assert oracle.scale_validate(float("nan")) == 0  # `NaN <= 0.0` is false in Rust too
    for d in kats["scale_dims"]:
        assert oracle.scale_out_dims(d["w"], d["h"], d["factor"]) == (0, d["ow"], d["oh"])


def test_infer_seg_model_shape(oracle_model, kats):
    """predict_onnx.rs:371-381: black 320x240 frame -> 2 tensors [21,240,320] (run at 1/4 size here
    for time; the full-size shape is asserted on the GPU path)."""
    k = kats["infer_seg_model"]
    img = np.zeros((k["h"] // 4, k["w"] // 4, 3), np.uint8)
    r = oracle_model.model_forward(oracle_model.pack_normalize(img))
    assert r["out"].shape == (k["shape"][0], k["h"] // 4, k["w"] // 4) == r["aux"].shape


# ---- committed goldens pin the oracle ----
def test_tables_match_golden(oracle, tables):
    assert (oracle.preproc_lut().view(np.uint32) == tables["preproc_lut"].view(np.uint32)).all()
    assert (oracle.color_lut() == tables["color_lut"]).all()


def test_preproc_formula(oracle):
    """predict_onnx.rs:128-136 restated in numpy f32, bit for bit."""
    lut = oracle.preproc_lut()
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std1 = np.float32(1.0) / np.array([0.229, 0.224, 0.225], np.float32)
    v = np.arange(256, dtype=np.float32) * np.float32(1.0) / np.float32(255.0)
    for c in range(3):
        ref = (v - mean[c]) * std1[c]
        assert (ref.view(np.uint32) == lut[c].view(np.uint32)).all()
    img = W.synth_frame(5, 7)
    chw = oracle.pack_normalize(img)
    assert chw[0, 2, 3] == lut[0, img[2, 3, 2]]  # channel 0 = R = byte 2 (BGR -> RGB)
    assert chw[2, 4, 6] == lut[2, img[4, 6, 0]]


def test_blob_is_deterministic(blob50, golden):
    assert hashlib.sha256(blob50).hexdigest().encode() == golden["blob_sha256"].tobytes()


@pytest.mark.parametrize("tag", ["64x48", "97x61"])
def test_oracle_against_golden(oracle_model, golden, tag):
    fr = golden[f"bgr_{tag}"]
    w, h = map(int, tag.split("x"))
    assert (fr == W.synth_frame(h, w)).all()
    chw = oracle_model.pack_normalize(fr)
    assert (chw.view(np.uint32) == golden[f"chw_{tag}"].view(np.uint32)).all()
    r = oracle_model.model_forward(chw)
    for n in ("out_low", "aux_low", "out", "aux"):
        g = golden[f"{n}_{tag}"]
        assert np.abs(r[n] - g).max() <= 1e-5 * np.abs(g).max(), n  # same code, thread count may differ
    rc, rgba = oracle_model.frame_advance(fr)
    assert rc == 0
    assert (rgba != golden[f"rgba_{tag}"]).any(axis=-1).mean() < 1e-3
    for mn, mode in (("nearest", 0), ("bilinear", 1)):
        for fac in (0.5, 0.37, 1.7):
            rc, sc = oracle_model.scale(fr, fac, mode)
            assert rc == 0 and (sc == golden[f"scale_{mn}_{fac}_{tag}"]).all()


def test_colorcode_edge_cases(oracle, golden):
    cc = golden["cc_in"]
    img = oracle.colorcode(cc)
    assert (img == golden["cc_rgba"]).all()
    assert (img[0, 0] == 0).all()                      # all negative -> class 0, alpha 0 -> transparent
    assert (img[0, 1] == oracle.color_code(0, 0.75)).all()  # exact tie -> first class wins
    assert (img[0, 2] == 0).all()                      # NaN never wins
    assert (img[0, 3] == oracle.color_code(3, np.inf)).all() and img[0, 3, 3] == 255
    assert img[0, 5, 3] == 0                           # all zero: strict '>' keeps alpha 0
    assert (img[0, 6] == oracle.color_code(20, 7.5)).all()  # class 20 -> palette[0], alpha saturates


def test_c_oracle_vs_torch_oracle(oracle_model, blob50):
    """Two independent restatements of the network agree (different summation order)."""
    from oracle.infur_oracle import TorchModel

    tm = TorchModel(blob50)
    fr = W.synth_frame(40, 56, index=3)
    chw = oracle_model.pack_normalize(fr)
    r = oracle_model.model_forward(chw)
    tl, ta = tm.forward_lowres(chw)
    for a, b in ((r["out_low"], tl.numpy()), (r["aux_low"], ta.numpy())):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()


@pytest.mark.parametrize("wh", [(1, 1), (3, 3), (2, 5), (9, 4)])
def test_c_oracle_vs_torch_oracle_tiny_frames(oracle_model, blob50, wh):
    """1-pixel feature maps: stride-2 convs whose taps fall entirely outside (floor, not truncation)."""
    from oracle.infur_oracle import TorchModel

    w, h = wh
    fr = W.synth_frame(h, w, index=w + h)
    chw = oracle_model.pack_normalize(fr)
    r = oracle_model.model_forward(chw, full=False)
    tl, ta = TorchModel(blob50).forward_lowres(chw)
    for a, b in ((r["out_low"], tl.numpy()), (r["aux_low"], ta.numpy())):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()


def test_upsample_integer_ratio(oracle):
    """x8 up-sample: interior samples follow the half-pixel rule, borders clamp."""
    x = np.arange(12, dtype=np.float32).reshape(1, 3, 4)
    up = oracle.upsample_bilinear(x, 24, 32)
    assert up.shape == (1, 24, 32)
    assert up[0, 0, 0] == x[0, 0, 0] and up[0, -1, -1] == x[0, -1, -1]
    assert abs(up[0, 0, 4] - (0.0 + 0.0625)) < 1e-6  # src x = 4.5/8 - 0.5 = 0.0625


def test_graph_flops_match_baseline():
    """BASELINE.md section 4."""
    f = W.conv_flops(1080, 1920)
    assert abs(f["total"] / 1e9 - 2342.26) < 0.01 and abs(f["conv3x3"] / 1e9 - 1519.27) < 0.01
    assert abs(W.conv_flops(540, 960)["total"] / 1e9 - 589.77) < 0.01
    assert abs(W.conv_flops(2160, 3840, 101)["total"] / 1e9 - 14278.25) < 0.01
    assert len(W.graph(50)) == 57 and len(W.graph(101)) == 108


def test_blob_roundtrip():
    ts = [("a", np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2), np.array([1, 2], np.float32))]
    blob = W.pack_blob(ts, 50, 21, True)
    meta, out = W.unpack_blob(blob)
    assert meta["depth"] == 50 and out[0][0] == "a" and (out[0][1] == ts[0][1]).all() and (out[0][2] == ts[0][2]).all()


# ---- independent corroboration of two restated third-party rules (round 2) ----
# Neither library below is the reference's; they are independent implementations of the same published conventions
# that happen to be in the image.  They pin the RULE (which source pixel / which interpolation coordinates), not the
# reference's bytes: DESIGN.md section 5 keeps those rows "unpinned".
@pytest.mark.parametrize("wh", [(640, 480), (1280, 720), (97, 61), (33, 17)])
def test_nearest_rule_matches_pillow_outside_exact_ties(oracle, wh):
    """fast_image_resize's Nearest (processing.rs:189) samples the source pixel under the CENTRE of the destination
    pixel: src = floor((dst + 0.5) * src_len / dst_len).  Pillow's NEAREST uses the same convention (in 16.16 fixed
    point, so it may land one pixel lower where (dst + 0.5) * scale is an exact integer): everywhere else the two
    must agree byte for byte."""
    PIL = pytest.importorskip("PIL.Image")
    w, h = wh
    fr = W.synth_frame(h, w, index=5)
    checked = 0
    for f in (0.5, 0.3, 0.77, 1.5, 2.0, 0.123):
        rc, out = oracle.scale(fr, f, 0)
        if rc:
            continue
        oh, ow = out.shape[:2]
        pil = np.asarray(PIL.fromarray(fr).resize((ow, oh), PIL.NEAREST))
        px = (np.arange(ow) + 0.5) * (w / ow)
        py = (np.arange(oh) + 0.5) * (h / oh)
        ok = (np.abs(py - np.round(py)) > 1e-6)[:, None] & (np.abs(px - np.round(px)) > 1e-6)[None, :]
        if ok.mean() < 0.05:  # integer ratios: every centre falls on a pixel boundary, nothing to compare
            continue
        checked += 1
        assert (pil[ok] == out[ok]).all(), (wh, f)
    assert checked >= 3


@pytest.mark.parametrize("dims", [(135, 240, 1080, 1920), (17, 23, 135, 181), (68, 120, 540, 960), (3, 4, 24, 32), (1, 1, 5, 7)])
def test_upsample_rule_matches_torch_interpolate(oracle, dims):
    """The model's final node is what torch.onnx writes for F.interpolate(mode='bilinear', align_corners=False):
    Resize(linear, pytorch_half_pixel).  torch's own CPU kernel is an independent implementation of that coordinate
    rule; the oracle's restatement must agree to rounding (a wrong rule -- align_corners, asymmetric -- is off by
    O(0.1) of the value range)."""
    import torch

    lh, lw, oh, ow = dims
    x = np.random.default_rng(lh * 1000 + lw).standard_normal((21, lh, lw)).astype(np.float32)
    a = oracle.upsample_bilinear(x, oh, ow)
    b = torch.nn.functional.interpolate(torch.from_numpy(x)[None], size=(oh, ow), mode="bilinear", align_corners=False)[0].numpy()
    assert np.abs(a - b).max() < 2e-4
    if lh > 1 and lw > 1:  # and the other conventions are measurably different, so the bound above means something
        c = torch.nn.functional.interpolate(torch.from_numpy(x)[None], size=(oh, ow), mode="bilinear", align_corners=True)[0].numpy()
        assert np.abs(a - c).max() > 1e-2


def test_uint8_input_arm_of_the_preproc(oracle):
    """predict_onnx.rs:114-122 (ColorRange::Uint8) + :296-301 (color_seq stays BGR for non-Float inputs): a model that
    declares a Uint8 image input is fed the frame's bytes -- no channel reversal, no scaling.  The oracle's restatement
    (oracle_pack_u8) against numpy, the whole-path oracle choosing the arm from the blob header, and both CPU oracles
    agreeing on such a model."""
    from oracle.infur_oracle import COracle, TorchModel

    fr = W.synth_frame(40, 56, index=3)
    chw = oracle.pack_u8(fr)
    assert chw.dtype == np.float32 and (chw == fr.transpose(2, 0, 1).astype(np.float32)).all()
    blob8 = W.synth_blob(input_u8=True)
    assert W.unpack_blob(blob8)[0]["input_u8"] is True and W.unpack_blob(W.synth_blob())[0]["input_u8"] is False
    co = COracle()
    assert co.model_load(blob8) == 0
    ref = co.model_forward(chw, full=True, low=False)
    rc, rgba = co.frame_advance(fr, 1.0)
    assert rc == 0 and (rgba == co.colorcode(ref["out"])).all()
    # fed the Float arm instead, the same weights give a different mask: the header is what selects the arm
    wrong = co.model_forward(co.pack_normalize(fr), full=True, low=False)
    assert (co.colorcode(wrong["out"]) != rgba).any()
    t_out, _ = TorchModel(blob8).forward(chw)
    assert np.abs(t_out - ref["out"]).max() / np.abs(ref["out"]).max() < 2e-5
