"""Quantised models (the QOperator int8 form the reference's own tests load: fcn-resnet50-12-int8.onnx, predict_onnx.rs:357-381):
u8 activations x s8 weights on v_mfma_i32_32x32x32_i8 with the QLinearConv / QLinearAdd / DequantizeLinear arithmetic in the
epilogues.  Integer accumulation is exact and every requantisation step is one defined f32 operation, so this is the one
forward where parity is BIT-EXACT: every layer's u8 tensor, the dequantised logits and the mask must equal the integer oracle
(oracle/infur_qoracle.py) byte for byte, at sizes that are ragged against every tile."""
import ctypes as C

import numpy as np
import pytest

from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qblob():
    from oracle import infur_qoracle as Q

    return Q.synth_qblob()


@pytest.mark.parametrize("size", [(96, 128), (72, 104), (135, 241), (8, 8)])
def test_every_layer_and_the_logits_are_bit_exact(qblob, oracle, size):
    from oracle import infur_qoracle as Q

    h, w = size
    fr = W.synth_frame(h, w, index=h)
    taps = {}
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr), taps)
    c = Context(device=0, keep_activations=True)
    m = Model(c).control(ModelCmd.LoadBlob(qblob))
    info = m.get_info()
    assert info.output_names == ["out", "aux"] and info.depth == 50
    rgba, _ = FramePath(c).advance(fr, 1.0)
    lo, la = m.lowres()
    for i, spec in enumerate(W.graph(50)):
        ref = taps[spec.name]
        if spec.role in ("cls", "auxcls"):
            continue  # the logit convs leave the stack dequantised: compared below
        buf = np.empty(64 << 18, np.float32) if i == 0 else buf
        cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
        got = buf[: cc.value * hh.value * ww.value].reshape(cc.value, hh.value, ww.value)
        assert (hh.value, ww.value) == ref.shape[1:] and cc.value >= ref.shape[0], spec.name
        assert (got[: ref.shape[0]] == ref.astype(np.float32)).all(), (spec.name, int((got[: ref.shape[0]] != ref).sum()))
        assert (got[ref.shape[0]:] == 0).all(), spec.name  # channel padding of the 64-channel tensors
    assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
    assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, h, w))).all()
    c.close()


def test_model_advance_and_tile_configurations(qblob, oracle):
    """Model::advance (full-resolution f32 outputs) on a quantised model; results do not depend on the tile configuration
    (autotuned here, heuristic there) -- integer accumulation has no summation order"""
    from oracle import infur_qoracle as Q

    fr = W.synth_frame(120, 200, index=9)
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr))
    outs = []
    for autotune in (True, False):
        c = Context(device=0, autotune=autotune)
        m = Model(c).control(ModelCmd.LoadBlob(qblob))
        out = []
        m.advance(fr, out)
        assert len(out) == 2 and out[0].shape == (21, 120, 200)
        assert (out[0].view(np.uint32) == oracle.upsample_bilinear(ref_lo, 120, 200).view(np.uint32)).all()
        assert (out[1].view(np.uint32) == oracle.upsample_bilinear(ref_aux, 120, 200).view(np.uint32)).all()
        outs.append(out[0])
        c.close()
    assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all()


def test_quantised_logits_track_the_float_model(qblob, oracle, blob50):
    """sanity of the quantiser, not of the kernels: the dequantised logits stay near the float model's"""
    from oracle.infur_oracle import TorchModel

    fr = W.synth_frame(96, 128, index=5)
    c = Context(device=0)
    m = Model(c).control(ModelCmd.LoadBlob(qblob))
    FramePath(c).advance(fr, 1.0)
    lo, _ = m.lowres()
    fl, _ = TorchModel(blob50).forward_lowres(oracle.pack_normalize(fr))
    fl = fl.numpy()
    assert np.abs(lo - fl).max() / np.abs(fl).max() < 0.15 and (lo.argmax(0) == fl.argmax(0)).mean() > 0.85
    c.close()


def test_switching_between_float_and_quantised_models(qblob, blob50, oracle):
    """ModelCmd::Load of a quantised file after a float one (and back) on the same context"""
    from oracle import infur_qoracle as Q

    fr = W.synth_frame(64, 96, index=1)
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(blob50))
        assert not m.get_info().quantised
        ref_f, _ = FramePath(c).advance(fr, 1.0)
        m.control(ModelCmd.LoadBlob(qblob))
        assert m.get_info().quantised and not m.get_info().resize_u8_heads  # (ABI 4: the host can tell what it loaded)
        # a host built against the ABI-3 struct (136 bytes) asks for its prefix only: nothing is written behind it
        from infur_amd import _lib

        buf = (C.c_uint8 * C.sizeof(_lib.ModelInfoC))()
        C.memset(buf, 0xEE, len(buf))
        assert c.L.infur_model_info_get_sized(c.h, buf, 136) == 0
        assert bytes(buf[:5]) == b"input" and all(b == 0xEE for b in buf[136:])
        assert c.L.infur_model_info_get_sized(c.h, buf, 0) == _lib.E_INVALID_ARG
        rq, _ = FramePath(c).advance(fr, 1.0)
        lo, _ = m.lowres()
        ref_lo, _ = Q.qforward(qblob, oracle.pack_normalize(fr))
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all()
        m.control(ModelCmd.LoadBlob(blob50))
        again, _ = FramePath(c).advance(fr, 1.0)
        assert (again == ref_f).all() and rq.shape == ref_f.shape


def test_loading_the_qoperator_onnx_file_equals_the_blob(qblob, oracle, tmp_path):
    """ModelCmd::Load("...-int8.onnx") (predict_onnx.rs:288-309 with the file of predict_onnx.rs:357-381): the QOperator graph is
    walked by its edges into the same INFURQ01 bytes, so frames come out identical; the file's own tensor names are reported"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_writer as OW
    from oracle import infur_qoracle as Q

    meta, convs, adds = W.unpack_qblob(qblob)
    p = tmp_path / "fcn-resnet50-int8.onnx"
    p.write_bytes(OW.fcn_qmodel(convs, adds, W.graph(50), order="shuffled", rng=np.random.default_rng(3)))
    fr = W.synth_frame(88, 120, index=2)
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr))
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.Load(str(p)))
        info = m.get_info()
        assert info.input_names == ["input"] and info.output_names == ["out", "aux"] and info.depth == 50 and info.quantised
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, 88, 120))).all()


def test_quantised_model_through_group_stream_and_batch(qblob, oracle):
    """the rest of the API surface with a quantised model: replication to other contexts (infur_group_weights_broadcast copies
    the quantisation tables with the arena), the sharded batch, the streaming ring with a second lane -- masks equal the
    single-context ones, which are the oracle's"""
    from infur_amd.app import StreamPath
    from infur_amd.processors import Group
    from oracle import infur_qoracle as Q

    imgs = [W.synth_frame(64 + 16 * (i % 2), 96, index=i) for i in range(5)]
    ctxs = [Context(device=0) for _ in range(3)]
    try:
        Model(ctxs[0]).control(ModelCmd.LoadBlob(qblob))
        ref = FramePath(ctxs[0]).advance_batch(imgs, 1.0)
        lo, _ = Q.qforward(qblob, oracle.pack_normalize(imgs[0]))
        assert (ref[0] == oracle.colorcode(oracle.upsample_bilinear(lo, *imgs[0].shape[:2]))).all()
        with Group(ctxs) as g:
            g.weights_broadcast(root=0)
            got = g.advance_batch(imgs, 1.0)
            assert len(got) == 5 and all((a == b).all() for a, b in zip(got, ref))
        for c in ctxs[1:]:
            solo, _ = FramePath(c).advance(imgs[1], 1.0)
            assert (solo == ref[1]).all()
        sp = StreamPath(ctxs[0], depth=3)
        sp.add_lane(ctxs[1])
        out = list(sp.run([(i, im) for i, im in enumerate(imgs)], 1.0))
        assert [o[0] for o in out] == list(range(5)) and all((o[1] == r).all() for o, r in zip(out, ref))
        sp.close()
        # a float model on one context and a quantised one on the other are different arithmetic: not a lane
        f = Context(device=0)
        Model(f).control(ModelCmd.LoadBlob(W.synth_blob(depth=50)))
        sp2 = StreamPath(f, depth=2)
        with pytest.raises(Exception):
            sp2.add_lane(ctxs[2])
        sp2.close()
        f.close()
    finally:
        for c in ctxs:
            c.close()


def test_replicas_of_a_quantised_model_do_not_point_into_the_roots_arena(qblob, oracle):
    """a replica re-bases EVERY device pointer of the model into its own arena -- the pixel-pair copies of
    layer1 (d_w2 / d_qbias2 / d_qmult2) used to keep pointing at the root's.  After the broadcast the root's arena is
    freed and its memory overwritten (a float model of another depth is loaded over it, then a large allocation is
    filled with 0xFF): the replicas must still produce the oracle's bytes.  An even pooled width takes the pair path."""
    import torch

    from infur_amd.processors import Group
    from oracle import infur_qoracle as Q

    fr = W.synth_frame(64, 96, index=7)  # pooled width 24: even -> layer1 on pixel pairs
    lo, _ = Q.qforward(qblob, oracle.pack_normalize(fr))
    want = oracle.colorcode(oracle.upsample_bilinear(lo, 64, 96))
    ctxs = [Context(device=0) for _ in range(3)]
    try:
        root = Model(ctxs[0]).control(ModelCmd.LoadBlob(qblob))
        with Group(ctxs) as g:
            g.weights_broadcast(root=0)
        root.control(ModelCmd.Load(""))  # unload: the root's arena goes back to the allocator
        junk = torch.full((256 << 20,), 0xFF, dtype=torch.uint8, device="cuda")  # ... and is overwritten
        torch.cuda.synchronize()
        root.control(ModelCmd.LoadBlob(W.synth_blob(depth=50)))
        for c in ctxs[1:]:
            got, _ = FramePath(c).advance(fr, 1.0)
            assert (got == want).all()
        del junk
    finally:
        for c in ctxs:
            c.close()


def test_resnet101_quantised_is_bit_exact(oracle):
    from oracle import infur_qoracle as Q

    qb = Q.synth_qblob(depth=101)
    fr = W.synth_frame(72, 88, index=4)
    taps = {}
    ref_lo, ref_aux = Q.qforward(qb, oracle.pack_normalize(fr), taps)
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(qb))
        assert m.get_info().depth == 101
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, 72, 88))).all()


@pytest.mark.parametrize("seed,size", [(0, (88, 120)), (1, (135, 241)), (2, (40, 56))])
def test_hostile_quantisation_parameters_are_bit_exact(oracle, seed, size):
    """tests/hostile_q.py: zero points anywhere in 0..255, power-of-two multipliers (exact .5 ties on every other accumulator:
    only round-half-to-EVEN passes), both saturation tails populated, s8 weights down to -128, biases to 2^20, residual
    sums whose inputs differ 4x in scale -- every layer's bytes, the logits and the mask must still equal the integer oracle"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from hostile_q import hostile_qblob
    from oracle import infur_qoracle as Q

    qb = hostile_qblob(seed=seed)
    h, w = size
    fr = W.synth_frame(h, w, index=seed)
    taps = {}
    ref_lo, ref_aux = Q.qforward(qb, oracle.pack_normalize(fr), taps)
    sat_lo = sat_hi = 0
    with Context(device=0, keep_activations=True) as c:
        m = Model(c).control(ModelCmd.LoadBlob(qb))
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        buf = np.empty(64 << 18, np.float32)
        for i, spec in enumerate(W.graph(50)):
            if spec.role in ("cls", "auxcls"):
                continue
            ref = taps[spec.name]
            sat_lo += int((ref == 0).sum())
            sat_hi += int((ref == 255).sum())
            cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
            c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
            got = buf[: cc.value * hh.value * ww.value].reshape(cc.value, hh.value, ww.value)
            assert (got[: ref.shape[0]] == ref.astype(np.float32)).all(), (spec.name, int((got[: ref.shape[0]] != ref).sum()))
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, h, w))).all()
    assert sat_lo > 1000 and sat_hi > 1000  # the set really reaches both clamps
    # the product form (no kept activations): stem + max-pool fused on the f16 MFMA, pooled before requantised
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(qb))
        rgba2, _ = FramePath(c).advance(fr, 1.0)
        lo2, la2 = m.lowres()
        assert (lo2.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la2.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba2 == rgba).all()


def test_configs0_640x480_quantised_frame_is_bit_exact(qblob, oracle):
    """BASELINE configs[0]: the reference's own CPU-runnable case -- a 640x480 frame through the int8 model (infur-test-gen's clip
    size + fcn-resnet50-12-int8.onnx, predict_onnx.rs:357-381) -- at full size: dequantised logits, full-resolution outputs and
    mask equal the integer oracle bit for bit"""
    from oracle import infur_qoracle as Q

    fr = W.synth_frame(480, 640, index=11)
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr))
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(qblob))
        out = []
        m.advance(fr, out)
        assert len(out) == 2 and out[0].shape == (21, 480, 640)
        assert (out[0].view(np.uint32) == oracle.upsample_bilinear(ref_lo, 480, 640).view(np.uint32)).all()
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, 480, 640))).all()


def test_full_1080p_quantised_frame_is_bit_exact(qblob, oracle):
    """the headline frame size through the quantised model: 1920x1080, every dequantised logit and every mask byte equal to the
    integer oracle (a float model can only be compared within a tolerance at this size; this one is defined bit for bit)"""
    from oracle import infur_qoracle as Q

    fr = W.synth_frame(1080, 1920, index=5)
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr))
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(qblob))
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        assert lo.shape == (21, 135, 240)
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, 1080, 1920))).all()


@pytest.mark.parametrize("cfg", ["15", "18"])
def test_conv1x1_q8_forced_on_every_1x1_is_bit_exact(cfg):
    """conv1x1_q8.hip (configuration 15 of the quantised mode; 18 = the same walk with the N tiles of an M tile shared out over
    several workgroups) is picked by measurement per shape; here it is forced wherever it is a candidate (INFUR_CONV_CFG is read
    once per process: a child process) and the layer-by-layer, hostile-parameter and ResNet-101 cases of this file run again"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, INFUR_CONV_CFG=cfg)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                        "every_layer or hostile or resnet101 or group_stream"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("cfg", ["19", "20"])
def test_lds_patch_3x3_forced_on_every_3x3_is_bit_exact(cfg):
    """conv3x3_halo.hip on the i8 MFMA (round 5: configurations 19 / 20 of the quantised mode -- the input patch of a 16 x 16 tile
    resident in LDS for all nine taps, QLinearConv's requantisation in its epilogue) forced onto every stride-1 3x3 conv, dilation 1,
    2 and 4, ragged tiles included: the layer-by-layer, hostile-parameter, ResNet-101 and 640x480 cases of this file run again in a
    child process and must still give the integer oracle's bytes"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, INFUR_CONV_CFG=cfg)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                        "every_layer or hostile or resnet101 or 640x480"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_layer1_padded_form_still_gives_the_oracles_bytes():
    """forward_q runs layer1 of a quantised model on PIXEL PAIRS when the pooled width is even (no channel padding, pair-arranged
    weights); INFUR_Q_NOPAIR=1 keeps the padded form (read once per process: a child process).  The whole-frame cases of this file
    run again with the pair view off (the every-layer cases keep activations, which is the padded form in either process)"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, INFUR_Q_NOPAIR="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "640x480 or 1080p or hostile or group_stream or pixel_pair_sizes"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("size", [(64, 96), (97, 130), (48, 8), (61, 200)])
def test_pixel_pair_sizes(qblob, oracle, size):
    """pooled widths 24 (pairs), 33 (odd: the padded form), 2 (one pair), 50 (pairs, odd height): logits and mask against the integer oracle"""
    from oracle import infur_qoracle as Q

    h, w = size
    fr = W.synth_frame(h, w, index=5)
    c = Context(device=0)
    m = Model(c).control(ModelCmd.LoadBlob(qblob))
    rgba, _ = FramePath(c).advance(fr, 1.0)
    lo, la = m.lowres()
    c.close()
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr))
    assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
    assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, h, w))).all()


@pytest.mark.parametrize("size", [(96, 128), (135, 241)])
def test_models_that_resize_before_they_dequantise(qblob, oracle, size, tmp_path):
    """QLinearConv -> Resize (u8) -> DequantizeLinear (blob flag bit 0; the order onnxruntime's QOperator quantiser writes when Resize
    is on its list): the heads' codes are interpolated in float, truncated to u8 as UpsampleBilinear<uint8_t> does, then dequantised --
    full-resolution outputs, mask and the low-res read-back against the oracle's restatement, through the blob and the ONNX file"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_writer as OW
    from oracle import infur_qoracle as Q

    h, w = size
    meta, convs, adds = W.unpack_qblob(qblob)
    blob_r = W.pack_qblob(convs, adds, 50, 21, True, resize_u8=True)
    p = tmp_path / "int8_resize_u8.onnx"
    p.write_bytes(OW.fcn_qmodel(convs, adds, W.graph(50), resize_u8=True))
    fr = W.synth_frame(h, w, index=3)
    (codes, aux_codes), params = Q.qforward_codes(qblob, oracle.pack_normalize(fr))
    ref = [Q.resize_u8_then_dequantise(cd, zp, sc, h, w, oracle.upsample_bilinear) for cd, (zp, sc) in zip((codes, aux_codes), params)]
    plain = oracle.upsample_bilinear(Q.qforward(qblob, oracle.pack_normalize(fr))[0], h, w)
    assert (ref[0] != plain).mean() > 0.2  # the two orders really are different functions
    for cmd in (ModelCmd.LoadBlob(blob_r), ModelCmd.Load(str(p))):
        with Context(device=0) as c:
            m = Model(c).control(cmd)
            out = []
            m.advance(fr, out)
            assert (out[0].view(np.uint32) == ref[0].view(np.uint32)).all() and (out[1].view(np.uint32) == ref[1].view(np.uint32)).all()
            rgba, _ = FramePath(c).advance(fr, 1.0)
            assert (rgba == oracle.colorcode(ref[0])).all()
            assert m.get_info().quantised and m.get_info().resize_u8_heads
            lo, la = m.lowres()  # read back as logits: DequantizeLinear of the codes
            want = Q.qforward(qblob, oracle.pack_normalize(fr))
            assert (lo.view(np.uint32) == want[0].view(np.uint32)).all() and (la.view(np.uint32) == want[1].view(np.uint32)).all()


def test_qdq_format_file_loads_and_runs_like_the_qoperator_file(qblob, oracle, tmp_path):
    """a QDQ-format file (onnxruntime's default output since 1.11) of the same quantised model: fused by the reader into the same
    blob, so ModelCmd::Load gives the same bits"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_writer as OW
    from oracle import infur_qoracle as Q

    meta, convs, adds = W.unpack_qblob(qblob)
    p = tmp_path / "fcn-resnet50-int8-qdq.onnx"
    p.write_bytes(OW.fcn_qmodel(convs, adds, W.graph(50), qdq=True, order="shuffled", rng=np.random.default_rng(8), resize_subgraph=True))
    fr = W.synth_frame(104, 152, index=6)
    ref_lo, ref_aux = Q.qforward(qblob, oracle.pack_normalize(fr))
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.Load(str(p)))
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        assert (lo.view(np.uint32) == ref_lo.view(np.uint32)).all() and (la.view(np.uint32) == ref_aux.view(np.uint32)).all()
        assert (rgba == oracle.colorcode(oracle.upsample_bilinear(ref_lo, 104, 152))).all()


def test_gpu_logits_hash_to_the_committed_golden_vectors():
    """tests/golden/int8_golden.json (made by tests/golden/make_int8_golden.py from the integer oracle): the HIP path's dequantised
    logits must hash to the committed values -- the fixture comparison that needs no oracle at run time"""
    import hashlib
    import json
    import os
    import re
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from hostile_q import hostile_qblob

    for case in json.load(open(os.path.join(here, "golden", "int8_golden.json")))["cases"]:
        seed = int(re.search(r"seed=(\d+)", case["model"]).group(1))
        h, w, idx = map(int, re.search(r"synth_frame\((\d+), (\d+), index=(\d+)\)", case["frame"]).groups())
        blob = hostile_qblob(seed=seed)
        assert hashlib.sha1(blob).hexdigest() == case["blob_sha1"]
        with Context(device=0) as c:
            m = Model(c).control(ModelCmd.LoadBlob(blob))
            FramePath(c).advance(W.synth_frame(h, w, index=idx), 1.0)
            lo, la = m.lowres()
        assert list(lo.shape) == case["out_low_shape"]
        assert hashlib.sha1(np.ascontiguousarray(lo).tobytes()).hexdigest() == case["out_low_sha1"]
        assert hashlib.sha1(np.ascontiguousarray(la).tobytes()).hexdigest() == case["aux_low_sha1"]
        assert np.bincount(lo.argmax(0).ravel(), minlength=21).tolist() == case["argmax_histogram"]
