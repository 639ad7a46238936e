"""conv1x1_b2b.hip: a bottleneck's conv3 + residual + ReLU and the NEXT bottleneck's conv1 + ReLU as one launch (f16 mode).
The fused launch must give the SAME BITS as the two launches it replaces -- every conv output of the network
(keep_activations), logits and mask -- at sizes whose pixel count is ragged against the 256-pixel workgroup tile, for
FCN-ResNet50 and -101, and the per-layer outputs must stay inside the f16 mode's tolerance of the f32 torch-CPU oracle.
INFUR_B2B=1 / 0 forces / forbids the fused form (the default measures both per shape and keeps the faster)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, sys.argv[1])
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd
depth = int(sys.argv[3])
blob = W.synth_blob(depth=depth)
out = {}
sizes = [(135, 241), (72, 104), (8, 8), (270, 480)] if depth == 50 else [(97, 161), (200, 264)]
for keep in (True, False):
    c = Context(device=0, dtype="f16", keep_activations=keep, profile=True)
    m = Model(c).control(ModelCmd.LoadBlob(blob))
    for (h, w) in sizes:
        fr = W.synth_frame(h, w, index=h)
        rgba, _ = FramePath(c).advance(fr, 1.0)
        lo, la = m.lowres()
        tag = f"{int(keep)}_{w}x{h}"
        out["lo_" + tag] = lo; out["la_" + tag] = la; out["rgba_" + tag] = rgba
        out["nb2b_" + tag] = np.array([sum(1 for r in c.profile() if r["kernel"] == "conv1x1_b2b_f16")])
        if keep:
            for i, spec in enumerate(W.graph(depth)):
                buf = np.empty(64 << 20, np.float32) if i == 0 else buf
                cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
                c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
                out[f"act{i}_" + tag] = buf[: cc.value * hh.value * ww.value].copy()
    c.close()
np.savez(sys.argv[2], **out)
"""


def run(b2b, depth, path, form=None):
    env = dict(os.environ)
    env["INFUR_B2B"] = str(b2b)
    env.pop("INFUR_B2B_FORM", None)
    if form is not None:
        env["INFUR_B2B_FORM"] = str(form)  # 1: 8 waves x 32 pixels, 2: 4 waves x 64 pixels, 3: the 128-pixel workgroup (4 waves x 32)
    subprocess.run([sys.executable, "-c", SCRIPT, ROOT, path, str(depth)], check=True, env=env, timeout=900)
    return np.load(path)


@pytest.mark.parametrize("depth,pairs,form", [(50, 6, 1), (50, 6, 3), (50, 6, 2), (101, 23, 1), (101, 23, 3), (50, 6, None)])
def test_fused_pair_is_bit_identical_to_two_launches(tmp_path, depth, pairs, form):
    """layer2: blocks 1-2 of 4, layer3: blocks 1..n-2 (block 0's conv3 is the two-source GEMM with the downsample branch,
    the last block's successor belongs to the next stage): 2 + 4 pairs in a ResNet-50, 2 + 21 in a ResNet-101.  Every
    workgroup form (the 256-pixel forms and round 4's 128-pixel one; None = the launcher's own choice by M)"""
    ref = run(0, depth, str(tmp_path / "two.npz"))
    got = run(1, depth, str(tmp_path / "fused.npz"), form)
    assert set(ref.files) == set(got.files)
    n_act = sum(1 for k in ref.files if k.startswith("act"))
    assert n_act > 100
    for k in ref.files:
        if k.startswith("nb2b_"):
            assert int(ref[k][0]) == 0 and int(got[k][0]) == pairs, (k, ref[k], got[k])
            continue
        assert ref[k].shape == got[k].shape, k
        assert (ref[k].view(np.uint8) == got[k].view(np.uint8)).all(), k


ORACLE_SCRIPT = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, sys.argv[1])
from infur_amd import weights as W
from infur_amd.processors import Context, Model, ModelCmd
from oracle.infur_oracle import COracle, TorchModel
F16_TOL = 5e-3  # the f16 mode's stated tolerance against the f32 oracle (tests/test_gpu_f16_r101.py)
blob = W.synth_blob()
co, tm = COracle(), TorchModel(blob)
c = Context(device=0, dtype="f16", keep_activations=True, profile=True)
m = Model(c).control(ModelCmd.LoadBlob(blob))
fr = W.synth_frame(135, 241, index=3)
out = []
m.advance(fr, out)
assert sum(1 for r in c.profile() if r["kernel"] == "conv1x1_b2b_f16") == 6
taps = {}
tm.forward_lowres(co.pack_normalize(fr), taps=taps)
worst = 0.0
for i, spec in enumerate(W.graph(50)):
    ref = taps[spec.name].numpy()
    buf = np.empty(ref.shape, np.float32)
    cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
    assert (cc.value, hh.value, ww.value) == ref.shape, spec.name
    e = float(np.abs(buf.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))
    worst = max(worst, e)
    assert e < F16_TOL, (spec.name, e)
print("f16 + fused pairs: per-layer worst rel err", worst)
"""


def test_fused_pair_per_layer_against_torch_oracle():
    """every conv output of the network with the fused launches FORCED (conv3 of block b and conv1 of block b + 1 come out of
    one kernel) against the f32 torch-CPU restatement, at the f16 mode's stated tolerance"""
    env = dict(os.environ)
    env["INFUR_B2B"] = "1"
    r = subprocess.run([sys.executable, "-c", ORACLE_SCRIPT, ROOT], check=True, env=env, timeout=900, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1])
