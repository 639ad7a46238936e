"""INFUR_ROCTX=1: every stage / layer launch of the frame path is a named roctx range (the counterpart of the reference's `tracing`
spans, infur/src/main.rs:18-24) -- checked end to end with `rocprofv3 --marker-trace`: the marker trace of one small frame must hold
the frame range, the fused stem and pre / post stages and a range per convolution, each naming the kernel that ran."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1])
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd
c = Context(device=0, dtype="f32")
Model(c).control(ModelCmd.LoadBlob(W.synth_blob(depth=50)))
FramePath(c).advance(W.synth_frame(96, 128, index=3), 1.0)
c.close()
"""


def test_marker_trace_names_every_layer(tmp_path):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        pytest.skip("rocprofv3 not installed")
    env = dict(os.environ, INFUR_ROCTX="1", TMPDIR=str(tmp_path))
    out = tmp_path / "trace"
    r = subprocess.run([prof, "--marker-trace", "--output-format", "csv", "-d", str(out), "--", sys.executable, "-c", SCRIPT, ROOT],
                       cwd=str(tmp_path), env=env, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = ""
    for f in glob.glob(str(out / "**" / "*marker*"), recursive=True):
        text += open(f, errors="replace").read()
    assert text, (os.listdir(out) if out.exists() else "no output directory", r.stderr[-1000:])
    for name in ("infur frame", "infur forward", "backbone.conv1+maxpool [stem_pool]", "backbone.layer1.0.conv1 [conv_igemm_f32<",
                 "backbone.layer4.2.conv3 [", "classifier.4 [", "out.resize+colorcode [upsample_argmax_shade]"):
        assert name in text, name
