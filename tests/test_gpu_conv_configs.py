"""Every tile configuration of the conv kernel (the autotuner may pick any of them) must give
the same bits: run the whole network with each configuration forced and compare logits."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd
blob = W.synth_blob()
out = {}
for dt in ("f32", "f16", "f32x"):
    c = Context(device=0, dtype=dt)
    m = Model(c).control(ModelCmd.LoadBlob(blob))
    fr = W.synth_frame(135, 241, index=4)
    rgba, _ = FramePath(c).advance(fr, 1.0)
    lo, la = m.lowres()
    out[dt + "_lo"] = lo; out[dt + "_la"] = la; out[dt + "_rgba"] = rgba
    c.close()
np.savez(sys.argv[2], **out)
"""


def run_with_cfg(cfg, path):
    env = dict(os.environ)
    if cfg is None:
        env.pop("INFUR_CONV_CFG", None)
    else:
        env["INFUR_CONV_CFG"] = str(cfg)
    subprocess.run([sys.executable, "-c", SCRIPT, ROOT, path], check=True, env=env, timeout=300)
    return np.load(path)


def test_all_tile_configurations_are_bit_identical(tmp_path):
    ref = run_with_cfg(0, str(tmp_path / "cfg0.npz"))
    # None = autotuned mix; 13, 14, 16, 17 = LDS-DMA staging, 15 = register-resident 1x1 (f16 only), 19 / 20 = the 3x3 form with the input
    # patch resident in LDS (conv3x3_halo.hip, f16 only)
    for cfg in list(range(1, 18)) + [19, 20, 21, None]:
        got = run_with_cfg(cfg, str(tmp_path / f"cfg{cfg}.npz"))
        for k in ref.files:
            assert (ref[k].view(np.uint8) == got[k].view(np.uint8)).all(), (cfg, k)
