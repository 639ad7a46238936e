#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run from the repo root in the build container:  python tests/golden/make_golden.py

Two kinds of data end up here (data only -- no reference source text):

* ``reference_kats.json`` -- the known-answer material of the reference's own tests for
  the hot path, transcribed as inputs/expected outputs:
    infur/src/decode_predict.rs:94-97   color_2
    infur/src/decode_predict.rs:100-116 decode_0to1
    infur/src/processing.rs:289-303     scale_from_size0 / scale_to_size0
    infur/src/app.rs:187,199,216        scaled frame dimensions
    infur/src/predict_onnx.rs:371-381   infer_seg_model output count/shape
* ``oracle_small.npz`` / ``oracle_tables.npz`` -- outputs of this repo's CPU oracle
  (oracle/infur_oracle.c, cross-checked against the torch-CPU restatement) on seeded
  synthetic frames and weights.  The reference itself (Rust + ONNX Runtime + a downloaded
  model file) cannot run in this environment, so these pin the ORACLE, not the reference:
  the network numerics remain "parity unpinned" (DESIGN.md).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from infur_amd import weights as W  # noqa: E402
from oracle.infur_oracle import COracle, TorchModel  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    co = COracle()
    kats = {
        "color_2": {"klass": 2, "alpha": 0.5, "unmultiplied_rgba": [25, 225, 255, 127],
                    "src": "infur/src/decode_predict.rs:94-97"},
        "decode_0to1": {"linspace": [0.0, 1.0, 22 * 24 * 32], "shape": [22, 24, 32], "width": 32, "height": 24,
                        "klass": 21, "palette_index": 1, "palette_rgb": [75, 25, 230], "last_alpha": 255,
                        "alpha_monotone": True, "src": "infur/src/decode_predict.rs:100-116"},
        "scale_from_size0": {"w": 0, "h": 10, "factor": 0.99, "error": "ZeroSizeIn",
                             "src": "infur/src/processing.rs:289-295"},
        "scale_to_size0": {"w": 10, "h": 10, "factor": 0.00000001, "error": "ZeroSizeOut",
                           "src": "infur/src/processing.rs:297-303"},
        "valid_scale_rejects": {"factors": [0.0, -1.0, -0.5], "error": "ValidScaleError",
                                "src": "infur/src/processing.rs:161-163"},
        "scale_dims": [
            {"w": 1280, "h": 720, "factor": 0.5, "ow": 640, "oh": 360, "src": "infur/src/app.rs:187"},
            {"w": 640, "h": 480, "factor": 0.5, "ow": 320, "oh": 240, "src": "infur/src/app.rs:199"},
            {"w": 1280, "h": 720, "factor": 2.0, "ow": 2560, "oh": 1440, "src": "infur/src/app.rs:216"},
        ],
        "infer_seg_model": {"w": 320, "h": 240, "n_outputs": 2, "shape": [21, 240, 320],
                            "src": "infur/src/predict_onnx.rs:371-381"},
        "palette_rgb": co.palette().tolist(),
        "palette_src": "infur/src/decode_predict.rs:9-30",
    }
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)

    # ---- oracle tables ----
    np.savez_compressed(os.path.join(HERE, "oracle_tables.npz"), preproc_lut=co.preproc_lut(),
                        color_lut=co.color_lut())

    # ---- small end-to-end vectors ----
    blob = W.synth_blob()
    sha = hashlib.sha256(blob).hexdigest()
    assert co.model_load(blob) == 0
    tm = TorchModel(blob)
    out = {"blob_sha256": np.frombuffer(sha.encode(), np.uint8), "seed": np.array([W.DEFAULT_SEED], np.uint64)}
    for (h, w) in ((48, 64), (61, 97)):
        tag = f"{w}x{h}"
        fr = W.synth_frame(h, w, index=0)
        chw = co.pack_normalize(fr)
        r = co.model_forward(chw)
        tl, ta = tm.forward_lowres(chw)
        for a, b in ((r["out_low"], tl.numpy()), (r["aux_low"], ta.numpy())):
            rel = np.abs(a - b).max() / np.abs(b).max()
            assert rel < 1e-4, f"C oracle and torch oracle disagree: {rel}"
        rc, rgba = co.frame_advance(fr)
        assert rc == 0
        out[f"bgr_{tag}"] = fr
        out[f"chw_{tag}"] = chw
        out[f"out_low_{tag}"] = r["out_low"]
        out[f"aux_low_{tag}"] = r["aux_low"]
        out[f"out_{tag}"] = r["out"]
        out[f"aux_{tag}"] = r["aux"]
        out[f"rgba_{tag}"] = rgba
        for mode, mn in ((0, "nearest"), (1, "bilinear")):
            for fac in (0.5, 0.37, 1.7):
                rc, sc = co.scale(fr, fac, mode)
                assert rc == 0
                out[f"scale_{mn}_{fac}_{tag}"] = sc
    # colorcode edge cases: ties, all-negative, NaN, (0,1), > 1, inf
    rng = np.random.default_rng(7)
    cc = rng.normal(0.3, 0.6, size=(21, 16, 24)).astype(np.float32)
    cc[:, 0, 0] = -1.0          # all negative -> class 0, alpha 0
    cc[:, 0, 1] = 0.75          # exact ties -> first wins
    cc[:, 0, 2] = np.nan        # NaN never wins
    cc[3, 0, 3] = np.inf
    cc[5, 0, 4] = 1.0           # exactly 1.0 -> 255
    cc[:, 0, 5] = 0.0           # all zero: strict > keeps class 0 / alpha 0
    cc[20, 0, 6] = 7.5
    out["cc_in"] = cc
    out["cc_rgba"] = co.colorcode(cc)
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **out)
    print("wrote fixtures; blob sha256", sha)


if __name__ == "__main__":
    main()
