#!/usr/bin/env python3
"""Golden vectors of the quantised path: the integer oracle (oracle/infur_qoracle.py) on tests/hostile_q.py's models -- both are
pure numpy / exact integer arithmetic, so the hashes are portable.  Pins the oracle (and the generator) against silent change; the
GPU path is compared with the oracle bit for bit in tests/test_gpu_quant.py, so these vectors pin it transitively.
    python tests/golden/make_int8_golden.py      (rewrites tests/golden/int8_golden.json)"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hostile_q import hostile_qblob  # noqa: E402
from infur_amd import quantize, weights as W  # noqa: E402
from oracle import infur_qoracle as Q  # noqa: E402


def case(seed, h, w, index):
    blob = hostile_qblob(seed=seed)
    lo, aux = Q.qforward(blob, quantize.normalise(W.synth_frame(h, w, index=index)))
    return {"model": f"tests/hostile_q.py hostile_qblob(seed={seed})", "blob_sha1": hashlib.sha1(blob).hexdigest(),
            "frame": f"weights.synth_frame({h}, {w}, index={index})", "out_low_shape": list(lo.shape),
            "out_low_sha1": hashlib.sha1(lo.tobytes()).hexdigest(), "aux_low_sha1": hashlib.sha1(aux.tobytes()).hexdigest(),
            "out_low_first8": [float(v) for v in lo.ravel()[:8]], "argmax_histogram": np.bincount(lo.argmax(0).ravel(), minlength=21).tolist()}


def build():
    return {"generated_by": "tests/golden/make_int8_golden.py", "cases": [case(0, 40, 56, 2), case(1, 33, 47, 5)]}


if __name__ == "__main__":
    json.dump(build(), open(os.path.join(ROOT, "tests", "golden", "int8_golden.json"), "w"), indent=1)
    print("wrote tests/golden/int8_golden.json")
