"""SURVEY section 5 (sanitizers) / VERDICT r2 item 6: everything behind `ModelCmd::Load` that parses untrusted bytes on the
host -- the ONNX wire-format reader (onnx_reader.cpp) and the INFURW01 header / directory checks (blob_dir.h, the code
model_load_dev runs before it touches the GPU) -- built with -fsanitize=address,undefined (`make asan`) and driven with
>= 10,000 seeded mutations (bit flips, random bytes, length-field extremes, truncations with the cut-off tail poisoned) of
  * the REAL output of PyTorch's ONNX exporter for the torchvision-shaped FCN-ResNet50 (tests/tv_fcn.py, 141 MB), and
  * an INFURW01 blob.
Every mutation must end in a format error with a message or in an accepted file whose blob passes the blob checks: no
crash, no sanitizer report (-fno-sanitize-recover aborts on the first one).  predict_onnx.rs:288-309: a load error is a
`Result`, never fatal."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "infur_amd", "csrc")
BIN = os.path.join(CSRC, "build", "fuzz_formats_asan")

N_BLOB, N_ONNX_PER_PROC, ONNX_PROCS = 8000, 700, 3  # 8000 + 2100 mutations
N_QBLOB, N_QONNX_PER_PROC = 4000, 500                # + 4000 + 1000 on the quantised formats


@pytest.fixture(scope="module")
def harness():
    subprocess.run(["make", "-C", CSRC, "-s", "asan"], check=True, timeout=600)
    assert os.path.exists(BIN)
    return BIN


def counts(out):
    kv = dict(p.split("=") for p in out.strip().splitlines()[-1].split())
    return int(kv["accepted"]), int(kv["rejected"])


def test_blob_header_and_directory_survive_mutation(harness, blob50, tmp_path):
    p = tmp_path / "r50.blob"
    p.write_bytes(blob50)
    r = subprocess.run([harness, "blob", str(p), str(N_BLOB), "0x1F0A2026"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    acc, rej = counts(r.stdout)
    assert acc + rej == N_BLOB and rej > 200 and acc > 200, (acc, rej)  # both outcomes are exercised


def test_exporter_file_survives_mutation(harness, exported50, tmp_path):
    _, _, model = exported50
    p = tmp_path / "exported50.onnx"
    p.write_bytes(model)
    procs = [subprocess.Popen([harness, "onnx", str(p), str(N_ONNX_PER_PROC), str(1000 + 17 * k)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for k in range(ONNX_PROCS)]
    tot_acc = tot_rej = 0
    for pr in procs:
        out, err = pr.communicate(timeout=1500)
        assert pr.returncode == 0, out[-1000:] + err[-4000:]
        a, r = counts(out)
        tot_acc += a
        tot_rej += r
    print(f"exporter file: {tot_acc} mutations accepted, {tot_rej} rejected with a format error")
    assert tot_acc + tot_rej == N_ONNX_PER_PROC * ONNX_PROCS and tot_rej > 300 and tot_acc > 100
    assert N_BLOB + N_ONNX_PER_PROC * ONNX_PROCS >= 10000


def test_quantised_blob_and_qoperator_file_survive_mutation(harness, tmp_path):
    """the INFURQ01 directory checks and the QOperator graph walker (onnx_qreader.cpp), same harness"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_writer as OW
    from test_onnx_quant_cpu import random_qmodel

    from infur_amd import weights as W

    specs, convs, adds = random_qmodel(seed=11)
    pb = tmp_path / "r50.qblob"
    pb.write_bytes(W.pack_qblob(convs, adds, 50, 21, True))
    po = tmp_path / "r50_int8.onnx"
    po.write_bytes(OW.fcn_qmodel(convs, adds, specs))
    pq = tmp_path / "r50_int8_qdq.onnx"  # the QDQ form of the same model: the fusion pass in front of the walker
    for c in convs[:12]:
        c.y_zp = 0 if c.name.endswith(("conv1", "conv2")) else c.y_zp
    pq.write_bytes(OW.fcn_qmodel(convs, adds, specs, qdq=True, resize_u8=True, resize_subgraph=True))
    procs = [subprocess.Popen([harness, "onnx", str(po if k == 0 else pq), str(N_QONNX_PER_PROC), str(5000 + 31 * k)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for k in range(2)]
    r = subprocess.run([harness, "blob", str(pb), str(N_QBLOB), "0xC0FFEE"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    acc, rej = counts(r.stdout)
    assert acc + rej == N_QBLOB and rej > 100 and acc > 100, (acc, rej)
    tot_acc = tot_rej = 0
    for pr in procs:
        out, err = pr.communicate(timeout=1500)
        assert pr.returncode == 0, out[-1000:] + err[-4000:]
        a, rj = counts(out)
        tot_acc += a
        tot_rej += rj
    print(f"QOperator file: {tot_acc} mutations accepted, {tot_rej} rejected with a format error")
    assert tot_acc + tot_rej == 2 * N_QONNX_PER_PROC and tot_rej > 150 and tot_acc > 30
