"""a NodeProto that lost its outputs (or inputs) must end in a format error, never in an out-of-range index
behind the C ABI -- `ModelCmd::Load` failures are a `Result` in the reference (infur/src/predict_onnx.rs:288-309).  The
byte-level mutation harness (tests/test_formats_asan_cpu.py) rarely produces this shape by chance (it needs a tag byte to
turn into a different, still well-formed field), so here it is produced on purpose: for EVERY operator type of the float
file, of the QOperator int8 file and of its QDQ form, the first and the last node of that type are re-serialised once
without outputs and once without inputs, and the file goes through infur_onnx_to_blob.  No GPU."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import onnx_writer as OW  # noqa: E402
from test_onnx_cpu import convert  # noqa: E402
from test_onnx_quant_cpu import random_qmodel  # noqa: E402

from infur_amd import weights as W


def _fields(buf):
    """[(field, wire type, raw bytes of the whole field, payload)] of one protobuf message"""
    out, p = [], 0
    while p < len(buf):
        start, key, sh = p, 0, 0
        while True:
            b = buf[p]
            p += 1
            key |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        field, wt = key >> 3, key & 7
        if wt == 0:
            while buf[p] & 0x80:
                p += 1
            p += 1
            payload = None
        elif wt == 1:
            p += 8
            payload = None
        elif wt == 5:
            p += 4
            payload = None
        elif wt == 2:
            n, sh = 0, 0
            while True:
                b = buf[p]
                p += 1
                n |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            payload = buf[p:p + n]
            p += n
        else:
            raise ValueError(wt)
        out.append((field, wt, buf[start:p], payload))
    return out


def node_ops(model: bytes):
    graph = next(pl for f, wt, _, pl in _fields(model) if f == 7 and wt == 2)
    ops = []
    for f, wt, _, pl in _fields(graph):
        if f == 1 and wt == 2:
            ops.append(next(p2 for f2, w2, _, p2 in _fields(pl) if f2 == 4 and w2 == 2).decode())
    return ops


def strip_node(model: bytes, index: int, drop_field: int) -> bytes:
    """the same ModelProto with node `index` (position in graph.node) re-serialised without its field 1 (inputs) / 2 (outputs)"""
    top = []
    for f, wt, raw, pl in _fields(model):
        if not (f == 7 and wt == 2):
            top.append(raw)
            continue
        g, k = [], 0
        for f1, w1, raw1, pl1 in _fields(pl):
            if f1 == 1 and w1 == 2:
                if k == index:
                    raw1 = OW.f_bytes(1, b"".join(r for f2, _, r, _ in _fields(pl1) if f2 != drop_field))
                k += 1
            g.append(raw1)
        top.append(OW.f_bytes(7, b"".join(g)))
    return b"".join(top)


def _models(blob50):
    tensors = W.unpack_blob(blob50)[1]
    specs, convs, adds = random_qmodel(seed=5)
    yield "float", OW.fcn_model(tensors, W.graph(50))[0]
    yield "qoperator", OW.fcn_qmodel(convs, adds, specs)
    for c in convs:  # (a Relu in front of a QuantizeLinear is the clamp only at zero point 0)
        if c.name.endswith(("conv1", "conv2")) or c.name in ("backbone.conv1", "classifier.0", "aux_classifier.0"):
            c.y_zp = 0
    yield "qdq", OW.fcn_qmodel(convs, adds, specs, qdq=True, resize_u8=True, resize_subgraph=True)


def test_nodes_without_outputs_or_inputs_are_format_errors(lib, blob50):
    n_cases = 0
    for kind, model in _models(blob50):
        rc, err, out = convert(lib, model)
        assert rc == 0 and out is not None, (kind, err)
        ops = node_ops(model)
        first, last = {}, {}
        for i, op in enumerate(ops):
            first.setdefault(op, i)
            last[op] = i
        for op in sorted(first):
            for idx in sorted({first[op], last[op]}):
                for drop, what in ((2, "outputs"), (1, "inputs")):
                    if drop == 1 and op == "Constant":
                        continue  # (a Constant has no inputs to lose)
                    bad = strip_node(model, idx, drop)
                    assert len(bad) < len(model)
                    rc, err, out = convert(lib, bad)
                    assert rc != 0 and out is None and err, (kind, op, idx, what)
                    n_cases += 1
    assert n_cases >= 60


@pytest.mark.parametrize("op", ["QuantizeLinear", "MaxPool"])
def test_the_two_reported_crashes(lib, blob50, op):
    """the advisor's reproductions: a QuantizeLinear / MaxPool of the QOperator file with an empty output list"""
    specs, convs, adds = random_qmodel(seed=6)
    model = OW.fcn_qmodel(convs, adds, specs)
    idx = node_ops(model).index(op)
    rc, err, out = convert(lib, strip_node(model, idx, 2))
    assert rc != 0 and out is None and "no outputs" in err, err
