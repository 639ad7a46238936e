import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; never part of the product path)."""
    from oracle.infur_oracle import COracle

    return COracle()


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "oracle_small.npz"))


@pytest.fixture(scope="session")
def tables():
    return np.load(os.path.join(GOLDEN, "oracle_tables.npz"))


@pytest.fixture(scope="session")
def blob50():
    from infur_amd import weights as W

    return W.synth_blob()


@pytest.fixture(scope="session")
def lib():
    from infur_amd import _lib

    return _lib.load()


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0.  No skip and no fallback: without a GPU this fails loudly."""
    from infur_amd.processors import Context

    c = Context(device=0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def model(ctx, blob50):
    from infur_amd.processors import Model, ModelCmd

    m = Model(ctx)
    m.control(ModelCmd.LoadBlob(blob50))
    return m


@pytest.fixture(scope="session")
def oracle_model(oracle, blob50):
    assert oracle.model_load(blob50) == 0
    return oracle


@pytest.fixture(scope="session")
def exported50():
    """torchvision-shaped FCN-ResNet50 as torch.nn modules with the synthetic parameters UNFOLDED, and the same module
    through PyTorch's own ONNX exporter (tests/tv_fcn.py): -> (module, folded reference tensors, ModelProto bytes)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tv_fcn

    m, folded = tv_fcn.synth_fcn(50)
    return m, folded, tv_fcn.export_onnx(m)


@pytest.fixture(scope="session")
def exported50_u8():
    """The same network behind a Uint8 NHWC image input (tests/tv_fcn.py::Uint8Front), exported by PyTorch's exporter:
    -> (module taking [1,H,W,3] u8, folded reference tensors, ModelProto bytes)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tv_fcn

    m, folded = tv_fcn.synth_fcn_u8(50)
    return m, folded, tv_fcn.export_onnx(m)
