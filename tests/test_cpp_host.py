"""Builds tests/cpp/host_test.cpp against include/infur_processor.hpp + libinfur_hip.so (g++,
no HIP headers needed: the boundary is plain C) and runs it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "host_test")
    lib = os.path.join(ROOT, "infur_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_test.cpp"), "-o", out,
                           "-L", lib, "-linfur_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_cpp_host_cpu(host_bin):
    r = subprocess.run([host_bin, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cpu ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_gpu(host_bin, blob50, tmp_path):
    p = tmp_path / "fcn50.infurw"
    p.write_bytes(blob50)
    r = subprocess.run([host_bin, "gpu", str(p)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gpu ok" in r.stdout, r.stdout + r.stderr


def _build(tmp, src, name):
    out = str(tmp / name)
    lib = os.path.join(ROOT, "infur_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, src),
                           "-o", out, "-L", lib, "-linfur_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    return out


@pytest.fixture(scope="module")
def pipeline_test_bin(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("cpp_pipe"), os.path.join("tests", "cpp", "pipeline_test.cpp"), "pipeline_test")


@pytest.fixture(scope="module")
def pipeline_cli(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("cpp_cli"), os.path.join("tools", "infur_pipeline.cpp"), "infur_pipeline")


def test_cpp_pipeline_cpu(pipeline_test_bin):
    """include/infur_pipeline.hpp: frame sources and the VideoPlayer state machine need no GPU"""
    r = subprocess.run([pipeline_test_bin, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cpu ok" in r.stdout, r.stdout + r.stderr


def test_cpp_pipeline_cli_fails_loudly_without_gpu_or_arguments(pipeline_cli, tmp_path):
    r = subprocess.run([pipeline_cli], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([pipeline_cli, "--width", "8", "--height", "8", "--model", "m", "--scale", "-1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "scale" in r.stderr.lower()
    import torch
    if not torch.cuda.is_available():  # no CPU fallback: the context cannot be created
        r = subprocess.run([pipeline_cli, "--width", "8", "--height", "8", "--model", "m", "--synthetic", "1"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "HIP" in r.stderr


@pytest.mark.gpu
def test_cpp_pipeline_gpu(pipeline_test_bin, blob50, tmp_path):
    """the reference's app tests (app.rs:175-253) on the C++ ProcessingApp; fused == unfused == streamed"""
    p = tmp_path / "fcn50.infurw"
    p.write_bytes(blob50)
    r = subprocess.run([pipeline_test_bin, "gpu", str(p)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gpu ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_pipeline_cli_matches_python_path(pipeline_cli, blob50, ctx, tmp_path):
    """bgr24 clip on stdin -> RGBA masks on stdout: the native front end writes the bytes the Python path computes"""
    import numpy as np

    from infur_amd import weights as W
    from infur_amd.processors import FramePath, Model, ModelCmd

    p = tmp_path / "fcn50.infurw"
    p.write_bytes(blob50)
    frames = [W.synth_frame(96, 160, index=i) for i in range(5)]
    clip = b"".join(f.tobytes() for f in frames)
    Model(ctx).control(ModelCmd.LoadBlob(blob50))
    fp = FramePath(ctx)
    want = b"".join(np.ascontiguousarray(fp.advance(f, 0.5)[0]).tobytes() for f in frames)
    for extra in ([], ["--lanes", "2", "--depth", "3"], ["--app"], ["--copy"], ["--copy", "--lanes", "2", "--depth", "3"]):
        r = subprocess.run([pipeline_cli, "--width", "160", "--height", "96", "--scale", "0.5", "--model", str(p)] + extra,
                           input=clip + (b"" if extra else b"\x01\x02\x03"), capture_output=True, timeout=300)
        if extra:
            assert r.returncode == 0, r.stderr.decode()
        else:  # a truncated trailing frame is reported (ExactReadError) after the complete ones were written
            assert r.returncode == 1 and b"short read" in r.stderr
        assert r.stdout == want, (extra, len(r.stdout), len(want))
        assert b"5 frames" in r.stderr
