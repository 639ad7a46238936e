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
