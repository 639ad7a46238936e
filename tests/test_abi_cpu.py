"""CPU tests of the C-ABI library: it loads, exports every declared symbol, its host-only
logic (Scale validation / dims, status strings) behaves like the reference, and it fails
loudly -- never falls back -- when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from infur_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "infur_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(infur_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/infur_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.infur_abi_version() == _lib.ABI_VERSION


def test_struct_sizes_match_header(lib):
    o = _lib.Options()
    lib.infur_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(_lib.Options) and o.compute_aux == 1 and o.device == 0


def test_status_strings_are_the_reference_messages(lib):
    # thiserror messages, processing.rs:163,203,205 and predict_onnx.rs:35
    assert _lib.status_string(_lib.E_INVALID_SCALE) == "Cannot scale by negative number"
    assert _lib.status_string(_lib.E_ZERO_SIZE_IN) == "scaling from 0-sized input"
    assert _lib.status_string(_lib.E_ZERO_SIZE_OUT) == "scaling to 0-sized output"
    assert _lib.status_string(_lib.E_SHAPE) == "couldn't transform image"


def dims(lib, w, h, f):
    ow, oh = C.c_uint32(0), C.c_uint32(0)
    rc = lib.infur_scale_out_dims(w, h, f, C.byref(ow), C.byref(oh))
    return rc, ow.value, oh.value


def test_scale_host_logic_matches_reference_kats(lib, kats):
    k = kats["scale_from_size0"]
    assert lib.infur_scale_validate(k["factor"]) == 0
    assert dims(lib, k["w"], k["h"], k["factor"])[0] == _lib.E_ZERO_SIZE_IN
    k = kats["scale_to_size0"]
    assert dims(lib, k["w"], k["h"], k["factor"])[0] == _lib.E_ZERO_SIZE_OUT
    for f in kats["valid_scale_rejects"]["factors"]:
        assert lib.infur_scale_validate(f) == _lib.E_INVALID_SCALE
    assert lib.infur_scale_validate(float("nan")) == 0
    assert dims(lib, 10, 10, float("nan"))[0] == _lib.E_ZERO_SIZE_OUT  # (w as f32 * NaN) as u32 == 0
    for d in kats["scale_dims"]:
        assert dims(lib, d["w"], d["h"], d["factor"]) == (0, d["ow"], d["oh"])
    assert dims(lib, 0, 10, 1.0) == (0, 0, 10)  # unit scale clones before any size check
    assert dims(lib, 1920, 1080, 0.5) == (0, 960, 540)


def test_scale_host_logic_matches_oracle(lib, oracle):
    rng = np.random.default_rng(3)
    for _ in range(300):
        w, h = int(rng.integers(0, 4000)), int(rng.integers(0, 3000))
        f = float(np.float32(rng.choice([rng.uniform(1e-6, 3.0), 1.0, 0.5, 1e-9, 7.77])))
        assert dims(lib, w, h, f) == oracle.scale_out_dims(w, h, f) or dims(lib, w, h, f)[0] == oracle.scale_out_dims(w, h, f)[0] != 0


def test_lowres_dims(lib, oracle):
    for h, w in ((1080, 1920), (540, 960), (480, 640), (240, 320), (61, 97), (48, 64), (1, 1), (2160, 3840)):
        a, b = C.c_uint32(0), C.c_uint32(0)
        assert lib.infur_model_lowres_dims(h, w, C.byref(a), C.byref(b)) == 0
        assert (a.value, b.value) == oracle.lowres_dims(h, w)


def test_no_gpu_fails_loudly(lib):
    """Without a HIP device the context cannot be created: an error, not a CPU fallback."""
    if lib.infur_device_count() > 0:
        pytest.skip("a GPU is visible: the loud-failure path is covered on CPU-only hosts")
    from infur_amd.processors import Context, InfurError

    with pytest.raises(InfurError) as e:
        Context(device=0)
    assert e.value.code == _lib.E_HIP and "no CPU fallback" in str(e.value)


def test_product_path_does_not_import_the_oracle():
    """infur_amd/ and include/ must never reference oracle/ (the oracle is the checker only)."""
    bad = []
    for base in ("infur_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            if "build" in dp or "__pycache__" in dp:
                continue
            for f in fs:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(import\s+oracle|from\s+oracle|infur_oracle\.h|libinfur_oracle)", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_library_does_not_link_rccl():
    """librccl is resolved with dlopen by the first group that needs a communicator (infur_multi.cpp), so that the
    single-GPU Processor path loads on a host without RCCL -- the library's only NEEDED accelerator runtime is libamdhip64"""
    import subprocess

    from infur_amd import _lib

    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    needed = [ln.split("[")[1].rstrip("]") for ln in out.splitlines() if "(NEEDED)" in ln]
    assert any(n.startswith("libamdhip64") for n in needed), needed
    assert not any("rccl" in n or "nccl" in n or "torch" in n for n in needed), needed
