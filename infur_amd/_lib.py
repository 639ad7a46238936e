"""ctypes binding of ``libinfur_hip.so`` (the C ABI declared in include/infur_hip.h).

There is no fallback: if the shared library is missing or does not load, importing
the product path fails loudly with instructions to build it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# INFUR_LIB_PATH: an instrumentation build of the same ABI (scripts/ktrace.py: `make ktrace` -> libinfur_hip_ktrace.so)
LIB_PATH = os.environ.get("INFUR_LIB_PATH") or os.path.join(_HERE, "libinfur_hip.so")

ABI_VERSION = 6  # INFUR_ABI_VERSION of include/infur_hip.h

# status codes (include/infur_hip.h)
OK = 0
E_INVALID_SCALE = 1
E_ZERO_SIZE_IN = 2
E_ZERO_SIZE_OUT = 3
E_SHAPE = 4
E_MODEL_NOT_LOADED = 5
E_MODEL_FORMAT = 6
E_HIP = 7
E_RCCL = 8
E_INVALID_ARG = 9
E_IO = 10
E_CAPACITY = 11

SCALE_NEAREST, SCALE_BILINEAR = 0, 1
DTYPE_F32 = 0
DTYPE_F16 = 1
DTYPE_F32_SPLIT = 2
DTYPE_F32_SPLIT_FP8 = 3  # split mode with the cross terms on the fp8 MX MFMA ("f32x")
DTYPE_F16_HL = 5  # three-byte tensors (f16 hi + e5m2 lo planes), two MFMA units per product ("f16hl")


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("compute_dtype", C.c_uint32),
        ("compute_aux", C.c_uint32),
        ("profile", C.c_uint32),
        ("keep_activations", C.c_uint32),
        ("winograd_min_cin", C.c_uint32),
        ("winograd_tile", C.c_uint32),
        ("no_autotune", C.c_uint32),
        ("no_fuse_downsample", C.c_uint32),
        ("no_fuse_stem_pool", C.c_uint32),
        ("no_fuse_b2b", C.c_uint32),
        ("stream", C.c_void_p),
    ]


class ModelInfoC(C.Structure):
    _fields_ = [
        ("input_name", C.c_char * 32),
        ("input0_dtype", C.c_char * 16),
        ("output_names", (C.c_char * 32) * 2),
        ("n_outputs", C.c_uint32),
        ("num_classes", C.c_uint32),
        ("depth", C.c_uint32),
        ("n_convs", C.c_uint32),
        ("weight_bytes", C.c_uint64),
        ("quantised", C.c_uint32),        # ABI 4
        ("resize_u8_heads", C.c_uint32),  # ABI 4
    ]


class KernelRecord(C.Structure):
    _fields_ = [
        ("name", C.c_char * 48),
        ("kernel", C.c_char * 32),
        ("ms", C.c_float),
        ("flops", C.c_double),
        ("bytes", C.c_double),
        ("algo_flops", C.c_double),
    ]


_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p
_sz = C.c_size_t
_u32 = C.c_uint32
_f = C.c_float

# every symbol include/infur_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "infur_abi_version": (C.c_uint32, []),
    "infur_status_string": (C.c_char_p, [C.c_int32]),
    "infur_device_count": (C.c_int32, []),
    "infur_options_default": (None, [C.POINTER(Options)]),
    "infur_ctx_create": (C.c_int32, [C.POINTER(Options), C.POINTER(_vp)]),
    "infur_ctx_destroy": (None, [_vp]),
    "infur_last_error": (C.c_char_p, [_vp]),
    "infur_ctx_synchronize": (C.c_int32, [_vp]),
    "infur_ctx_stream": (_vp, [_vp]),
    "infur_scale_validate": (C.c_int32, [_f]),
    "infur_scale_out_dims": (C.c_int32, [_u32, _u32, _f, _u32p, _u32p]),
    "infur_scale": (C.c_int32, [_vp, _vp, _u32, _u32, _f, _u32, _vp, _sz, _u32p, _u32p]),
    "infur_scale_dev": (C.c_int32, [_vp, _vp, _u32, _u32, _f, _u32, _vp, _sz, _u32p, _u32p]),
    "infur_model_load": (C.c_int32, [_vp, C.c_char_p]),
    "infur_onnx_to_blob": (C.c_int32, [_vp, _sz, C.POINTER(_vp), C.POINTER(_sz), C.c_char_p, _sz]),
    "infur_buffer_free": (None, [_vp]),
    "infur_model_load_blob": (C.c_int32, [_vp, _vp, _sz]),
    "infur_model_load_blob_dev": (C.c_int32, [_vp, _vp, _sz]),
    "infur_model_unload": (C.c_int32, [_vp]),
    "infur_model_info_get": (C.c_int32, [_vp, C.POINTER(ModelInfoC)]),
    "infur_model_info_get_sized": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "infur_model_advance": (C.c_int32, [_vp, _vp, _u32, _u32, _vp, _vp, _u32p]),
    "infur_model_advance_dev": (C.c_int32, [_vp, _vp, _u32, _u32, _vp, _vp, _u32p]),
    "infur_model_warmup": (C.c_int32, [_vp, _u32, _u32]),
    "infur_ctx_set_graph_replay": (C.c_int32, [_vp, _u32]),
    "infur_ctx_graph_stats": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(_u32)]),
    "infur_model_lowres_dims": (C.c_int32, [_u32, _u32, _u32p, _u32p]),
    "infur_model_read_lowres": (C.c_int32, [_vp, _vp, _vp, _u32p, _u32p]),
    "infur_debug_read_activation": (C.c_int32, [_vp, _u32, _vp, _sz, _u32p, _u32p, _u32p]),
    "infur_pack_normalize": (C.c_int32, [_vp, _vp, _u32, _u32, _vp]),
    "infur_pack_normalize_dev": (C.c_int32, [_vp, _vp, _u32, _u32, _vp]),
    "infur_colorcode": (C.c_int32, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "infur_colorcode_dev": (C.c_int32, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "infur_bgr_to_rgba": (C.c_int32, [_vp, _vp, _u32, _u32, _vp]),
    "infur_bgr_to_rgba_dev": (C.c_int32, [_vp, _vp, _u32, _u32, _vp]),
    "infur_frame_advance": (C.c_int32, [_vp, _vp, _u32, _u32, _f, _u32, _vp, _sz, _vp, _u32p, _u32p]),
    "infur_frame_advance_dev": (C.c_int32, [_vp, _vp, _u32, _u32, _f, _u32, _vp, _sz, _vp, _u32p, _u32p]),
    "infur_stream_create": (C.c_int32, [_vp, _u32, C.POINTER(_vp)]),
    "infur_stream_destroy": (None, [_vp]),
    "infur_stream_add_lane": (C.c_int32, [_vp, _vp]),
    "infur_stream_submit": (C.c_int32, [_vp, _vp, _u32, _u32, _f, _u32, C.c_uint64]),
    "infur_stream_pending": (C.c_uint32, [_vp]),
    "infur_stream_next_dims": (C.c_int32, [_vp, C.POINTER(C.c_uint64), _u32p, _u32p]),
    "infur_stream_collect": (C.c_int32, [_vp, _vp, _sz, _vp, C.POINTER(C.c_uint64), _u32p, _u32p]),
    "infur_stream_acquire": (C.c_int32, [_vp, _u32, _u32, _f, C.POINTER(_vp)]),
    "infur_stream_commit": (C.c_int32, [_vp, _u32, _u32, _f, _u32, C.c_uint64]),
    "infur_stream_collect_view": (C.c_int32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint64), _u32p, _u32p]),
    "infur_stream_release": (C.c_int32, [_vp]),
    "infur_stream_abandon": (C.c_int32, [_vp]),
    "infur_host_alloc": (C.c_int32, [_sz, C.POINTER(_vp)]),
    "infur_host_free": (C.c_int32, [_vp]),
    "infur_host_is_pinned": (C.c_uint32, [_vp]),
    "infur_batch_advance": (C.c_int32, [_vp, C.POINTER(_vp), _u32p, _u32p, _u32, _f, _u32, C.POINTER(_vp),
                                        C.POINTER(_sz), _u32p, _u32p]),
    "infur_group_create": (C.c_int32, [C.POINTER(_vp), _u32, C.POINTER(_vp)]),
    "infur_group_destroy": (None, [_vp]),
    "infur_group_last_error": (C.c_char_p, [_vp]),
    "infur_group_size": (C.c_uint32, [_vp]),
    "infur_group_uses_rccl": (C.c_uint32, [_vp]),
    "infur_group_worker_numa_node": (C.c_int32, [_vp, _u32]),
    "infur_group_weights_broadcast": (C.c_int32, [_vp, _u32]),
    "infur_group_batch_advance": (C.c_int32, [_vp, C.POINTER(_vp), _u32p, _u32p, _u32, _f, _u32, C.POINTER(_vp),
                                              C.POINTER(_sz), _u32p, _u32p]),
    "infur_weights_broadcast": (C.c_int32, [C.POINTER(_vp), _u32]),
    "infur_batch_advance_multi": (C.c_int32, [C.POINTER(_vp), _u32, C.POINTER(_vp), _u32p, _u32p, _u32, _f, _u32,
                                              C.POINTER(_vp), C.POINTER(_sz), _u32p, _u32p]),
    "infur_split_range": (C.c_int32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _u32p]),
    "infur_tune_export": (C.c_int32, [_vp, C.c_char_p, _sz, C.POINTER(_sz)]),
    "infur_tune_import": (C.c_int32, [_vp, C.c_char_p, _sz]),
    "infur_profile_enable": (C.c_int32, [_vp, _u32]),
    "infur_profile_count": (C.c_int32, [_vp, _u32p]),
    "infur_profile_get": (C.c_int32, [_vp, _u32, C.POINTER(KernelRecord)]),
    "infur_dev_alloc": (C.c_int32, [_vp, _sz, C.POINTER(_vp)]),
    "infur_dev_free": (C.c_int32, [_vp, _vp]),
    "infur_memcpy_h2d": (C.c_int32, [_vp, _vp, _vp, _sz]),
    "infur_memcpy_d2h": (C.c_int32, [_vp, _vp, _vp, _sz]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP extension; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C infur_amd/csrc). "
            "infur_amd has no CPU fallback."
        )
    # PyTorch wheels bundle their own HIP/HSA runtime (soname libamdhip64.so.7, same as
    # /opt/rocm's).  Two runtimes in one process cannot both own the GPU, so when torch is
    # installed it is imported FIRST: the loader then binds libinfur_hip.so's libamdhip64.so.7
    # dependency to the copy torch already mapped and the process has a single runtime
    # (needed for torch.distributed/RCCL next to our kernels).  The same holds for librccl.so.1,
    # which libinfur_hip.so links for infur_group_* (torch bundles its own copy under that soname).
    # Without torch the system libraries from the library's RUNPATH are used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.infur_abi_version() != ABI_VERSION:
        raise ImportError("libinfur_hip.so ABI version mismatch")
    _lib = lib
    return lib


def status_string(code: int) -> str:
    return load().infur_status_string(code).decode()
