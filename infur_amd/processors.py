"""Host-side mirror of the reference's ``Processor`` plugin surface over the C ABI.

The reference defines one trait, ``Processor`` (infur/src/processing.rs:23-60), and the
per-frame path instantiates it three times: ``Scale`` (processing.rs:179-282),
``Model<f32>`` (infur/src/predict_onnx.rs:146-345) and ``ColorCode``
(infur/src/decode_predict.rs:38-84), wired together by ``ProcessingApp::advance``
(infur/src/app.rs:107-153).  The classes below keep the same names, commands, argument
meaning and error behaviour; the arithmetic happens in ``libinfur_hip.so`` (hand-written
gfx950 kernels).  There is no CPU fallback.

Rust ``&mut Output`` parameters become mutable holders: ``Slot`` for ``Option<T>``
outputs, a plain ``list`` for ``Vec<ArrayD<f32>>``.  Typed ``Result`` errors become
exceptions carrying the status code of include/infur_hip.h.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Generic, List, Optional, TypeVar

import numpy as np

from . import _lib

T = TypeVar("T")

# tile configurations measured on MI355X for the BASELINE configs (scripts/tune.py writes it)
TUNE_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_tune_gfx950.txt")


# --------------------------------------------------------------------------- #
# errors (thiserror enums of the reference)
# --------------------------------------------------------------------------- #
class InfurError(Exception):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        self.detail = detail
        super().__init__(detail or _lib.status_string(code))


class ValidScaleError(InfurError):
    """processing.rs:145-168 -- "Cannot scale by negative number"."""


class ScaleProcError(InfurError):
    """processing.rs:201-211 -- ZeroSizeIn / ZeroSizeOut."""

    @property
    def kind(self) -> str:
        return {_lib.E_ZERO_SIZE_IN: "ZeroSizeIn", _lib.E_ZERO_SIZE_OUT: "ZeroSizeOut"}.get(self.code, "Other")


class ModelCmdError(InfurError):
    """predict_onnx.rs:41-48 -- the model could not be loaded."""


class ModelProcError(InfurError):
    """predict_onnx.rs:33-39 -- ShapeError / RuntimeError while processing."""


class Slot(Generic[T]):
    """A mutable ``Option<T>`` the callee may fill or reuse (Rust ``&mut Option<T>``)."""

    def __init__(self, value: Optional[T] = None):
        self.value = value

    def is_some(self) -> bool:
        return self.value is not None

    def take(self) -> Optional[T]:
        v, self.value = self.value, None
        return v


@dataclass
class Frame:
    """processing.rs:9-18: frame id + packed BGR image ([h, w, 3] u8, row-major, no padding)."""

    id: int
    img: np.ndarray

    def __eq__(self, other):  # PartialEq compares ids only (processing.rs:14-18)
        return isinstance(other, Frame) and self.id == other.id


def bgr_image(w: int, h: int) -> np.ndarray:
    """``BgrImage::new(w, h)``: zero-filled packed BGR."""
    return np.zeros((h, w, 3), np.uint8)


def _check_bgr(img: np.ndarray) -> np.ndarray:
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ModelProcError(_lib.E_SHAPE, f"expected packed BGR u8 [h,w,3], got {img.dtype} {img.shape}")
    return np.ascontiguousarray(img)


# --------------------------------------------------------------------------- #
# context
# --------------------------------------------------------------------------- #
class Context:
    """One GPU + one HIP stream + device arena (``infur_ctx``).  Not thread-safe."""

    def __init__(self, device: int = 0, compute_aux: bool = True, profile: bool = False,
                 keep_activations: bool = False, stream: Optional[int] = None, dtype: str = "f32",
                 winograd_min_cin: int = 0, winograd_tile: int = 0, autotune: bool = True, fuse_downsample: bool = True,
                 fuse_stem_pool: bool = True, fuse_b2b: bool = True, graph_replay: bool = False):
        L = self.L = _lib.load()
        o = _lib.Options()
        L.infur_options_default(C.byref(o))
        o.device = device
        o.compute_dtype = {"f32": _lib.DTYPE_F32, "f16": _lib.DTYPE_F16, "f32s": _lib.DTYPE_F32_SPLIT, "f32x": _lib.DTYPE_F32_SPLIT_FP8,
                           "f16hl": _lib.DTYPE_F16_HL}[dtype]
        self.dtype = dtype
        o.compute_aux = 1 if compute_aux else 0
        o.profile = 1 if profile else 0
        o.keep_activations = 1 if keep_activations else 0
        o.winograd_min_cin = winograd_min_cin  # 0 = default (256), 0xFFFFFFFF = direct convs only
        o.winograd_tile = winograd_tile  # 0 = default F(6x6,3x3); 2 / 4 / 6 = forced
        o.no_autotune = 0 if autotune else 1
        o.no_fuse_downsample = 0 if fuse_downsample else 1
        o.no_fuse_stem_pool = 0 if fuse_stem_pool else 1
        o.no_fuse_b2b = 0 if fuse_b2b else 1  # f16 mode: conv3 + residual and the next block's conv1 as one launch
        o.stream = stream
        h = C.c_void_p(None)
        rc = L.infur_ctx_create(C.byref(o), C.byref(h))
        if rc != _lib.OK:
            raise InfurError(rc, f"infur_ctx_create(device={device}) failed: {_lib.status_string(rc)} "
                                 f"({L.infur_device_count()} HIP devices visible; there is no CPU fallback)")
        self.h = h
        self.device = device
        if autotune and os.path.exists(TUNE_DB):  # measured tile configurations for the common shapes
            self.load_tuning(TUNE_DB)
        if graph_replay:  # the fused frame path as a hipGraph once a frame shape has settled (small frames: launch-bound)
            self.check(L.infur_ctx_set_graph_replay(h, 1))

    def graph_stats(self):
        """(graphs captured so far, frames replayed from a graph, graphs cached now)"""
        a, b, n = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        self.check(self.L.infur_ctx_graph_stats(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def load_tuning(self, path: str) -> None:
        txt = open(path, "rb").read()
        self.check(self.L.infur_tune_import(self.h, txt, len(txt)))

    def split_range(self):
        """dtype "f32s" only: (max |activation| fed to a GEMM, max |Winograd-domain input|, saturated) of the last frame."""
        a, w, sat = C.c_float(0), C.c_float(0), C.c_uint32(0)
        self.check(self.L.infur_split_range(self.h, C.byref(a), C.byref(w), C.byref(sat)))
        return a.value, w.value, bool(sat.value)

    def tuning_text(self) -> str:
        n = C.c_size_t(0)
        self.check(self.L.infur_tune_export(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        self.check(self.L.infur_tune_export(self.h, buf, n.value, C.byref(n)))
        return buf.raw[: n.value].decode()

    def last_error(self) -> str:
        return self.L.infur_last_error(self.h).decode()

    def check(self, rc: int, exc=InfurError):
        if rc != _lib.OK:
            raise exc(rc, self.last_error() or _lib.status_string(rc))

    def synchronize(self):
        self.check(self.L.infur_ctx_synchronize(self.h))

    @property
    def stream(self) -> int:
        return self.L.infur_ctx_stream(self.h) or 0

    def profile(self) -> List[dict]:
        """Kernel records of the last advance (needs ``profile=True``)."""
        n = C.c_uint32(0)
        self.check(self.L.infur_profile_count(self.h, C.byref(n)))
        out = []
        rec = _lib.KernelRecord()
        for i in range(n.value):
            self.check(self.L.infur_profile_get(self.h, i, C.byref(rec)))
            out.append({"name": rec.name.decode(), "kernel": rec.kernel.decode(), "ms": rec.ms,
                        "flops": rec.flops, "bytes": rec.bytes, "algo_flops": rec.algo_flops})
        return out

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.infur_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# --------------------------------------------------------------------------- #
# the trait
# --------------------------------------------------------------------------- #
class Processor:
    """processing.rs:23-60."""

    def control(self, cmd):
        raise NotImplementedError

    def advance(self, inp, out):
        raise NotImplementedError

    def is_dirty(self) -> bool:
        raise NotImplementedError


class Scale(Processor):
    """Scale frames by a constant factor (processing.rs:179-282).

    Command = f32, Input = Output = Option<Frame>.  ``mode`` selects the resampler:
    nearest is the reference's (processing.rs:189); bilinear is the north-star extension.
    """

    def __init__(self, ctx: Context, mode: int = _lib.SCALE_NEAREST):
        self.ctx = ctx
        self.mode = mode
        self.factor = np.float32(1.0)  # Default, processing.rs:185-193
        self.dirty = True

    def control(self, cmd: float) -> "Scale":
        factor = np.float32(cmd)
        rc = self.ctx.L.infur_scale_validate(float(factor))
        if rc != _lib.OK:
            raise ValidScaleError(rc)  # state untouched, like `cmd.try_into()?` (processing.rs:221)
        self.dirty = bool(factor != self.factor)  # processing.rs:222 (NaN != NaN -> dirty)
        self.factor = factor
        return self

    def is_dirty(self) -> bool:
        return self.dirty

    def is_unit_scale(self) -> bool:
        return bool(self.factor == np.float32(1.0))

    def advance(self, inp: Optional[Frame], out: Slot) -> None:
        self.dirty = False  # processing.rs:233
        if inp is None:
            return
        if self.is_unit_scale():  # clone, processing.rs:238-242
            out.value = Frame(inp.id, inp.img.copy())
            return
        img = _check_bgr(inp.img)
        h, w = img.shape[:2]
        L = self.ctx.L
        ow, oh = C.c_uint32(0), C.c_uint32(0)
        rc = L.infur_scale_out_dims(w, h, float(self.factor), C.byref(ow), C.byref(oh))
        if rc != _lib.OK:
            raise ScaleProcError(rc)
        nw, nh = ow.value, oh.value
        # get or create the output frame; re-allocate only on size change (processing.rs:260-268)
        fr = out.value
        if fr is None or fr.img.shape[0] != nh or fr.img.shape[1] != nw or not fr.img.flags["C_CONTIGUOUS"]:
            fr = Frame(inp.id, bgr_image(nw, nh))
            out.value = fr
        fr.id = inp.id
        rc = L.infur_scale(self.ctx.h, img.ctypes.data, w, h, float(self.factor), self.mode,
                           fr.img.ctypes.data, fr.img.nbytes, C.byref(ow), C.byref(oh))
        self.ctx.check(rc, ScaleProcError)


@dataclass
class ModelCmd:
    """predict_onnx.rs:267-270: ``ModelCmd::Load(path)``; empty path unloads."""

    path: str = ""
    blob: Optional[bytes] = None  # extension: load an in-memory INFURW01 blob

    @staticmethod
    def Load(path: str) -> "ModelCmd":
        return ModelCmd(path=path)

    @staticmethod
    def LoadBlob(blob: bytes) -> "ModelCmd":
        return ModelCmd(blob=blob)


@dataclass
class ModelInfo:
    """predict_onnx.rs:56-62."""

    input_names: List[str]
    input0_dtype: str
    output_names: List[str]
    num_classes: int = 0
    depth: int = 0
    weight_bytes: int = 0
    quantised: bool = False        # a QOperator / QDQ int8 model: runs on the i8 MFMA whatever the context's dtype
    resize_u8_heads: bool = False  # ... whose file resizes the u8 logits before DequantizeLinear


class Model(Processor):
    """Segmentation model session (predict_onnx.rs:146-345).

    Command = ModelCmd, Input = BgrImage, Output = Vec<ArrayD<f32>> (a ``list`` here).
    """

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def control(self, cmd: ModelCmd) -> "Model":
        L, h = self.ctx.L, self.ctx.h
        if cmd.blob is not None:
            rc = L.infur_model_load_blob(h, cmd.blob, len(cmd.blob))
        else:
            rc = L.infur_model_load(h, cmd.path.encode())  # "" unloads (predict_onnx.rs:310-312)
        self.ctx.check(rc, ModelCmdError)
        return self

    def is_dirty(self) -> bool:
        return False  # predict_onnx.rs:336-338

    def get_info(self) -> Optional[ModelInfo]:
        mi = _lib.ModelInfoC()
        rc = self.ctx.L.infur_model_info_get(self.ctx.h, C.byref(mi))
        if rc == _lib.E_MODEL_NOT_LOADED:
            return None
        self.ctx.check(rc)
        outs = [bytes(mi.output_names[i]).split(b"\0", 1)[0].decode() for i in range(mi.n_outputs)]
        return ModelInfo([mi.input_name.decode()], mi.input0_dtype.decode(), outs, mi.num_classes, mi.depth,
                         mi.weight_bytes, bool(mi.quantised), bool(mi.resize_u8_heads))

    def advance(self, img: np.ndarray, out: list) -> None:
        """Fills ``out`` with the model's outputs, each [num_classes, h, w] f32 -- [out, aux], or [out] alone for a
        model without the aux head / a context created with ``compute_aux=False`` (``get_info().output_names``);
        untouched when no model is loaded."""
        info = self.get_info()
        if info is None:
            return  # Ok(()) with `out` untouched (predict_onnx.rs:318,333)
        img = _check_bgr(img)
        h, w = img.shape[:2]
        k = info.num_classes
        bufs = [np.empty((k, h, w), np.float32) for _ in info.output_names]
        n = C.c_uint32(0)
        rc = self.ctx.L.infur_model_advance(self.ctx.h, img.ctypes.data, w, h, bufs[0].ctypes.data,
                                            bufs[1].ctypes.data if len(bufs) > 1 else None, C.byref(n))
        self.ctx.check(rc, ModelProcError)
        assert n.value == len(bufs)
        out.clear()  # predict_onnx.rs:326
        out.extend(bufs)

    def warmup(self, w: int, h: int) -> None:
        """Allocate the arena and pick tile configurations for w x h frames before the first real one."""
        self.ctx.check(self.ctx.L.infur_model_warmup(self.ctx.h, w, h), ModelProcError)

    def lowres(self):
        """Output-stride-8 logits of the last advance: (out_low, aux_low) [K, lh, lw] f32; aux_low is None for a
        one-output model."""
        info = self.get_info()
        L, h = self.ctx.L, self.ctx.h
        lh, lw = C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(L.infur_model_read_lowres(h, None, None, C.byref(lh), C.byref(lw)))
        o = np.empty((info.num_classes, lh.value, lw.value), np.float32)
        a = np.empty_like(o) if len(info.output_names) > 1 else None
        self.ctx.check(L.infur_model_read_lowres(h, o.ctypes.data, a.ctypes.data if a is not None else None,
                                                 C.byref(lh), C.byref(lw)))
        return o, a


class ColorCode(Processor):
    """Per-pixel argmax + confidence-shaded RGBA (decode_predict.rs:38-84).

    Input = Array3<f32> [K, H, W]; Output = Option<ColorImage> (``Slot`` of [H, W, 4] u8,
    premultiplied r,g,b,a -- the memory layout of epaint's ``Color32``).
    """

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def control(self, cmd=None) -> "ColorCode":
        return self

    def is_dirty(self) -> bool:
        return False

    def advance(self, inp: np.ndarray, out: Slot) -> None:
        if inp.ndim != 3:
            raise InfurError(_lib.E_SHAPE, f"expected [K,H,W], got {inp.shape}")
        khw = np.ascontiguousarray(inp, np.float32)
        k, h, w = khw.shape
        img = out.value
        if img is None or img.shape[:2] != (h, w):  # re-create only on size change (decode_predict.rs:58-65)
            img = np.zeros((h, w, 4), np.uint8)
            img[..., 3] = 255  # Color32::BLACK
            out.value = img
        self.ctx.check(self.ctx.L.infur_colorcode(self.ctx.h, khw.ctypes.data, k, h, w, img.ctypes.data))


def pack_normalize(ctx: Context, img: np.ndarray) -> np.ndarray:
    """The pre-proc stage on its own (predict_onnx.rs:103-137): BGR u8 HWC -> RGB f32 CHW."""
    img = _check_bgr(img)
    h, w = img.shape[:2]
    out = np.empty((3, h, w), np.float32)
    ctx.check(ctx.L.infur_pack_normalize(ctx.h, img.ctypes.data, w, h, out.ctypes.data))
    return out


# --------------------------------------------------------------------------- #
# the fused per-frame path (what ProcessingApp::advance does per frame, app.rs:107-153)
# --------------------------------------------------------------------------- #
class FramePath:
    """scale -> model -> decode(out[0]) in one call, nothing materialised at full resolution."""

    def __init__(self, ctx: Context, scale_mode: int = _lib.SCALE_NEAREST):
        self.ctx = ctx
        self.scale_mode = scale_mode

    def advance(self, img: np.ndarray, factor: float = 1.0, want_scaled: bool = False):
        """-> (rgba [oh,ow,4] u8 or None when no model is loaded, scaled BGR or None)."""
        img = _check_bgr(img)
        h, w = img.shape[:2]
        L = self.ctx.L
        rc = L.infur_scale_validate(float(np.float32(factor)))
        if rc:
            raise ValidScaleError(rc)
        ow, oh = C.c_uint32(0), C.c_uint32(0)
        rc = L.infur_scale_out_dims(w, h, float(np.float32(factor)), C.byref(ow), C.byref(oh))
        if rc:
            raise ScaleProcError(rc)
        rgba = np.empty((oh.value, ow.value, 4), np.uint8)
        scaled = np.empty((oh.value, ow.value, 3), np.uint8) if want_scaled else None
        rc = L.infur_frame_advance(self.ctx.h, img.ctypes.data, w, h, float(np.float32(factor)), self.scale_mode,
                                   rgba.ctypes.data, rgba.nbytes, scaled.ctypes.data if want_scaled else None,
                                   C.byref(ow), C.byref(oh))
        if rc == _lib.E_MODEL_NOT_LOADED:
            return None, scaled  # mask cleared (app.rs:127-129)
        self.ctx.check(rc)
        return rgba, scaled

    def advance_batch(self, imgs, factor: float = 1.0, outs=None):
        """A batch of independent frames (BASELINE configs[3]) -> list of masks, in order.  ``outs``: caller-owned mask arrays to fill
        (e.g. ``app.PinnedArray(...).array``: pinned frames and masks travel by DMA without the staging copies)."""
        L = self.ctx.L
        imgs = [_check_bgr(i) for i in imgs]
        n = len(imgs)
        f = float(np.float32(factor))
        given = outs
        outs = []
        for k, im in enumerate(imgs):
            ow, oh = C.c_uint32(0), C.c_uint32(0)
            rc = L.infur_scale_out_dims(im.shape[1], im.shape[0], f, C.byref(ow), C.byref(oh))
            if rc:
                raise ScaleProcError(rc)
            if given is not None:
                if given[k].shape != (oh.value, ow.value, 4) or given[k].dtype != np.uint8 or not given[k].flags.c_contiguous:
                    raise ValueError(f"outs[{k}] must be a contiguous uint8 array of shape {(oh.value, ow.value, 4)}")
                outs.append(given[k])
            else:
                outs.append(np.empty((oh.value, ow.value, 4), np.uint8))
        fp = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ws = (C.c_uint32 * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_uint32 * n)(*[im.shape[0] for im in imgs])
        caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
        self.ctx.check(L.infur_batch_advance(self.ctx.h, fp, ws, hs, n, f, self.scale_mode, op, caps, None, None))
        return outs

    def advance_dev(self, d_bgr: int, w: int, h: int, factor: float, d_rgba: int, rgba_capacity: int,
                    d_scaled: int = 0):
        """Device-resident form: pointers are raw device addresses; asynchronous on ctx.stream."""
        ow, oh = C.c_uint32(0), C.c_uint32(0)
        rc = self.ctx.L.infur_frame_advance_dev(self.ctx.h, d_bgr, w, h, float(np.float32(factor)), self.scale_mode,
                                                d_rgba, rgba_capacity, d_scaled or None, C.byref(ow), C.byref(oh))
        self.ctx.check(rc)
        return ow.value, oh.value


# --------------------------------------------------------------------------- #
# several GPUs from one process (include/infur_hip.h: infur_group_*)
# --------------------------------------------------------------------------- #
class Group:
    """``infur_group``: n contexts (one per GPU) driven together from one host process -- one worker thread per
    context, RCCL weight broadcast over xGMI, contiguous frame slices with no data-path collective
    (BASELINE configs[3]).  The reference runs all processors on one thread of one process
    (infur/src/main.rs:38-40); this is how that host reaches 8 GPUs."""

    def __init__(self, ctxs: List[Context]):
        if not ctxs:
            raise ValueError("a group needs at least one context")
        self.ctxs = list(ctxs)
        self.L = ctxs[0].L
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        g = C.c_void_p(None)
        rc = self.L.infur_group_create(arr, len(ctxs), C.byref(g))
        if rc != _lib.OK:
            raise InfurError(rc, ctxs[0].last_error())
        self.g = g

    def check(self, rc: int):
        if rc != _lib.OK:
            raise InfurError(rc, self.L.infur_group_last_error(self.g).decode() or _lib.status_string(rc))

    @property
    def uses_rccl(self) -> bool:
        return bool(self.L.infur_group_uses_rccl(self.g))

    def worker_numa_nodes(self) -> List[int]:
        """NUMA node each worker thread is pinned to (-1: not pinned)"""
        return [int(self.L.infur_group_worker_numa_node(self.g, i)) for i in range(len(self))]

    def __len__(self):
        return self.L.infur_group_size(self.g)

    def weights_broadcast(self, root: int = 0) -> None:
        self.check(self.L.infur_group_weights_broadcast(self.g, root))

    def advance_batch(self, imgs, factor: float = 1.0, scale_mode: int = _lib.SCALE_NEAREST, outs=None):
        imgs = [_check_bgr(i) for i in imgs]
        n = len(imgs)
        f = float(np.float32(factor))
        given = outs
        outs = []
        for k, im in enumerate(imgs):
            ow, oh = C.c_uint32(0), C.c_uint32(0)
            rc = self.L.infur_scale_out_dims(im.shape[1], im.shape[0], f, C.byref(ow), C.byref(oh))
            if rc:
                raise ScaleProcError(rc)
            if given is not None:
                if given[k].shape != (oh.value, ow.value, 4) or given[k].dtype != np.uint8 or not given[k].flags.c_contiguous:
                    raise ValueError(f"outs[{k}] must be a contiguous uint8 array of shape {(oh.value, ow.value, 4)}")
                outs.append(given[k])
            else:
                outs.append(np.empty((oh.value, ow.value, 4), np.uint8))
        fp = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ws = (C.c_uint32 * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_uint32 * n)(*[im.shape[0] for im in imgs])
        caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
        self.check(self.L.infur_group_batch_advance(self.g, fp, ws, hs, n, f, scale_mode, op, caps, None, None))
        return outs

    def close(self):
        if getattr(self, "g", None):
            self.L.infur_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
