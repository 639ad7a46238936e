"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module, and only as the checker -- never the product path
(``infur_amd/``).  Two independent restatements live here:

* ``COracle``  -- ctypes binding of ``libinfur_oracle.so`` (oracle/infur_oracle.c), the
  plain-C restatement of the whole path (scale, pre-proc, FCN-ResNet forward,
  ColorCode).  Citations to the reference are in the C source.
* ``torch_forward`` -- the same FCN-ResNet graph through torch-CPU functional ops
  (oneDNN convs).  It cross-checks the C oracle's network (different summation
  order, so tolerance-checked) and is the strong CPU timing baseline.

PARITY STATUS: ColorCode KATs and Scale dims/errors are pinned by the reference's
tests; the network forward, nearest sampling rule, epaint premultiply and Resize
coordinate rule are restatements of third-party code and are "parity unpinned"
against ONNX Runtime (see oracle/infur_oracle.h and DESIGN.md); the network is
corroborated against torch's own module graph of the exported model
(tests/test_onnx_exporter_cpu.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libinfur_oracle.so")

OK, E_INVALID_SCALE, E_ZERO_SIZE_IN, E_ZERO_SIZE_OUT, E_SHAPE, E_MODEL_FORMAT = 0, 1, 2, 3, 4, 6


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "infur_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def _p(a: Optional[np.ndarray], t):
    return a.ctypes.data_as(t) if a is not None else None


class COracle:
    """ctypes binding of the C restatement."""

    def __init__(self, threads: int = 0):
        build()
        L = self.L = C.CDLL(_LIB)
        L.oracle_scale_validate.argtypes = [C.c_float]
        L.oracle_scale_out_dims.argtypes = [C.c_uint32, C.c_uint32, C.c_float, _u32p, _u32p]
        L.oracle_scale.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, _u8p, _u32p, _u32p]
        L.oracle_preproc_lut.argtypes = [_f32p]
        L.oracle_pack_normalize.argtypes = [_u8p, C.c_uint32, C.c_uint32, _f32p]
        L.oracle_pack_u8.argtypes = [_u8p, C.c_uint32, C.c_uint32, _f32p]
        L.oracle_palette.argtypes = [_u8p]
        L.oracle_color32_from_rgba_unmultiplied.argtypes = [C.c_uint8] * 4 + [_u8p]
        L.oracle_color_code.argtypes = [C.c_size_t, C.c_float, _u8p]
        L.oracle_colorcode.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p]
        L.oracle_argmax.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, _u8p]
        L.oracle_model_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.oracle_model_free.argtypes = [C.c_void_p]
        L.oracle_model_num_classes.argtypes = [C.c_void_p]
        L.oracle_model_lowres_dims.argtypes = [C.c_uint32, C.c_uint32, _u32p, _u32p]
        L.oracle_model_forward.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_uint32, _f32p, _f32p, _f32p, _f32p]
        L.oracle_upsample_bilinear.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, _f32p, C.c_uint32, C.c_uint32]
        L.oracle_frame_advance.argtypes = [C.c_void_p, _u8p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, _u8p, _u32p, _u32p]
        L.oracle_set_threads.argtypes = [C.c_int]
        for f in ("oracle_preproc_lut", "oracle_pack_normalize", "oracle_pack_u8", "oracle_palette", "oracle_color32_from_rgba_unmultiplied",
                  "oracle_color_code", "oracle_colorcode", "oracle_argmax", "oracle_model_free", "oracle_model_lowres_dims",
                  "oracle_upsample_bilinear", "oracle_set_threads"):
            getattr(L, f).restype = None
        if threads:
            L.oracle_set_threads(threads)
        self._model = None

    # ---- Scale ----
    def scale_validate(self, factor: float) -> int:
        return self.L.oracle_scale_validate(factor)

    def scale_out_dims(self, w: int, h: int, factor: float) -> Tuple[int, int, int]:
        ow, oh = C.c_uint32(0), C.c_uint32(0)
        rc = self.L.oracle_scale_out_dims(w, h, factor, C.byref(ow), C.byref(oh))
        return rc, ow.value, oh.value

    def scale(self, bgr: np.ndarray, factor: float, mode: int = 0) -> Tuple[int, Optional[np.ndarray]]:
        h, w = bgr.shape[:2]
        rc = self.scale_validate(factor)
        if rc:
            return rc, None
        rc, ow, oh = self.scale_out_dims(w, h, factor)
        if rc:
            return rc, None
        out = np.empty((oh, ow, 3), np.uint8)
        a, b = C.c_uint32(0), C.c_uint32(0)
        src = np.ascontiguousarray(bgr)
        rc = self.L.oracle_scale(_p(src, _u8p), w, h, factor, mode, _p(out, _u8p), C.byref(a), C.byref(b))
        return rc, out

    # ---- pre-proc ----
    def preproc_lut(self) -> np.ndarray:
        lut = np.empty(768, np.float32)
        self.L.oracle_preproc_lut(_p(lut, _f32p))
        return lut.reshape(3, 256)

    def pack_normalize(self, bgr: np.ndarray) -> np.ndarray:
        h, w = bgr.shape[:2]
        out = np.empty((3, h, w), np.float32)
        src = np.ascontiguousarray(bgr)
        self.L.oracle_pack_normalize(_p(src, _u8p), w, h, _p(out, _f32p))
        return out

    def pack_u8(self, bgr: np.ndarray) -> np.ndarray:
        """The input of a Uint8-input model (predict_onnx.rs:114-122): planes B, G, R holding the bytes."""
        h, w = bgr.shape[:2]
        out = np.empty((3, h, w), np.float32)
        src = np.ascontiguousarray(bgr)
        self.L.oracle_pack_u8(_p(src, _u8p), w, h, _p(out, _f32p))
        return out

    # ---- ColorCode ----
    def palette(self) -> np.ndarray:
        p = np.empty(60, np.uint8)
        self.L.oracle_palette(_p(p, _u8p))
        return p.reshape(20, 3)

    def from_rgba_unmultiplied(self, r, g, b, a) -> np.ndarray:
        o = np.empty(4, np.uint8)
        self.L.oracle_color32_from_rgba_unmultiplied(r, g, b, a, _p(o, _u8p))
        return o

    def color_code(self, klass: int, alpha: float) -> np.ndarray:
        o = np.empty(4, np.uint8)
        self.L.oracle_color_code(klass, alpha, _p(o, _u8p))
        return o

    def color_lut(self) -> np.ndarray:
        """[20,256,4] premultiplied RGBA for (palette entry, alpha byte)."""
        pal = self.palette()
        lut = np.empty((20, 256, 4), np.uint8)
        for k in range(20):
            for a in range(256):
                lut[k, a] = self.from_rgba_unmultiplied(int(pal[k, 0]), int(pal[k, 1]), int(pal[k, 2]), a)
        return lut

    def colorcode(self, khw: np.ndarray) -> np.ndarray:
        k, h, w = khw.shape
        src = np.ascontiguousarray(khw, np.float32)
        out = np.empty((h, w, 4), np.uint8)
        self.L.oracle_colorcode(_p(src, _f32p), k, h, w, _p(out, _u8p))
        return out

    def argmax(self, khw: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        k, h, w = khw.shape
        src = np.ascontiguousarray(khw, np.float32)
        kl = np.empty((h, w), np.uint8)
        al = np.empty((h, w), np.uint8)
        self.L.oracle_argmax(_p(src, _f32p), k, h, w, _p(kl, _u8p), _p(al, _u8p))
        return kl, al

    # ---- model ----
    def model_load(self, blob: bytes) -> int:
        self.model_unload()
        h = C.c_void_p(None)
        rc = self.L.oracle_model_load(blob, len(blob), C.byref(h))
        if rc == 0:
            self._model = h
            self.num_classes = self.L.oracle_model_num_classes(h)
        return rc

    def model_unload(self):
        if self._model is not None:
            self.L.oracle_model_free(self._model)
            self._model = None

    def lowres_dims(self, h: int, w: int) -> Tuple[int, int]:
        a, b = C.c_uint32(0), C.c_uint32(0)
        self.L.oracle_model_lowres_dims(h, w, C.byref(a), C.byref(b))
        return a.value, b.value

    def model_forward(self, chw: np.ndarray, full: bool = True, low: bool = True):
        """-> dict(out, aux, out_low, aux_low) of f32 arrays (None where not requested)."""
        assert self._model is not None
        _, h, w = chw.shape
        lh, lw = self.lowres_dims(h, w)
        K = self.num_classes
        r = {
            "out": np.empty((K, h, w), np.float32) if full else None,
            "aux": np.empty((K, h, w), np.float32) if full else None,
            "out_low": np.empty((K, lh, lw), np.float32) if low else None,
            "aux_low": np.empty((K, lh, lw), np.float32) if low else None,
        }
        src = np.ascontiguousarray(chw, np.float32)
        rc = self.L.oracle_model_forward(self._model, _p(src, _f32p), h, w, _p(r["out"], _f32p), _p(r["aux"], _f32p),
                                         _p(r["out_low"], _f32p), _p(r["aux_low"], _f32p))
        if rc:
            raise RuntimeError(f"oracle_model_forward rc={rc}")
        return r

    def upsample_bilinear(self, x: np.ndarray, oh: int, ow: int) -> np.ndarray:
        k, ih, iw = x.shape
        src = np.ascontiguousarray(x, np.float32)
        out = np.empty((k, oh, ow), np.float32)
        self.L.oracle_upsample_bilinear(_p(src, _f32p), k, ih, iw, _p(out, _f32p), oh, ow)
        return out

    def frame_advance(self, bgr: np.ndarray, factor: float = 1.0, scale_mode: int = 0):
        assert self._model is not None
        h, w = bgr.shape[:2]
        rc, ow, oh = self.scale_out_dims(w, h, factor)
        if rc:
            return rc, None
        out = np.empty((oh, ow, 4), np.uint8)
        a, b = C.c_uint32(0), C.c_uint32(0)
        src = np.ascontiguousarray(bgr)
        rc = self.L.oracle_frame_advance(self._model, _p(src, _u8p), w, h, factor, scale_mode, _p(out, _u8p), C.byref(a), C.byref(b))
        return rc, out

    def __del__(self):
        try:
            self.model_unload()
        except Exception:
            pass


# --------------------------------------------------------------------------- #
# torch-CPU restatement of the network (independent summation order)
# --------------------------------------------------------------------------- #
class TorchModel:
    """FCN-ResNet forward through torch CPU functional ops, fed by the INFURW01 blob.

    Replaces ``session.run`` (infur/src/predict_onnx.rs:138).  Architecture =
    torchvision fcn_resnet50/101 with BN folded (see infur_amd/weights.py::graph).
    """

    def __init__(self, blob: bytes, threads: int = 0, float64: bool = False):
        """``float64``: parameters and arithmetic in double precision -- the reference the hostile-parameter tests grade
        both the f32 oracle and the HIP modes against (tests/test_gpu_hostile.py)."""
        import torch

        from infur_amd import weights as W

        self.torch = torch
        if threads:
            torch.set_num_threads(threads)
        meta, tensors = W.unpack_blob(blob)
        self.meta = meta
        self.specs = W.graph(meta["depth"], meta["num_classes"], meta["aux"])
        assert len(self.specs) == len(tensors)
        self.params = []
        for spec, (name, w, b) in zip(self.specs, tensors):
            assert spec.name == name and w.shape == (spec.cout, spec.cin, spec.k, spec.k), name
            self.params.append((torch.from_numpy(np.array(w)), torch.from_numpy(np.array(b))))
        self.float64 = float64
        if float64:
            self.params = [(w.double(), b.double()) for w, b in self.params]

    def forward_lowres(self, chw: np.ndarray, taps=None):
        """[3,h,w] f32 -> (out_low [K,lh,lw], aux_low or None) as torch tensors.

        ``taps``: optional dict filled with per-conv outputs (name -> tensor [C,H,W]).
        """
        torch = self.torch
        F = torch.nn.functional
        x = torch.from_numpy(np.ascontiguousarray(chw, np.float32))[None]
        if self.float64:
            x = x.double()
        it = iter(zip(self.specs, self.params))

        def conv(x, residual=None):
            s, (w, b) = next(it)
            y = F.conv2d(x, w, b, stride=s.stride, padding=s.pad, dilation=s.dil)
            if residual is not None:
                y = y + residual
            if s.relu:
                y = F.relu(y)
            if taps is not None:
                taps[s.name] = y[0]
            return y, s

        with torch.no_grad():
            x, _ = conv(x)
            x = F.max_pool2d(x, 3, 2, 1)
            l3 = None
            specs = self.specs
            i = 1
            while specs[i].role == "conv1":
                has_down = specs[i + 3].role == "down"
                t, _ = conv(x)
                t, _ = conv(t)
                if has_down:
                    # blob order: conv1, conv2, conv3, downsample -- evaluate the identity first
                    s3, p3 = next(it)
                    sd, (wd, bd) = next(it)
                    idt = F.conv2d(x, wd, bd, stride=sd.stride)
                    if taps is not None:
                        taps[sd.name] = idt[0]
                    y = F.relu(F.conv2d(t, p3[0], p3[1]) + idt)
                    if taps is not None:
                        taps[s3.name] = y[0]
                    i += 4
                else:
                    y, s3 = conv(t, residual=x)
                    i += 3
                x = y
                if s3.name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
                    l3 = x
            h, _ = conv(x)
            out_low, _ = conv(h)
            aux_low = None
            if self.meta["aux"]:
                a, _ = conv(l3)
                aux_low, _ = conv(a)
                aux_low = aux_low[0]
        return out_low[0], aux_low

    def forward(self, chw: np.ndarray):
        """-> (out [K,h,w], aux [K,h,w]) numpy, torch's own bilinear (align_corners=False)."""
        F = self.torch.nn.functional
        _, h, w = chw.shape
        ol, al = self.forward_lowres(chw)
        out = F.interpolate(ol[None], size=(h, w), mode="bilinear", align_corners=False)[0].numpy()
        aux = None
        if al is not None:
            aux = F.interpolate(al[None], size=(h, w), mode="bilinear", align_corners=False)[0].numpy()
        return out, aux
