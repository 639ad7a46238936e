"""Static quantisation of a float FCN-ResNet blob into the QOperator form the reference's own tests load
(`fcn-resnet50-12-int8.onnx`, infur-test-gen/build.rs:88-93): INFURW01 -> INFURQ01 (infur_amd/weights.py documents both).

What ONNX Runtime's / Neural Compressor's static quantisation tools do: u8 activations with one scale and zero point per
tensor from calibration ranges (a tensor behind a ReLU is calibrated from 0, so its zero point is 0 and the ReLU becomes the
clamp), s8 weights with one scale per output channel and zero point 0, int32 bias in units of x_scale * w_scale[o].  The
float forward used for calibration runs on the CPU (torch); nothing here touches the GPU path.  Used by bench.py's int8 side
measurement and by the tests (no pretrained or zoo file exists in this environment: the model is the seeded synthetic one)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import weights as W

f32 = np.float32


def normalise(bgr: np.ndarray) -> np.ndarray:
    """packed BGR u8 [h,w,3] -> the reference's pre-proc [3,h,w] f32, RGB planes: ((v * 1) / 255 - mean) * (1 / std) with every
    step one f32 operation in this order (infur/src/predict_onnx.rs:126-137)"""
    mean = np.array([0.485, 0.456, 0.406], f32)
    inv = f32(1.0) / np.array([0.229, 0.224, 0.225], f32)
    x = bgr[..., ::-1].transpose(2, 0, 1).astype(f32) * f32(1.0) / f32(255.0)
    return ((x - mean[:, None, None]) * inv[:, None, None]).astype(f32)


def _act_params(lo: float, hi: float) -> Tuple[float, int]:
    lo, hi = min(0.0, float(lo)), max(0.0, float(hi))
    scale = max((hi - lo) / 255.0, 1e-8)
    zp = int(np.clip(np.rint(-lo / scale), 0, 255))
    return float(f32(scale)), zp


def quantise_model(float_blob: bytes, calib_chw: List[np.ndarray]) -> bytes:
    """INFURW01 float blob + calibration inputs (normalised [3,h,w] f32) -> INFURQ01 quantised blob"""
    import torch

    F = torch.nn.functional
    meta, tensors = W.unpack_blob(float_blob)
    specs = W.graph(meta["depth"], meta["num_classes"], meta["aux"])
    params = [(torch.from_numpy(np.array(w)), torch.from_numpy(np.array(b))) for _, w, b in tensors]
    rng: Dict[str, List[float]] = {}

    def see(name, t):
        lo, hi = float(t.min()), float(t.max())
        r = rng.setdefault(name, [lo, hi])
        r[0], r[1] = min(r[0], lo), max(r[1], hi)

    with torch.no_grad():
        for chw in calib_chw:
            x = torch.from_numpy(np.ascontiguousarray(chw, np.float32))[None]
            see("input", x)
            it = iter(zip(specs, params))

            def conv(x, relu):
                s, (w, b) = next(it)
                y = F.conv2d(x, w, b, stride=s.stride, padding=s.pad, dilation=s.dil)
                if relu:
                    y = F.relu(y)
                see(s.name, y)
                return y, s

            x, _ = conv(x, True)
            x = F.max_pool2d(x, 3, 2, 1)
            i, l3, blk = 1, None, 0
            while specs[i].role == "conv1":
                has_down = specs[i + 3].role == "down"
                t, _ = conv(x, True)
                t, _ = conv(t, True)
                y3, s3 = conv(t, False)  # the QLinearConv of conv3 has no ReLU: the Add follows
                idt = x
                if has_down:
                    idt, _ = conv(x, False)
                x = F.relu(y3 + idt)
                see(f"add{blk}", x)
                blk += 1
                i += 4 if has_down else 3
                if s3.name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
                    l3 = x
            h, _ = conv(x, True)
            conv(h, False)
            if meta["aux"]:
                a, _ = conv(l3, True)
                conv(a, False)

    act = {k: _act_params(*v) for k, v in rng.items()}
    convs: List[W.QConv] = []
    adds: List[W.QAdd] = []
    # which tensor feeds each conv: walk the graph again, names only
    src_of: Dict[str, str] = {}
    cur, i, blk, l3n = "backbone.conv1", 1, 0, None
    src_of["backbone.conv1"] = "input"
    while specs[i].role == "conv1":
        has_down = specs[i + 3].role == "down"
        src_of[specs[i].name] = cur
        src_of[specs[i + 1].name] = specs[i].name
        src_of[specs[i + 2].name] = specs[i + 1].name
        if has_down:
            src_of[specs[i + 3].name] = cur
        a_p, c_p = act[specs[i + 2].name], act[f"add{blk}"]
        b_p = act[specs[i + 3].name] if has_down else act[cur]
        adds.append(W.QAdd(a_p[0], a_p[1], b_p[0], b_p[1], c_p[0], c_p[1]))
        act[f"blockout{blk}"] = c_p
        name3 = specs[i + 2].name
        cur = f"add{blk}"
        blk += 1
        i += 4 if has_down else 3
        if name3.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
            l3n = cur
    src_of[specs[i].name] = cur
    src_of[specs[i + 1].name] = specs[i].name
    if meta["aux"]:
        src_of[specs[i + 2].name] = l3n
        src_of[specs[i + 3].name] = specs[i + 2].name
    for s, (_, w, b) in zip(specs, tensors):
        xs, xz = act[src_of[s.name]]
        ys, yz = act[s.name]
        w = np.asarray(w, np.float64)
        amax = np.abs(w).reshape(s.cout, -1).max(1)
        ws = np.maximum(amax / 127.0, 1e-12).astype(f32)
        wq = np.clip(np.rint(w / ws.astype(np.float64)[:, None, None, None]), -127, 127).astype(np.int8)
        bq = np.rint(np.asarray(b, np.float64) / (np.float64(f32(xs)) * ws.astype(np.float64))).astype(np.int64)
        bq = np.clip(bq, -2**31 + 1, 2**31 - 1).astype(np.int32)
        convs.append(W.QConv(s.name, wq, ws, bq, xs, xz, ys, yz))
    return W.pack_qblob(convs, adds, meta["depth"], meta["num_classes"], meta["aux"])


def synth_qblob(depth: int = 50, calib: int = 3, size: Tuple[int, int] = (96, 128)) -> bytes:
    """the seeded synthetic model, statically quantised on `calib` synthetic frames"""
    frames = [normalise(W.synth_frame(size[0], size[1], index=100 + k)) for k in range(calib)]
    return quantise_model(W.synth_blob(depth=depth), frames)


def _load_float_blob(path: str) -> bytes:
    """an INFURW01 blob, or a float ONNX file through the library's host-only converter (no GPU needed)"""
    import ctypes as C

    data = open(path, "rb").read()
    if data[:8] == b"INFURW01":
        return data
    from . import _lib

    L = _lib.load()
    blob, n = C.c_void_p(None), C.c_size_t(0)
    err = C.create_string_buffer(512)
    if L.infur_onnx_to_blob(data, len(data), C.byref(blob), C.byref(n), err, 512) != 0:
        raise SystemExit(f"{path}: {err.value.decode()}")
    out = C.string_at(blob, n.value)
    L.infur_buffer_free(blob)
    if out[:8] != b"INFURW01":
        raise SystemExit(f"{path} is already a quantised model")
    return out


def main(argv=None) -> int:
    """python -m infur_amd.quantize model.onnx|model.blob out.qblob [--frames clip.bgr24 --width W --height H] [--calib N]

    Static quantisation of a float FCN-ResNet50/101 into an INFURQ01 file that ModelCmd::Load takes.  Calibration frames: raw bgr24
    frames (ffmpeg's `-pix_fmt bgr24 -f rawvideo`, the reference's wire format: ff-video/src/decoder.rs:53-64), the first N of the
    file; without --frames, N synthetic frames.  The float forward of the calibration runs on the CPU (torch)."""
    import argparse

    ap = argparse.ArgumentParser(prog="python -m infur_amd.quantize", description=main.__doc__)
    ap.add_argument("model")
    ap.add_argument("out")
    ap.add_argument("--frames", help="raw bgr24 clip for calibration")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--calib", type=int, default=4, help="number of calibration frames")
    a = ap.parse_args(argv)
    blob = _load_float_blob(a.model)
    frames = []
    if a.frames:
        fb = a.width * a.height * 3
        with open(a.frames, "rb") as f:
            for _ in range(a.calib):
                raw = f.read(fb)
                if len(raw) < fb:
                    break
                frames.append(np.frombuffer(raw, np.uint8).reshape(a.height, a.width, 3))
        if not frames:
            raise SystemExit(f"{a.frames} holds no complete {a.width}x{a.height} bgr24 frame")
    else:
        frames = [W.synth_frame(a.height, a.width, index=i) for i in range(a.calib)]
    q = quantise_model(blob, [normalise(fr) for fr in frames])
    with open(a.out, "wb") as f:
        f.write(q)
    meta, convs, adds = W.unpack_qblob(q)
    print(f"{a.out}: FCN-ResNet{meta['depth']}, {meta['n_convs']} QLinearConv + {meta['n_adds']} QLinearAdd, {len(q) / 1e6:.1f} MB "
          f"(float model {len(blob) / 1e6:.1f} MB), calibrated on {len(frames)} frame(s) of {a.width}x{a.height}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
