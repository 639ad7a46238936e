"""Headless counterpart of the reference's processing graph and frame sources.

* ``VideoPlayer``   -- infur/src/processing.rs:62-139 over pluggable frame sources.  The reference
  reads raw ``bgr24`` frames from an ffmpeg child's stdout with ``read_exact(W*H*3)``
  (ff-video/src/decoder.rs:53-64,156-165); ``RawVideoSource`` reads exactly that wire format from
  any file object / pipe, ``SyntheticSource`` generates the deterministic test frames (ffmpeg and
  the lavfi ``testsrc`` clips of infur-test-gen are not available here).
* ``ProcessingApp`` -- infur/src/app.rs:51-158: vid -> scale -> model -> decode(out[0]) plus the
  BGR -> RGBA display copy, same commands, same dirty/id semantics, same errors.
* ``StreamPath``    -- the bounded-queue streaming mode (infur/src/main.rs:27-99,105) over
  ``infur_stream_*``: uploads, kernels and downloads of consecutive frames overlap.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import BinaryIO, Callable, List, Optional, Union

import numpy as np

from . import _lib
from .processors import (ColorCode, Context, Frame, FramePath, InfurError, Model, ModelCmd, ModelInfo, Processor,
                         Scale, Slot, bgr_image)


# --------------------------------------------------------------------------- #
# frame sources
# --------------------------------------------------------------------------- #
class VideoProcError(InfurError):
    """ff-video/src/error.rs: errors while reading frames."""

    def __init__(self, kind: str, detail: str = ""):
        Exception.__init__(self, detail or kind)
        self.kind = kind
        self.code = -1
        self.detail = detail


class FFVideoError(InfurError):
    """ff-video/src/error.rs: the source could not be opened / closed."""

    def __init__(self, detail: str):
        Exception.__init__(self, detail)
        self.code = -1
        self.detail = detail


class FrameSource:
    """What ``FFMpegDecoder`` is to the reference's VideoPlayer: dims + read_frame + close."""

    width: int
    height: int
    frame_counter: int = 0

    def empty_image(self) -> np.ndarray:
        return bgr_image(self.width, self.height)

    def read_frame(self, image: np.ndarray) -> int:
        raise NotImplementedError

    def close(self) -> None:
        pass


class RawVideoSource(FrameSource):
    """Packed bgr24 frames of known WxH from a binary stream (file, pipe, socket file).

    This is the output of ``ffmpeg -i <in> -an -f image2pipe -fflags nobuffer -pix_fmt bgr24
    -c:v rawvideo pipe:1`` (decoder.rs:53-64); plug a real ffmpeg in with
    ``RawVideoSource(subprocess.Popen(cmd, stdout=PIPE).stdout, w, h)``.
    """

    def __init__(self, stream: BinaryIO, width: int, height: int, close_stream: bool = True):
        if width <= 0 or height <= 0:
            raise FFVideoError(f"bad video dimensions {width}x{height}")
        self.stream, self.width, self.height = stream, width, height
        self.frame_counter = 0
        self._close_stream = close_stream

    def read_frame(self, image: np.ndarray) -> int:
        view = memoryview(image).cast("B")
        n, got = len(view), 0
        while got < n:  # read_exact
            k = self.stream.readinto(view[got:])
            if not k:
                break
            got += k
        if got == 0:
            raise VideoProcError("FinishedNormally", "video stream ended")
        if got < n:
            raise VideoProcError("ExactReadError", f"short frame: {got} of {n} bytes")
        self.frame_counter += 1  # decoder.rs:163-164: ids start at 1
        return self.frame_counter

    def close(self) -> None:
        if self._close_stream:
            try:
                self.stream.close()
            except Exception as e:  # pragma: no cover
                raise FFVideoError(str(e))


class SyntheticSource(FrameSource):
    """Deterministic frames (infur_amd.weights.synth_frame); ``n_frames=None`` never ends."""

    def __init__(self, width: int, height: int, n_frames: Optional[int] = None, seed_offset: int = 0):
        self.width, self.height, self.n_frames, self.seed_offset = width, height, n_frames, seed_offset
        self.frame_counter = 0

    def read_frame(self, image: np.ndarray) -> int:
        from .weights import synth_frame

        if self.n_frames is not None and self.frame_counter >= self.n_frames:
            raise VideoProcError("FinishedNormally", "video stream ended")
        image[...] = synth_frame(self.height, self.width, index=self.seed_offset + self.frame_counter)
        self.frame_counter += 1
        return self.frame_counter


# --------------------------------------------------------------------------- #
# VideoPlayer (processing.rs:62-139)
# --------------------------------------------------------------------------- #
@dataclass
class VideoCmd:
    kind: str
    source: Optional[Union[FrameSource, Callable[[], FrameSource]]] = None
    paused: bool = False

    @staticmethod
    def Play(source) -> "VideoCmd":
        """Start or restart playing from ``source`` (a FrameSource or a factory of one)."""
        return VideoCmd("play", source=source)

    @staticmethod
    def Pause(paused: bool) -> "VideoCmd":
        return VideoCmd("pause", paused=paused)

    @staticmethod
    def Stop() -> "VideoCmd":
        return VideoCmd("stop")


class VideoPlayer(Processor):
    """Writes video frames at command (processing.rs:73-139). Input = (), Output = Option<Frame>."""

    def __init__(self):
        self.vid: Optional[FrameSource] = None
        self.paused = False

    def close_video(self):
        v, self.vid = self.vid, None
        if v is not None:
            v.close()

    def control(self, cmd: VideoCmd) -> "VideoPlayer":
        if cmd.kind == "play":
            self.close_video()
            src = cmd.source() if callable(cmd.source) else cmd.source
            if not isinstance(src, FrameSource):
                raise FFVideoError("Play needs a FrameSource")
            self.vid = src
        elif cmd.kind == "pause":
            self.paused = cmd.paused
        elif cmd.kind == "stop":
            self.close_video()
        return self

    def is_dirty(self) -> bool:
        return (not self.paused) and self.vid is not None  # processing.rs:110-112

    def advance(self, _inp, out: Slot) -> None:
        if self.paused or self.vid is None:
            return
        vid = self.vid
        fr = out.value
        if fr is None:
            fr = Frame(0, vid.empty_image())
            out.value = fr
        elif fr.img.shape[0] != vid.height or fr.img.shape[1] != vid.width:
            fr.img = vid.empty_image()  # re-create on size change only (processing.rs:121-131)
        try:
            fr.id = vid.read_frame(fr.img)
        except VideoProcError as e:
            if e.kind == "FinishedNormally":
                self.close_video()  # processing.rs:133-135
            raise


# --------------------------------------------------------------------------- #
# ProcessingApp (app.rs)
# --------------------------------------------------------------------------- #
@dataclass
class AppCmd:
    kind: str
    arg: object = None

    @staticmethod
    def Video(cmd: VideoCmd) -> "AppCmd":
        return AppCmd("video", cmd)

    @staticmethod
    def Scale(factor: float) -> "AppCmd":
        return AppCmd("scale", factor)

    @staticmethod
    def Model(cmd: ModelCmd) -> "AppCmd":
        return AppCmd("model", cmd)

    @staticmethod
    def Exit() -> "AppCmd":
        return AppCmd("exit")


@dataclass
class GUIFrame:
    """app.rs:64-69: frame id, the scaled frame as RGBA for display, the optional mask."""

    id: int
    buffer: np.ndarray  # [h, w, 4] u8, r,g,b,255
    decoded_buffer: Optional[np.ndarray]  # [h, w, 4] u8 premultiplied, or None without a model

    @property
    def size(self):
        return [self.buffer.shape[1], self.buffer.shape[0]]  # ColorImage.size = [w, h]


@dataclass
class AppInfo:
    model_info: Optional[ModelInfo]


class ProcessingApp(Processor):
    """vid -> scale -> model -> decode, one frame per ``generate()`` (app.rs:84-158).

    ``fused=True`` (default) takes the device-resident route for the model + decode step
    (``infur_frame_advance`` semantics: no full-resolution logits, app.rs only decodes out[0],
    :116); ``fused=False`` chains the three processors exactly as app.rs:112-123 does.  Both
    produce identical masks (tested).
    """

    def __init__(self, ctx: Context, fused: bool = True, scale_mode: int = _lib.SCALE_NEAREST):
        self.ctx = ctx
        self.fused = fused
        self.vid = VideoPlayer()
        self.scale = Scale(ctx, scale_mode)
        self.model = Model(ctx)
        self.decoder = ColorCode(ctx)
        self.frame = Slot()
        self.scaled_frame = Slot()
        self.decoded_img = Slot()
        self.to_exit = False
        self._unit = FramePath(ctx, scale_mode)

    def info(self) -> AppInfo:
        return AppInfo(self.model.get_info())

    def control(self, cmd: AppCmd) -> "ProcessingApp":
        if cmd.kind == "video":
            self.vid.control(cmd.arg)
        elif cmd.kind == "scale":
            self.scale.control(cmd.arg)
        elif cmd.kind == "exit":
            self.to_exit = True
        elif cmd.kind == "model":
            self.model.control(cmd.arg)
        return self

    def is_dirty(self) -> bool:
        return self.vid.is_dirty() or self.scale.is_dirty()  # app.rs:155-157

    def advance(self, _inp=None, _out=None) -> Optional[GUIFrame]:
        self.vid.advance((), self.frame)  # app.rs:108
        if self.is_dirty():  # only Scale is gated (app.rs:109-111)
            self.scale.advance(self.frame.value, self.scaled_frame)
        sf = self.scaled_frame.value
        if sf is None:
            return None
        if self.fused:
            rgba, _ = self._unit.advance(sf.img, 1.0)  # scaled frame -> mask, nothing at full res
            self.decoded_img.value = rgba  # None when no model is loaded (app.rs:127-129)
        else:
            out: List[np.ndarray] = []
            self.model.advance(sf.img, out)  # runs on every generate(), dirty or not (app.rs:113-114)
            if out:
                self.decoder.advance(out[0], self.decoded_img)  # only out[0] (app.rs:116)
            else:
                self.decoded_img.value = None
        h, w = sf.img.shape[:2]
        buf = np.empty((h, w, 4), np.uint8)
        self.ctx.check(self.ctx.L.infur_bgr_to_rgba(self.ctx.h, np.ascontiguousarray(sf.img).ctypes.data, w, h,
                                                    buf.ctypes.data))
        dec = self.decoded_img.value
        return GUIFrame(sf.id, buf, None if dec is None else dec.copy())

    def generate(self) -> Optional[GUIFrame]:
        return self.advance()


# --------------------------------------------------------------------------- #
# streaming (main.rs:27-99: bounded frame channel, back-pressure)
# --------------------------------------------------------------------------- #
class StreamPath:
    """Ring of ``depth`` in-flight frames (the reference's ``sync_channel(2)``, main.rs:105)."""

    def __init__(self, ctx: Context, depth: int = 2, scale_mode: int = _lib.SCALE_NEAREST):
        self.ctx, self.depth, self.scale_mode = ctx, depth, scale_mode
        h = C.c_void_p(None)
        ctx.check(ctx.L.infur_stream_create(ctx.h, depth, C.byref(h)))
        self.h = h

    def add_lane(self, other: Context) -> None:
        """A second context of the same device takes every other frame (kernels of consecutive frames overlap)."""
        self.ctx.check(self.ctx.L.infur_stream_add_lane(self.h, other.h))

    def pending(self) -> int:
        return self.ctx.L.infur_stream_pending(self.h)

    def submit(self, img: np.ndarray, factor: float, frame_id: int) -> None:
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        self.ctx.check(self.ctx.L.infur_stream_submit(self.h, img.ctypes.data, w, h, float(np.float32(factor)),
                                                      self.scale_mode, frame_id))

    def collect(self, want_scaled: bool = False):
        """-> (frame_id, rgba [oh,ow,4], scaled bgr or None) of the oldest pending frame."""
        L = self.ctx.L
        fid, ow, oh = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(L.infur_stream_next_dims(self.h, C.byref(fid), C.byref(ow), C.byref(oh)))
        rgba = np.empty((oh.value, ow.value, 4), np.uint8)
        scaled = np.empty((oh.value, ow.value, 3), np.uint8) if want_scaled else None
        self.ctx.check(L.infur_stream_collect(self.h, rgba.ctypes.data, rgba.nbytes,
                                              scaled.ctypes.data if want_scaled else None, C.byref(fid), C.byref(ow),
                                              C.byref(oh)))
        return fid.value, rgba, scaled

    # ---- zero-copy ingest / egress (ABI 5): the ring's pinned slots lent to the caller ----
    def acquire(self, w: int, h: int, factor: float) -> np.ndarray:
        """-> the next slot's pinned input buffer as an [h, w, 3] u8 array to fill in place (what the reference's decoder does with its
        reused BgrImage, ff-video/src/decoder.rs:156-165); then ``commit``.

        LIFETIME (this array and the views of ``collect_view``): numpy views over pinned memory the LIBRARY owns.  The input view is
        valid until ``commit`` / ``abandon`` / the next ``acquire``; the output views until ``release`` (or a copying ``collect`` of the same
        frame).  After that -- and after ``close()``, a re-acquire with another frame size (the slot is re-allocated) or a failed batch call
        that drops the ring -- they dangle: copy what has to outlive the slot (``run_zero_copy`` yields copies for that reason)."""
        p = C.c_void_p(None)
        self.ctx.check(self.ctx.L.infur_stream_acquire(self.h, w, h, float(np.float32(factor)), C.byref(p)))
        buf = (C.c_uint8 * (w * h * 3)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(h, w, 3)

    def commit(self, w: int, h: int, factor: float, frame_id: int) -> None:
        self.ctx.check(self.ctx.L.infur_stream_commit(self.h, w, h, float(np.float32(factor)), self.scale_mode, frame_id))

    def abandon(self) -> None:
        """Give an acquired slot back uncommitted (end of input / read error after ``acquire``); idempotent."""
        self.ctx.check(self.ctx.L.infur_stream_abandon(self.h))

    def collect_view(self, want_scaled: bool = False):
        """-> (frame_id, rgba view [oh,ow,4], scaled view or None): arrays over the pinned output slot, valid until ``release``."""
        L = self.ctx.L
        fid, ow, oh = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        pr, ps = C.c_void_p(None), C.c_void_p(None)
        self.ctx.check(L.infur_stream_collect_view(self.h, C.byref(pr), C.byref(ps) if want_scaled else None, C.byref(fid), C.byref(ow), C.byref(oh)))
        n = ow.value * oh.value
        rgba = np.frombuffer((C.c_uint8 * (n * 4)).from_address(pr.value), np.uint8).reshape(oh.value, ow.value, 4)
        scaled = None
        if want_scaled and ps.value:
            scaled = np.frombuffer((C.c_uint8 * (n * 3)).from_address(ps.value), np.uint8).reshape(oh.value, ow.value, 3)
        return fid.value, rgba, scaled

    def release(self) -> None:
        self.ctx.check(self.ctx.L.infur_stream_release(self.h))

    def run_zero_copy(self, frames, factor: float, fill=None):
        """As ``run`` through acquire / commit / collect_view / release: ``fill(slot, item)`` writes the frame into the pinned slot (default:
        ``slot[...] = item``, one copy -- a real producer read()s the pipe into the slot instead); yields (id, COPY of the mask)."""
        fill = fill or (lambda slot, img: np.copyto(slot, img))
        for fid, img in frames:
            if self.pending() >= self.depth:
                i, rgba, _ = self.collect_view()
                out = rgba.copy()
                self.release()
                yield i, out
            h, w = img.shape[:2]
            slot = self.acquire(w, h, factor)
            try:
                fill(slot, img)
            except BaseException:
                self.abandon()  # a producer that fails after acquire must not leave the ring unusable for submit()
                raise
            self.commit(w, h, factor, fid)
        while self.pending():
            i, rgba, _ = self.collect_view()
            out = rgba.copy()
            self.release()
            yield i, out

    def run(self, frames, factor: float):
        """Generator: push frames (iterable of (id, img)) through the ring, yield (id, rgba) in order."""
        for fid, img in frames:
            if self.pending() >= self.depth:
                yield self.collect()[:2]
            self.submit(img, factor, fid)
        while self.pending():
            yield self.collect()[:2]

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.infur_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedArray:
    """A numpy array over pinned host memory from ``infur_host_alloc``: frames / masks the caller owns and reuses; the batch calls
    (``FramePath.advance_batch``, ``Group.batch_advance``) move such buffers by DMA without their staging copies."""

    def __init__(self, shape, dtype=np.uint8):
        self.L = _lib.load()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p(None)
        rc = self.L.infur_host_alloc(n, C.byref(p))
        if rc != 0:
            raise MemoryError(f"infur_host_alloc({n}) failed: {_lib.status_string(rc)}")
        self.p = p
        self.array = np.frombuffer((C.c_uint8 * n).from_address(p.value), dtype).reshape(shape)

    def close(self):
        if getattr(self, "p", None) is not None and self.p:
            self.array = None
            self.L.infur_host_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
