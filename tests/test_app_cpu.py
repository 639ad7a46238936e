"""Host-side semantics of the frame sources and VideoPlayer (no GPU needed)."""
import io

import numpy as np
import pytest

from infur_amd.app import RawVideoSource, SyntheticSource, VideoCmd, VideoPlayer, VideoProcError
from infur_amd.processors import Slot
from infur_amd.weights import synth_frame


def raw_clip(w, h, n):
    frames = [synth_frame(h, w, index=i) for i in range(n)]
    return frames, io.BytesIO(b"".join(f.tobytes() for f in frames))


def test_raw_video_source_reads_bgr24_wire_format():
    """ff-video/src/decoder.rs:156-165: read_exact(W*H*3), ids start at 1, EOF = FinishedNormally."""
    frames, bio = raw_clip(16, 12, 3)
    src = RawVideoSource(bio, 16, 12)
    img = src.empty_image()
    for i, f in enumerate(frames):
        assert src.read_frame(img) == i + 1 and (img == f).all()
    with pytest.raises(VideoProcError) as e:
        src.read_frame(img)
    assert e.value.kind == "FinishedNormally"
    short = RawVideoSource(io.BytesIO(frames[0].tobytes()[:-5]), 16, 12)
    with pytest.raises(VideoProcError) as e:
        short.read_frame(img)
    assert e.value.kind == "ExactReadError"


def test_video_player_semantics():
    """processing.rs:96-139: Play / Pause / Stop, dirty flag, frame buffer reuse, close at EOF."""
    vp = VideoPlayer()
    out = Slot()
    assert not vp.is_dirty()
    vp.advance((), out)
    assert out.value is None  # no video: nothing written
    frames, bio = raw_clip(16, 12, 2)
    vp.control(VideoCmd.Play(RawVideoSource(bio, 16, 12)))
    assert vp.is_dirty()
    vp.advance((), out)
    assert out.value.id == 1 and (out.value.img == frames[0]).all()
    buf = out.value.img
    vp.control(VideoCmd.Pause(True))
    assert not vp.is_dirty()
    vp.advance((), out)
    assert out.value.id == 1  # paused: frame untouched
    vp.control(VideoCmd.Pause(False))
    vp.advance((), out)
    assert out.value.id == 2 and out.value.img is buf and (buf == frames[1]).all()  # buffer reused
    with pytest.raises(VideoProcError):
        vp.advance((), out)  # EOF: error relayed, player closes itself
    assert vp.vid is None and not vp.is_dirty() and out.value.id == 2
    # size change re-creates the image
    vp.control(VideoCmd.Play(SyntheticSource(8, 6)))
    vp.advance((), out)
    assert out.value.img.shape == (6, 8, 3) and out.value.id == 1
    vp.control(VideoCmd.Stop())
    assert vp.vid is None
