"""GPU tests of the app graph and the streaming mode: the reference's own app tests
(infur/src/app.rs:175-253) re-expressed on synthetic clips of the same dimensions, plus
fused-vs-unfused and streamed-vs-direct equality."""
import io
import os

import numpy as np
import pytest

from infur_amd import weights as W
from infur_amd.app import (AppCmd, PinnedArray, ProcessingApp, RawVideoSource, StreamPath, SyntheticSource, VideoCmd,
                           VideoProcError)
from infur_amd.processors import Context, FramePath, InfurError, ModelCmd, ValidScaleError
from infur_amd import _lib

pytestmark = pytest.mark.gpu


def short_large_input():  # 1280x720 (the doc comments in app.rs:165,169 are swapped; assertions rule)
    return SyntheticSource(1280, 720, n_frames=150)


def long_small_input():  # 640x480
    return SyntheticSource(640, 480, n_frames=400)


@pytest.fixture()
def app(ctx):
    a = ProcessingApp(ctx)
    a.control(AppCmd.Model(ModelCmd.Load("")))  # the reference's app tests run without a model
    return a


def test_void(app):  # app.rs:175-180
    assert app.generate() is None
    assert app.generate() is None


def test_scale(app):  # app.rs:182-189
    app.control(AppCmd.Video(VideoCmd.Play(short_large_input())))
    app.control(AppCmd.Scale(0.5))
    f2 = app.generate()
    assert f2.size == [1280 // 2, 720 // 2] and f2.decoded_buffer is None


def test_switch_scale(app):  # app.rs:191-201
    app.control(AppCmd.Video(VideoCmd.Play(long_small_input())))
    assert app.generate().size == [640, 480]
    app.control(AppCmd.Scale(0.5))
    assert app.generate().size == [640 // 2, 480 // 2]


def test_switch_video_then_scale(app):  # app.rs:203-218
    app.control(AppCmd.Video(VideoCmd.Play(long_small_input())))
    assert app.generate().size == [640, 480]
    app.control(AppCmd.Video(VideoCmd.Play(short_large_input())))
    assert app.generate().size == [1280, 720]
    app.control(AppCmd.Scale(2.0))
    assert app.generate().size == [1280 * 2, 720 * 2]


def test_scaled_frame_after_stopped_video(app):  # app.rs:220-236
    app.control(AppCmd.Video(VideoCmd.Play(short_large_input())))
    f1 = app.generate()
    assert f1.size == [1280, 720]
    app.control(AppCmd.Video(VideoCmd.Stop()))
    f2 = app.generate()
    assert f1.id == f2.id and not app.is_dirty()
    app.control(AppCmd.Scale(0.5))
    assert app.is_dirty()
    f3 = app.generate()
    assert f2.id == f3.id and f3.size == [1280 // 2, 720 // 2]


def test_pause_video(app):  # app.rs:238-252
    app.control(AppCmd.Video(VideoCmd.Play(long_small_input())))
    f1 = app.generate()
    app.control(AppCmd.Video(VideoCmd.Pause(True)))
    assert not app.is_dirty()
    f2 = app.generate()
    assert f1.id == f2.id and not app.is_dirty()
    app.control(AppCmd.Video(VideoCmd.Pause(False)))
    assert app.is_dirty()
    f3 = app.generate()
    assert f2.id != f3.id


def test_errors_are_relayed_not_fatal(app):  # main.rs:69-71,94-96
    with pytest.raises(ValidScaleError):
        app.control(AppCmd.Scale(-1.0))
    app.control(AppCmd.Video(VideoCmd.Play(SyntheticSource(32, 24, n_frames=1))))
    assert app.generate().id == 1
    with pytest.raises(VideoProcError):
        app.generate()  # FinishedNormally is relayed and the player closes
    f = app.generate()  # processing continues on the last frame
    assert f.id == 1 and not app.is_dirty()


def test_display_conversion(ctx):  # app.rs:132-144
    for (w, h) in ((64, 48), (97, 61), (5, 3)):
        fr = W.synth_frame(h, w, index=2)
        out = np.empty((h, w, 4), np.uint8)
        ctx.check(ctx.L.infur_bgr_to_rgba(ctx.h, fr.ctypes.data, w, h, out.ctypes.data))
        assert (out[..., 0] == fr[..., 2]).all() and (out[..., 1] == fr[..., 1]).all()
        assert (out[..., 2] == fr[..., 0]).all() and (out[..., 3] == 255).all()


def test_app_with_model_fused_equals_unfused(ctx, blob50):
    """ProcessingApp with a model: the fused route == Scale/Model/ColorCode chained as in app.rs:112-123."""
    frames = [W.synth_frame(96, 160, index=i) for i in range(3)]
    clip = b"".join(f.tobytes() for f in frames)
    masks = {}
    for fused in (True, False):
        a = ProcessingApp(ctx, fused=fused)
        a.control(AppCmd.Model(ModelCmd.LoadBlob(blob50)))
        assert a.info().model_info.output_names == ["out", "aux"]
        a.control(AppCmd.Video(VideoCmd.Play(RawVideoSource(io.BytesIO(clip), 160, 96))))
        a.control(AppCmd.Scale(0.5))
        out = []
        for _ in range(3):
            g = a.generate()
            assert g.size == [80, 48] and g.decoded_buffer.shape == (48, 80, 4)
            out.append((g.id, g.buffer.copy(), g.decoded_buffer.copy()))
        masks[fused] = out
    for (i0, b0, m0), (i1, b1, m1) in zip(masks[True], masks[False]):
        assert i0 == i1 and (b0 == b1).all() and (m0 == m1).all()
    assert [m[0] for m in masks[True]] == [1, 2, 3]


def test_streaming_equals_direct(ctx, model):
    """BASELINE configs[2] in miniature: frames pushed through the depth-2 ring come back in
    order and identical to the synchronous path; back-pressure when the ring is full."""
    frames = [(i + 1, W.synth_frame(135, 240, index=i)) for i in range(7)]
    sp = StreamPath(ctx, depth=2)
    got = list(sp.run(frames, 0.5))
    assert [g[0] for g in got] == [f[0] for f in frames]
    fp = FramePath(ctx)
    for (fid, rgba), (_, img) in zip(got, frames):
        ref, _ = fp.advance(img, 0.5)
        assert rgba.shape == (67, 120, 4) and (rgba == ref).all(), fid
    # explicit back-pressure: a third submit without collecting is refused
    sp.submit(frames[0][1], 0.5, 100)
    sp.submit(frames[1][1], 0.5, 101)
    assert sp.pending() == 2
    with pytest.raises(InfurError) as e:
        sp.submit(frames[2][1], 0.5, 102)
    assert e.value.code == _lib.E_CAPACITY
    fid, rgba, scaled = sp.collect(want_scaled=True)
    assert fid == 100 and scaled.shape == (67, 120, 3)
    assert sp.collect()[0] == 101 and sp.pending() == 0
    with pytest.raises(InfurError):
        sp.collect()
    sp.close()


def test_zero_copy_streaming_equals_copying(ctx, model):
    """VERDICT r4 item 2 (ABI 5): acquire / commit / collect_view / release lend the ring's PINNED slots to the caller -- the decoder
    fills the frame in place (ff-video/src/decoder.rs:156-165 does that with its reused BgrImage), the mask is read in place.  Same
    masks, same order as the copying calls; the two kinds may be mixed frame by frame; misuse is refused, not undefined."""
    frames = [(i + 1, W.synth_frame(135, 240, index=i)) for i in range(7)]
    sp = StreamPath(ctx, depth=2)
    ref = list(sp.run(frames, 0.5))
    got = list(sp.run_zero_copy(frames, 0.5))
    assert [g[0] for g in got] == [f[0] for f in frames]
    for (fid, a), (_, b) in zip(got, ref):
        assert a.shape == (67, 120, 4) and (a == b).all(), fid
    # a source that read()s straight into the slot: RawVideoSource.read_frame takes any writable array
    clip = io.BytesIO(b"".join(f.tobytes() for _, f in frames))
    src = RawVideoSource(clip, 240, 135)
    out = []
    for _ in frames:
        if sp.pending() >= 2:
            fid, rgba, _s = sp.collect_view()
            out.append((fid, rgba.copy()))
            sp.release()
        slot = sp.acquire(240, 135, 0.5)
        assert slot.shape == (135, 240, 3) and ctx.L.infur_host_is_pinned(slot.ctypes.data) == 1
        fid = src.read_frame(slot)
        sp.commit(240, 135, 0.5, fid)
    while sp.pending():
        fid, rgba, scaled = sp.collect_view(want_scaled=True)
        assert scaled.shape == (67, 120, 3)
        again = sp.collect_view()  # idempotent until released
        assert again[0] == fid and again[1].ctypes.data == rgba.ctypes.data
        out.append((fid, rgba.copy()))
        sp.release()
    assert [o[0] for o in out] == [f[0] for f in frames]
    for (_, a), (_, b) in zip(out, ref):
        assert (a == b).all()
    # mixed: zero-copy in, copying out; copying in, view out
    np.copyto(sp.acquire(240, 135, 0.5), frames[0][1])
    sp.commit(240, 135, 0.5, 50)
    sp.submit(frames[1][1], 0.5, 51)
    fid, rgba, _s = sp.collect()
    assert fid == 50 and (rgba == ref[0][1]).all()
    fid, view, _s = sp.collect_view()
    assert fid == 51 and (view == ref[1][1]).all()
    fid2, rgba2, _s = sp.collect()  # a copying collect of the viewed frame releases it
    assert fid2 == 51 and (rgba2 == ref[1][1]).all() and sp.pending() == 0
    # misuse
    with pytest.raises(InfurError) as e:
        sp.commit(240, 135, 0.5, 1)  # nothing acquired
    assert e.value.code == _lib.E_INVALID_ARG
    with pytest.raises(InfurError):
        sp.release()  # nothing viewed
    sp.acquire(240, 135, 0.5)
    with pytest.raises(InfurError) as e:
        sp.commit(240, 136, 0.5, 1)  # not the acquired frame size
    assert e.value.code == _lib.E_INVALID_ARG
    with pytest.raises(InfurError) as e:
        sp.submit(frames[0][1], 0.5, 2)  # a slot is acquired
    assert e.value.code == _lib.E_INVALID_ARG
    big = sp.acquire(480, 270, 1.0)  # acquiring again re-sizes the same slot
    assert big.shape == (270, 480, 3)
    np.copyto(big, W.synth_frame(270, 480, index=9))
    sp.commit(480, 270, 1.0, 7)
    np.copyto(sp.acquire(240, 135, 0.5), frames[2][1])
    sp.commit(240, 135, 0.5, 8)
    with pytest.raises(InfurError) as e:
        sp.acquire(240, 135, 0.5)  # both slots in flight
    assert e.value.code == _lib.E_CAPACITY
    fid, rgba, _s = sp.collect_view()
    assert fid == 7 and rgba.shape == (270, 480, 4) and (rgba == FramePath(ctx).advance(W.synth_frame(270, 480, index=9), 1.0)[0]).all()
    sp.release()
    assert sp.collect()[0] == 8
    with pytest.raises(InfurError) as e:
        sp.acquire(240, 135, -1.0)
    assert e.value.code == _lib.E_INVALID_SCALE
    # ABI 6: a producer that acquires and then finds no frame (EOF, read error) abandons the slot -- copying submits work
    # again; a FAILED acquire voids an earlier successful one (its commit must not pass against a slot that may have lost its buffers)
    sp.acquire(240, 135, 0.5)
    sp.abandon()
    sp.abandon()  # idempotent
    with pytest.raises(InfurError) as e:
        sp.commit(240, 135, 0.5, 3)  # nothing acquired any more
    assert e.value.code == _lib.E_INVALID_ARG
    sp.submit(frames[0][1], 0.5, 90)  # the ring is usable for copying submits again
    fid, rgba, _s = sp.collect()
    assert fid == 90 and (rgba == ref[0][1]).all()
    sp.acquire(240, 135, 0.5)
    with pytest.raises(InfurError):
        sp.acquire(240, 135, -1.0)  # fails ...
    with pytest.raises(InfurError) as e:
        sp.commit(240, 135, 0.5, 4)  # ... and leaves nothing acquired
    assert e.value.code == _lib.E_INVALID_ARG
    sp.submit(frames[1][1], 0.5, 91)
    assert sp.collect()[0] == 91

    def failing_fill(slot, img):
        raise OSError("decoder died")

    with pytest.raises(OSError):
        list(sp.run_zero_copy(frames[:1], 0.5, fill=failing_fill))
    sp.submit(frames[2][1], 0.5, 92)  # run_zero_copy abandoned the slot on its way out
    assert sp.collect()[0] == 92
    sp.close()


def test_batch_with_pinned_caller_buffers(ctx, model):
    """frames and masks in memory from infur_host_alloc travel by DMA, without the pageable <-> pinned staging copies: same masks"""
    imgs = [W.synth_frame(96 + 8 * (i % 3), 128, index=i) for i in range(7)]
    fp = FramePath(ctx)
    ref = fp.advance_batch(imgs, 0.5)
    pin_in = [PinnedArray(im.shape) for im in imgs]
    pin_out = [PinnedArray(r.shape) for r in ref]
    for p, im in zip(pin_in, imgs):
        np.copyto(p.array, im)
    assert ctx.L.infur_host_is_pinned(pin_in[0].array.ctypes.data) == 1 and ctx.L.infur_host_is_pinned(imgs[0].ctypes.data) == 0
    for o in pin_out:
        o.array[...] = 7
    got = fp.advance_batch([p.array for p in pin_in], 0.5, outs=[o.array for o in pin_out])
    for g, o, r in zip(got, pin_out, ref):
        assert g is o.array and (g == r).all()
    # mixed: pinned frames, pageable masks and the other way round
    got = fp.advance_batch([p.array for p in pin_in], 0.5)
    assert all((g == r).all() for g, r in zip(got, ref))
    for o in pin_out:
        o.array[...] = 9
    got = fp.advance_batch(imgs, 0.5, outs=[o.array for o in pin_out])
    assert all((g == r).all() for g, r in zip(got, ref))
    for p in pin_in + pin_out:
        p.close()


def test_streaming_full_size_throughput(ctx, model):
    """configs[2]: 1080p frames, scale 0.5, streamed from host memory; report frames/s incl. PCIe."""
    import time

    n = 24
    frames = [(i, W.synth_frame(1080, 1920, index=i % 4)) for i in range(n)]
    sp = StreamPath(ctx, depth=3)
    list(sp.run(frames[:4], 0.5))  # warm-up: allocations
    t0 = time.perf_counter()
    out = list(sp.run(frames, 0.5))
    el = time.perf_counter() - t0
    assert len(out) == n and out[0][1].shape == (540, 960, 4)
    print(f"streamed 1080p->960x540: {n / el:.1f} frames/s from host buffers (PCIe inclusive)")
    assert n / el > 30.0  # the 30 fps stream of configs[2] is sustained
    sp.close()


def test_batch_advance(ctx, model):
    """BASELINE configs[3] on one rank: a batch of independent frames (mixed sizes) == frame by frame."""
    imgs = [W.synth_frame(96 + 8 * (i % 3), 128, index=i) for i in range(8)]
    fp = FramePath(ctx)
    masks = fp.advance_batch(imgs, 0.5)
    assert len(masks) == 8
    for im, m in zip(imgs, masks):
        ref, _ = fp.advance(im, 0.5)
        assert m.shape == ref.shape and (m == ref).all()
    assert fp.advance_batch([], 1.0) == []


def test_stream_cli_raw_bgr24_in_rgba_out(ctx, model, tmp_path):
    """the CLI reads the ffmpeg image2pipe/bgr24 wire format and writes masks in order."""
    import subprocess
    import sys

    frames = [W.synth_frame(96, 160, index=i) for i in range(5)]
    fin, fout = tmp_path / "clip.bgr", tmp_path / "masks.rgba"
    fin.write_bytes(b"".join(f.tobytes() for f in frames))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "infur_amd.stream_cli", "--width", "160", "--height", "96", "--scale", "0.5",
                        "--synthetic-weights", "--input", str(fin), "--output", str(fout)], cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "5 frames" in r.stderr
    got = np.frombuffer(fout.read_bytes(), np.uint8).reshape(5, 48, 80, 4)
    fp = FramePath(ctx)
    for g, f in zip(got, frames):
        ref, _ = fp.advance(f, 0.5)
        assert (g == ref).all()
    # the default reads the pipe straight into the ring's pinned slots; --copy is the copying submit / collect path
    fout2 = tmp_path / "masks_copy.rgba"
    r = subprocess.run([sys.executable, "-m", "infur_amd.stream_cli", "--width", "160", "--height", "96", "--scale", "0.5", "--copy",
                        "--synthetic-weights", "--input", str(fin), "--output", str(fout2)], cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert fout2.read_bytes() == fout.read_bytes()
