"""GPU tests of the round-2 host layer: one-output models, failed reloads, stream lifetime, warm-up, and the
multi-GPU group (RCCL weight broadcast + frame-batch sharding) exercised on ONE device -- two contexts on
device 0, directly and through a one-rank RCCL communicator -- plus bench.py's self-launched N = 2 run."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from infur_amd import _lib
from infur_amd import weights as W
from infur_amd.app import StreamPath
from infur_amd.processors import Context, FramePath, Group, InfurError, Model, ModelCmd, ModelCmdError, ModelProcError

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- one-output models
def test_compute_aux_off_has_one_output(blob50, oracle_model):
    """Context(compute_aux=False): ModelInfo lists one output, Model.advance returns a 1-element list -- what the
    reference's `Vec<ArrayD<f32>>` would hold for a one-output model (predict_onnx.rs:326-330) -- and `out` is
    unchanged by skipping the aux head."""
    fr = W.synth_frame(64, 96, index=3)
    with Context(device=0, compute_aux=False) as c1, Context(device=0) as c2:
        m1, m2 = Model(c1).control(ModelCmd.LoadBlob(blob50)), Model(c2).control(ModelCmd.LoadBlob(blob50))
        assert m1.get_info().output_names == ["out"] and m2.get_info().output_names == ["out", "aux"]
        o1, o2 = [], []
        m1.advance(fr, o1)
        m2.advance(fr, o2)
        assert len(o1) == 1 and len(o2) == 2 and o1[0].shape == (21, 64, 96)
        assert (o1[0].view(np.uint32) == o2[0].view(np.uint32)).all()
        lo, la = m1.lowres()
        assert la is None and lo.shape[0] == 21
        # asking the C ABI for aux on a one-output context fails up front
        aux = np.empty((21, 64, 96), np.float32)
        n = C.c_uint32(7)
        rc = c1.L.infur_model_advance(c1.h, fr.ctypes.data, 96, 64, None, aux.ctypes.data, C.byref(n))
        assert rc == _lib.E_INVALID_ARG and "one output" in c1.last_error()


def test_model_without_aux_head():
    """An INFURW01 blob / ONNX file without the aux head (55 convs): one output, logits match the oracle."""
    from oracle.infur_oracle import COracle

    oracle = COracle()  # its own instance: the session-wide oracle keeps the two-headed model loaded
    blob = W.synth_blob(aux=False)
    fr = W.synth_frame(48, 64, index=1)
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(blob))
        info = m.get_info()
        assert info.output_names == ["out"]
        out = []
        m.advance(fr, out)
        assert len(out) == 1
        lo, la = m.lowres()
        assert la is None
        assert oracle.model_load(blob) == 0
        ref = oracle.model_forward(oracle.pack_normalize(fr), full=False)
        err = np.abs(lo - ref["out_low"]).max() / np.abs(ref["out_low"]).max()
        assert err < 1e-3, err
        rgba, _ = FramePath(c).advance(fr)
        assert rgba.shape == (48, 64, 4)


# ---------------------------------------------------------------- reload / lifetime / warm-up
def test_failed_reload_keeps_the_loaded_model(blob50, tmp_path):
    """Model::control leaves the old session in place on any load error (predict_onnx.rs:288-309)."""
    fr = W.synth_frame(48, 64)
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(blob50))
        ref, _ = FramePath(c).advance(fr)
        bad = bytearray(blob50)
        bad[32 + 56:32 + 64] = (2 ** 64 - 4).to_bytes(8, "little")  # conv 0 weight offset that wraps in off + n
        with pytest.raises(ModelCmdError) as e:
            m.control(ModelCmd.LoadBlob(bytes(bad)))
        assert e.value.code == _lib.E_MODEL_FORMAT and "out of range" in str(e.value)
        trunc = tmp_path / "trunc.infurw"
        trunc.write_bytes(blob50[: len(blob50) // 2])
        with pytest.raises(ModelCmdError):
            m.control(ModelCmd.Load(str(trunc)))
        with pytest.raises(ModelCmdError):
            m.control(ModelCmd.Load("/nonexistent/model.onnx"))
        assert m.get_info() is not None
        again, _ = FramePath(c).advance(fr)
        assert (again == ref).all()
        m.control(ModelCmd.Load(""))  # the explicit unload still unloads
        assert m.get_info() is None


def test_stream_outlives_context_safely(blob50):
    """closing the context before its StreamPath must not touch freed memory."""
    c = Context(device=0)
    Model(c).control(ModelCmd.LoadBlob(blob50))
    sp = StreamPath(c, depth=2)
    sp.submit(W.synth_frame(48, 64), 1.0, 1)
    c.close()  # releases the ring; the handle becomes an empty shell
    assert c.L.infur_stream_pending(sp.h) == 0
    rc = c.L.infur_stream_submit(sp.h, W.synth_frame(48, 64).ctypes.data, 64, 48, 1.0, 0, 2)
    assert rc == _lib.E_INVALID_ARG
    sp.close()  # frees only the handle
    # the usual order keeps working
    with Context(device=0) as c2:
        Model(c2).control(ModelCmd.LoadBlob(blob50))
        sp2 = StreamPath(c2, depth=2)
        sp2.submit(W.synth_frame(48, 64), 1.0, 5)
        assert sp2.collect()[0] == 5
        sp2.close()


def test_warmup_pretunes_and_size_changes_trim_the_arena(blob50):
    """infur_model_warmup: the first real frame at that size triggers no trial launches (tuning text unchanged);
    after a run of small frames the big frame's arena has been returned and results are unchanged."""
    big, small = W.synth_frame(128, 192, index=2), W.synth_frame(48, 64, index=2)
    with Context(device=0) as c:
        m = Model(c).control(ModelCmd.LoadBlob(blob50))
        with pytest.raises(InfurError):
            m.warmup(0, 10)
        m.warmup(192, 128)
        tuned = c.tuning_text()
        fp = FramePath(c)
        ref_big, _ = fp.advance(big)
        assert c.tuning_text() == tuned
        ref_small = [fp.advance(small)[0] for _ in range(8)]  # > kPoolTrimAfter frames at the small size
        assert all((r == ref_small[0]).all() for r in ref_small)
        again, _ = fp.advance(big)
        assert (again == ref_big).all()
    with Context(device=0) as c:
        with pytest.raises(InfurError) as e:
            Model(c).warmup(64, 48)
        assert e.value.code == _lib.E_MODEL_NOT_LOADED


# ---------------------------------------------------------------- the group: several contexts, one process
@pytest.mark.parametrize("force_rccl", [False, True])
def test_group_broadcast_and_sharded_batch(blob50, force_rccl, monkeypatch):
    """BASELINE configs[3] in one process: weights loaded on context 0 only, replicated by
    infur_group_weights_broadcast, 7 frames of mixed sizes split 3/2/2 over three contexts on device 0; masks equal
    the single-context results, in frame order.  force_rccl routes the replication through ncclBroadcast on a
    one-rank communicator (ncclCommInitAll) so the RCCL code path executes on a 1-GPU box."""
    if force_rccl:
        monkeypatch.setenv("INFUR_FORCE_RCCL", "1")
    else:
        monkeypatch.delenv("INFUR_FORCE_RCCL", raising=False)
    imgs = [W.synth_frame(64 + 16 * (i % 2), 96, index=i) for i in range(7)]
    ctxs = [Context(device=0) for _ in range(3)]
    try:
        Model(ctxs[0]).control(ModelCmd.LoadBlob(blob50))
        ref = FramePath(ctxs[0]).advance_batch(imgs, 0.5)
        with Group(ctxs) as g:
            assert len(g) == 3 and g.uses_rccl == force_rccl
            with pytest.raises(InfurError) as e:
                g.weights_broadcast(root=2)
            assert e.value.code == _lib.E_MODEL_NOT_LOADED
            g.weights_broadcast(root=0)
            for c in ctxs[1:]:
                info = Model(c).get_info()
                assert info is not None and info.output_names == ["out", "aux"] and info.depth == 50
            got = g.advance_batch(imgs, 0.5)
            assert len(got) == 7
            for a, b in zip(got, ref):
                assert a.shape == b.shape and (a == b).all()
            # every context really holds a working replica
            for c in ctxs[1:]:
                solo, _ = FramePath(c).advance(imgs[0], 0.5)
                assert (solo == ref[0]).all()
            assert g.advance_batch([], 1.0) == []
            assert len(g.advance_batch(imgs[:2], 0.5)) == 2  # fewer frames than contexts
            # a failing frame reports the owning context and its message (frame 5 belongs to context 2)
            outs = [np.empty_like(r) for r in ref]
            n = len(imgs)
            fp = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
            op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
            ws = (C.c_uint32 * n)(*[i.shape[1] for i in imgs])
            hs = (C.c_uint32 * n)(*[i.shape[0] for i in imgs])
            caps = (C.c_size_t * n)(*[o.nbytes if k != 5 else 4 for k, o in enumerate(outs)])
            rc = g.L.infur_group_batch_advance(g.g, fp, ws, hs, n, 0.5, 0, op, caps, None, None)
            msg = g.L.infur_group_last_error(g.g).decode()
            assert rc == _lib.E_CAPACITY and "context 2" in msg and "frames 5..6" in msg, msg
            for k in range(5):  # the other slices completed
                assert (outs[k] == ref[k]).all()
    finally:
        for c in ctxs:
            c.close()


def test_group_rejects_mixed_dtypes_and_one_shot_forms(blob50):
    a, b = Context(device=0), Context(device=0, dtype="f32s")
    try:
        Model(a).control(ModelCmd.LoadBlob(blob50))
        with Group([a, b]) as g:
            with pytest.raises(InfurError) as e:
                g.weights_broadcast(0)
            assert e.value.code == _lib.E_INVALID_ARG
    finally:
        b.close()
    b = Context(device=0)
    try:
        arr = (C.c_void_p * 2)(a.h, b.h)
        assert a.L.infur_weights_broadcast(arr, 2) == _lib.OK
        imgs = [W.synth_frame(48, 64, index=i) for i in range(3)]
        outs = [np.empty((48, 64, 4), np.uint8) for _ in imgs]
        fp = (C.c_void_p * 3)(*[i.ctypes.data for i in imgs])
        op = (C.c_void_p * 3)(*[o.ctypes.data for o in outs])
        ws, hs = (C.c_uint32 * 3)(64, 64, 64), (C.c_uint32 * 3)(48, 48, 48)
        caps = (C.c_size_t * 3)(*[o.nbytes for o in outs])
        assert a.L.infur_batch_advance_multi(arr, 2, fp, ws, hs, 3, 1.0, 0, op, caps, None, None) == _lib.OK
        for im, o in zip(imgs, outs):
            assert (o == FramePath(a).advance(im)[0]).all()
        dup = (C.c_void_p * 2)(a.h, a.h)
        g = C.c_void_p(None)
        assert a.L.infur_group_create(dup, 2, C.byref(g)) == _lib.E_INVALID_ARG
    finally:
        a.close()
        b.close()


def test_stream_with_two_lanes_equals_direct(blob50):
    """infur_stream_add_lane: consecutive frames run on alternating contexts of one device; masks and their order are
    those of the single-context path, and the lane's context may be destroyed before the stream."""
    frames = [(i + 10, W.synth_frame(96 + 8 * (i % 2), 160, index=i)) for i in range(9)]
    a, b = Context(device=0), Context(device=0)
    Model(a).control(ModelCmd.LoadBlob(blob50))
    with Group([a, b]) as g:
        g.weights_broadcast(0)
    sp = StreamPath(a, depth=3)
    sp.add_lane(b)
    with pytest.raises(InfurError):
        sp.add_lane(b)  # twice
    got = list(sp.run(frames, 0.5))
    assert [g_[0] for g_ in got] == [f[0] for f in frames]
    fp = FramePath(a)
    for (fid, rgba), (_, img) in zip(got, frames):
        ref, _ = fp.advance(img, 0.5)
        assert (rgba == ref).all(), fid
    b.close()  # a lane goes first: the stream becomes an empty handle
    assert a.L.infur_stream_pending(sp.h) == 0
    assert a.L.infur_stream_submit(sp.h, frames[0][1].ctypes.data, 160, 96, 0.5, 0, 1) == _lib.E_INVALID_ARG
    sp.close()
    a.close()


# ---------------------------------------------------------------- bench.py launches its own ranks
def test_bench_self_launches_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` (exactly what the driver runs) must start its own two ranks.  On this 1-GPU box
    both ranks share device 0 and the collective runs over gloo; the JSON line reports n_gpus == 2 and both ranks
    produce identical masks for the same frame (mask_sha of frame 0 is all-gathered and compared in bench.py)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--frames-per-step", "2", "--width", "320", "--height", "240", "--no-cpu-baseline",
                        "--no-split", "--no-side"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["weights_bcast_ms"] >= 0 and j["config"]["ranks_agree_on_frame0_mask"] is True


def test_bench_rccl_branch_executes_on_one_rank(tmp_path):
    """The `nccl` (= RCCL) branch of bench.py's N > 1 path on the 1-GPU box: RCCL refuses two ranks on one device, so
    the launcher starts ONE rank and INFUR_BENCH_FORCE_COLLECTIVES=1 makes bench.py initialise the nccl process group
    and run every collective of the multi-rank path (blob broadcast from a device buffer, barrier, MAX all-reduce of
    the elapsed time, all-gather of the mask hashes) over that one-rank communicator."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env["INFUR_BENCH_FORCE_COLLECTIVES"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend",
                        "nccl", "--steps", "2", "--warmup", "1", "--frames-per-step", "2", "--width", "320", "--height",
                        "240", "--no-cpu-baseline", "--no-split", "--no-side"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["backend"] == "nccl"
    assert j["config"]["weights_bcast_ms"] > 0 and j["config"]["ranks_agree_on_frame0_mask"] is True


def test_configs3_batch64_through_the_group_on_one_gpu(blob50):
    """BASELINE configs[3] at full size in one process: 64 distinct 1080p frames, eight contexts (here all on device 0;
    one per GPU on an 8-GPU node), weights loaded once and replicated by infur_group_weights_broadcast, the batch split
    into eight contiguous slices by infur_group_batch_advance.  Every mask comes back, in frame order, and equals what a
    single context computes for that frame (checked on one frame of each slice)."""
    n_ctx, n = 8, 64
    frames = [W.synth_frame(1080, 1920, index=500 + i) for i in range(n)]
    ctxs = [Context(device=0) for _ in range(n_ctx)]
    try:
        Model(ctxs[0]).control(ModelCmd.LoadBlob(blob50))
        with Group(ctxs) as g:
            g.weights_broadcast(0)
            masks = g.advance_batch(frames, 1.0)
        assert len(masks) == n and all(m.shape == (1080, 1920, 4) for m in masks)
        fp = FramePath(ctxs[0])
        for s in range(n_ctx):
            k = s * (n // n_ctx) + (s % (n // n_ctx))
            solo, _ = fp.advance(frames[k], 1.0)
            assert (solo == masks[k]).all(), k
        assert len({m.tobytes()[:4096] for m in masks}) > 1  # distinct frames, distinct masks
    finally:
        for c in ctxs:
            c.close()


# ---------------------------------------------------------------- round 3: the multi-GPU host path, hardened without hardware
def test_rccl_is_resolved_lazily_and_its_absence_is_an_error_of_the_group_only(tmp_path):
    """libinfur_hip.so does not link librccl: with RCCL unobtainable (INFUR_RCCL_LIB names nothing) the
    single-context Processor path works, a one-device group works (device-to-device copies), and only a group that
    needs a communicator fails -- with INFUR_E_RCCL and the loader's message, at infur_group_create."""
    script = r"""
import sys
sys.path.insert(0, sys.argv[1])
from infur_amd import _lib, weights as W
from infur_amd.processors import Context, FramePath, Group, InfurError, Model, ModelCmd
import os
blob = W.synth_blob()
a, b = Context(device=0), Context(device=0)
Model(a).control(ModelCmd.LoadBlob(blob))
fr = W.synth_frame(48, 64)
ref, _ = FramePath(a).advance(fr, 1.0)                  # single-GPU path: no RCCL anywhere
with Group([a, b]) as g:                               # one device: no communicator needed
    assert not g.uses_rccl
    g.weights_broadcast(0)
assert (FramePath(b).advance(fr, 1.0)[0] == ref).all()
os.environ["INFUR_FORCE_RCCL"] = "1"
try:
    Group([a, b])
    raise SystemExit("a group that needs RCCL was created without it")
except InfurError as e:
    assert e.code == _lib.E_RCCL and "not available" in str(e), str(e)
print("ok")
"""
    env = dict(os.environ)
    env["INFUR_RCCL_LIB"] = str(tmp_path / "no_such_librccl.so")
    env.pop("INFUR_FORCE_RCCL", None)
    r = subprocess.run([sys.executable, "-c", script, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


def test_injected_rccl_failure_leaves_every_context_as_it_was(blob50, monkeypatch):
    """INFUR_E_RCCL from the collective (injected: INFUR_RCCL_INJECT_FAIL=broadcast; a healthy box never produces one):
    the call reports it with RCCL's message, the receive arenas are released, the receiving context keeps what it had (here:
    nothing, then a model of its own) and the same group succeeds once the fault is gone."""
    monkeypatch.setenv("INFUR_FORCE_RCCL", "1")
    a, b = Context(device=0), Context(device=0)
    try:
        Model(a).control(ModelCmd.LoadBlob(blob50))
        fr = W.synth_frame(48, 64, index=2)
        ref, _ = FramePath(a).advance(fr, 1.0)
        with Group([a, b]) as g:
            assert g.uses_rccl
            monkeypatch.setenv("INFUR_RCCL_INJECT_FAIL", "broadcast")
            with pytest.raises(InfurError) as e:
                g.weights_broadcast(0)
            assert e.value.code == _lib.E_RCCL and "ncclBroadcast" in str(e.value)
            assert Model(b).get_info() is None  # nothing half-adopted
            assert (FramePath(a).advance(fr, 1.0)[0] == ref).all()  # the root is untouched
            mb = Model(b).control(ModelCmd.LoadBlob(W.synth_blob(aux=False)))  # b gets a model of its own ...
            with pytest.raises(InfurError):
                g.weights_broadcast(0)
            assert mb.get_info().output_names == ["out"]  # ... and keeps it through a second failed broadcast
            monkeypatch.delenv("INFUR_RCCL_INJECT_FAIL")
            g.weights_broadcast(0)
            assert Model(b).get_info().output_names == ["out", "aux"]
            assert (FramePath(b).advance(fr, 1.0)[0] == ref).all()
        monkeypatch.setenv("INFUR_RCCL_INJECT_FAIL", "init")
        with pytest.raises(InfurError) as e:
            Group([a, b])
        assert e.value.code == _lib.E_RCCL and "ncclCommInitAll" in str(e.value)
    finally:
        a.close()
        b.close()


def test_group_workers_are_pinned_to_their_gpus_numa_node(blob50, monkeypatch):
    """every worker reports the NUMA node it is pinned to: the node sysfs gives for the GPU's PCI address, or -1 where the
    host exposes none (containers); INFUR_NO_NUMA_PIN=1 switches the pinning off; results do not depend on it"""
    import torch

    a, b = Context(device=0), Context(device=0)
    try:
        Model(a).control(ModelCmd.LoadBlob(blob50))
        imgs = [W.synth_frame(48, 64, index=i) for i in range(4)]
        with Group([a, b]) as g:
            nodes = g.worker_numa_nodes()
            g.weights_broadcast(0)
            got = g.advance_batch(imgs, 1.0)
        pr = torch.cuda.get_device_properties(0)
        path = f"/sys/bus/pci/devices/{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0/numa_node"
        want = int(open(path).read()) if os.path.exists(path) else -1
        print("worker NUMA nodes", nodes, "sysfs says", want)
        assert len(nodes) == 2 and nodes[0] == nodes[1] and nodes[0] >= -1
        if want >= 0 and os.path.exists(f"/sys/devices/system/node/node{want}/cpulist"):
            assert nodes[0] == want
        monkeypatch.setenv("INFUR_NO_NUMA_PIN", "1")
        with Group([a, b]) as g:
            assert g.worker_numa_nodes() == [-1, -1]
            again = g.advance_batch(imgs, 1.0)
        for x, y in zip(got, again):
            assert (x == y).all()
    finally:
        a.close()
        b.close()


def test_batch_ring_is_persistent_and_follows_the_frame_size(blob50):
    """infur_batch_advance keeps its depth-3 ring on the context: a second batch allocates nothing new (device memory
    unchanged), a batch of much smaller frames shrinks it, a failing batch drops it and the next one works again"""
    import torch

    with Context(device=0) as c:
        Model(c).control(ModelCmd.LoadBlob(blob50))
        fp = FramePath(c)
        big = [W.synth_frame(270, 480, index=i) for i in range(5)]
        small = [W.synth_frame(32, 48, index=i) for i in range(5)]
        ref_big = [fp.advance(f, 1.0)[0] for f in big]
        ref_small = [fp.advance(f, 1.0)[0] for f in small]
        assert all((x == y).all() for x, y in zip(fp.advance_batch(big, 1.0), ref_big))
        free0 = torch.cuda.mem_get_info(0)[0]
        for _ in range(3):
            assert all((x == y).all() for x, y in zip(fp.advance_batch(big, 1.0), ref_big))
        assert torch.cuda.mem_get_info(0)[0] == free0, "a repeated batch of the same size allocated device memory"
        # a batch that ALTERNATES large and small frames must not free / reallocate the ring's buffers frame by frame
        mixed = [big[0], small[0], big[1], small[1], big[2], small[2]]
        ref_mixed = [ref_big[0], ref_small[0], ref_big[1], ref_small[1], ref_big[2], ref_small[2]]
        assert all((x == y).all() for x, y in zip(fp.advance_batch(mixed, 1.0), ref_mixed))
        free1 = torch.cuda.mem_get_info(0)[0]
        for _ in range(3):
            assert all((x == y).all() for x, y in zip(fp.advance_batch(mixed, 1.0), ref_mixed))
            assert torch.cuda.mem_get_info(0)[0] == free1, "alternating frame sizes made the ring reallocate"
        for _ in range(6):  # (the activation arena trims itself after a few frames of the new size as well)
            assert all((x == y).all() for x, y in zip(fp.advance_batch(small, 1.0), ref_small))
        assert torch.cuda.mem_get_info(0)[0] > free0, "the ring kept its 480x270 buffers for 48x32 frames"
        # a failing batch (mask buffer of frame 1 too small) leaves the context usable
        outs = [np.empty_like(r) for r in ref_small]
        n = len(small)
        fr = (C.c_void_p * n)(*[i.ctypes.data for i in small])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ws = (C.c_uint32 * n)(*[i.shape[1] for i in small])
        hs = (C.c_uint32 * n)(*[i.shape[0] for i in small])
        caps = (C.c_size_t * n)(*[o.nbytes if k != 1 else 4 for k, o in enumerate(outs)])
        assert c.L.infur_batch_advance(c.h, fr, ws, hs, n, 1.0, 0, op, caps, None, None) == _lib.E_CAPACITY
        assert all((x == y).all() for x, y in zip(fp.advance_batch(small, 1.0), ref_small))


def test_stream_lane_must_match_the_streams_arithmetic(blob50):
    """a lane with another Winograd tile (or no model yet) would make odd and even frames differ"""
    a, b, d = Context(device=0), Context(device=0, winograd_tile=4), Context(device=0)
    try:
        Model(a).control(ModelCmd.LoadBlob(blob50))
        Model(b).control(ModelCmd.LoadBlob(blob50))
        sp = StreamPath(a, depth=3)
        with pytest.raises(InfurError) as e:
            sp.add_lane(b)
        assert e.value.code == _lib.E_INVALID_ARG and "winograd_tile" in str(e.value)
        with pytest.raises(InfurError) as e:
            sp.add_lane(d)  # same options, but nothing loaded
        assert e.value.code == _lib.E_MODEL_NOT_LOADED
        sp.close()
    finally:
        for c in (a, b, d):
            c.close()


def test_bench_self_launches_eight_ranks_on_one_gpu(tmp_path):
    """the driver's N = 8 command, `python bench.py --gpus 8`, on the 1-GPU box: eight ranks share device 0 over gloo
    (tiny frames), the line reports n_gpus == 8 and all ranks agree on frame 0's mask"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "1",
                        "--warmup", "1", "--frames-per-step", "1", "--contexts-per-gpu", "1", "--width", "160", "--height", "120",
                        "--no-cpu-baseline", "--no-split", "--no-side", "--no-profile"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 8 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["ranks_agree_on_frame0_mask"] is True and j["config"]["oversubscribed"] is True


POOL_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1])
from infur_amd.processors import Context

ctxs = [Context(device=0) for _ in range(16)]
streams = [c.stream for c in ctxs]
assert all(s != 0 for s in streams)
assert all(streams[i] != streams[i + 1] for i in range(15))
assert len(set(streams)) == 8 and streams[:8] == streams[8:], streams
for c in ctxs:
    c.close()
again = [Context(device=0) for _ in range(8)]
assert set(c.stream for c in again) == set(streams)  # the same eight streams: closing a context destroys nothing
import torch

st = torch.cuda.Stream()
with Context(device=0, stream=st.cuda_stream) as c:
    assert c.stream == st.cuda_stream and c.stream not in streams  # a host's own stream is used as it is
for c in again:
    c.close()
print("pool ok")
"""


def test_library_streams_come_from_a_pool_of_eight():
    """Contexts created without a stream of their own get pool streams (infur_ctx_create: eight per device, created in one go, handed
    out round-robin, never destroyed) -- consecutive contexts must get different streams (two frames in flight only overlap on
    different hardware queues), the pool must not grow, and a context's stream must survive the context it was lent to.  In a process
    of its own (round 6): the hand-out order depends on which slots are held when the test starts -- the session's `ctx` fixture holds
    one, and where the round-robin cursor stands relative to it depends on how many contexts earlier tests created."""
    r = subprocess.run([sys.executable, "-c", POOL_SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "pool ok" in r.stdout, r.stderr[-2000:]
