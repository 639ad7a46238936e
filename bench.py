#!/usr/bin/env python3
"""bench.py -- 1080p frames/s of the FCN-ResNet50 segmentation path on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (packed BGR frame -> [scale] -> fused pre-proc + stem ->
FCN-ResNet50 incl. aux head -> bilinear up-sample + argmax + shade -> RGBA mask) over one batch
of ``--frames-per-step`` distinct synthetic frames per GPU, inputs and outputs resident in HBM.
N > 1 runs one rank per GPU under torch.distributed.run -- started by the driver, or by this script
itself when it finds no launcher around it (`python bench.py --gpus N`): the weight blob is
broadcast once over RCCL (timed separately, outside the region), frames are sharded with no
data-path collective ("weak" scaling: per-GPU work is fixed).  Rank 0 prints ONE JSON line.

The CPU oracle (oracle/) is used here only for the ``cpu_baseline`` leg and a one-frame
parity spot check; it is never part of the measured path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
CONV_KERNELS = ("conv_igemm_", "conv1x1_", "conv3x3_", "conv_hl")  # tiled implicit GEMM; A-resident 1x1 and the fused conv3 -> next conv1 pair (f16 mode)
COPY_CEILING_GBS = 5450.0  # measured: a streaming copy of 0.5-4 GB sustains 5.3-5.6 TB/s read + write on this pool (profiles/r03_copy_ceiling.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--scale-mode", type=int, default=0, help="0 nearest (reference), 1 bilinear")
    ap.add_argument("--frames-per-step", type=int, default=8)
    ap.add_argument("--contexts-per-gpu", type=int, default=3,
                    help="contexts (HIP streams + arenas) per GPU working on different frames of the batch at the same time: the "
                         "tail of one frame's kernel overlaps the next frame's (f32: 2: +3 %% frames/s, 3: +4.5 %%; 1: strictly one frame at a time)")
    ap.add_argument("--no-aux", action="store_true", help="skip the aux head (the ONNX graph always evaluates it)")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="CPU baseline budget")
    ap.add_argument("--kernels", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--no-split", action="store_true", help="skip the f32_split_mode side measurement")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements of BASELINE configs[2] (1080p stream at scale 0.5, PCIe inclusive) and "
                         "configs[4] (4K FCN-ResNet101 f16)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "f32s", "f32x", "f16hl", "i8"],
                    help="conv-stack arithmetic: f32 (BASELINE configs[1], the default and the parity mode) or f16 "
                         "operands with f32 accumulation (configs[4]'s mode)")
    ap.add_argument("--winograd-min-cin", type=int, default=0,
                    help="f32 stride-1 3x3 convs with Cin >= this run as Winograd F(2x2,3x3); 0 = library default, -1 = never")
    ap.add_argument("--winograd-tile", type=int, default=0, choices=[0, 2, 4, 6], help="Winograd output tile (0 = default: 6)")
    ap.add_argument("--depth", type=int, default=50, choices=[50, 101], help="backbone: FCN-ResNet50 (default) / 101")
    ap.add_argument("--cpu-probe", default=None, help=argparse.SUPPRESS)  # internal: child process of cpu_baseline
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise the "
                         "multi-process path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the
    driver's own multi-GPU form does (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), and hand
    its exit code back.  With fewer visible GPUs than ranks the ranks share devices and the collective runs over
    gloo (RCCL cannot put two ranks on one device); the JSON line says so (`config.oversubscribed`)."""
    import socket
    import subprocess

    import torch

    argv = list(sys.argv[1:])
    if torch.cuda.device_count() < a.gpus and a.backend == "nccl":
        sys.stderr.write(f"bench.py: {torch.cuda.device_count()} GPU(s) visible for {a.gpus} ranks: ranks share devices, "
                         "collective over gloo\n")
        argv += ["--backend", "gloo"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def cpu_model_string():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(blob, frame, budget_s):
    """The oracle's whole path on the host cores: pre-proc (C) + FCN-ResNet50 (torch-CPU / oneDNN) + up-sample +
    ColorCode (C).  Bounded sample of the bench workload: per thread count one un-timed warm-up frame (oneDNN
    creates its primitives on first use) and then two timed whole frames, at the reference's own setting (ONNX
    Runtime pinned to 3 intra-op threads, predict_onnx.rs:292), at 16 and at 32 threads.  SURVEY 8d also asks for
    ALL host cores: oneDNN collapses far above its best thread count, so that sample is bounded by first timing a
    1/64-area probe frame and running the whole frame only if the scaled estimate fits the remaining budget;
    otherwise the probe-scaled estimate is reported and labelled as such.  `value` is the best measured rate."""
    import numpy as np
    import torch

    from oracle.infur_oracle import COracle, TorchModel

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    co = COracle(threads=min(cores, 16))
    tm = TorchModel(blob)
    h, w = frame.shape[:2]

    def one_frame(fr):
        fh, fw = fr.shape[:2]
        t0 = time.perf_counter()
        chw = co.pack_normalize(fr)
        lo, _ = tm.forward_lowres(chw)
        full = co.upsample_bilinear(lo.numpy(), fh, fw)
        co.colorcode(full)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    spent = lambda: time.perf_counter() - t_start  # noqa: E731
    results, samples = {}, {}
    for th in (16, 3, 32):
        if th > cores or spent() > budget_s:
            continue
        torch.set_num_threads(th)
        one_frame(frame)  # warm-up, not timed
        ts = [one_frame(frame)]
        if spent() < budget_s:
            ts.append(one_frame(frame))
        results[th] = min(ts)
        samples[th] = len(ts)
    all_cores = None
    if cores not in results:
        # SURVEY 8d's all-cores sample, in a child process under a hard time limit: oneDNN at 256 threads on this
        # class of host needs tens of seconds for a 240x135 frame, so the sample must not be able to stall the bench
        import subprocess

        ph, pw = max(h // 8, 32), max(w // 8, 32)
        limit = 20.0
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", f"{cores},{pw},{ph}"],
                               capture_output=True, text=True, timeout=limit)
            tp = float(r.stdout.strip().splitlines()[-1])
            est = tp * (h * w) / float(ph * pw)
            all_cores = {"threads": cores, "frames_per_s": 1.0 / est,
                         "kind": f"extrapolated by pixel count from one {pw}x{ph} probe frame ({tp * 1e3:.0f} ms at {cores} threads, "
                                 "after one warm-up frame)"}
        except subprocess.TimeoutExpired:
            all_cores = {"threads": cores, "frames_per_s": None,
                         "kind": f"a {pw}x{ph} probe frame (1/64 of the pixels) + its warm-up did not finish within {limit:.0f} s at "
                                 f"{cores} threads: < {1.0 / (limit / 2 * 64):.5f} frames/s 1080p-equivalent; oneDNN collapses far "
                                 "above its best thread count on this host"}
        except (ValueError, IndexError, OSError):
            all_cores = {"threads": cores, "frames_per_s": None, "kind": "probe failed"}
    # the plain C / OpenMP restatement (oracle/infur_oracle.c) at the reference's 3 threads, on a 1/64-area
    # probe frame (a whole 1080p frame would take minutes at its ~6 GFLOP/s), scaled to 1080p by conv FLOPs
    b1 = None
    try:
        from infur_amd import weights as W

        ph, pw = max(h // 8, 32), max(w // 8, 32)
        co3 = COracle(threads=3)
        if co3.model_load(blob) == 0:
            pf = np.ascontiguousarray(frame[:ph, :pw])
            t0 = time.perf_counter()
            co3.model_forward(co3.pack_normalize(pf), full=False)
            tp = time.perf_counter() - t0
            ratio = W.conv_flops(h, w)["total"] / W.conv_flops(ph, pw)["total"]
            b1 = {"threads": 3, "frames_per_s": 1.0 / (tp * ratio), "probe_seconds": tp,
                  "kind": f"B1: C/OpenMP restatement of the whole forward on a {pw}x{ph} probe frame, scaled to {w}x{h} by conv FLOPs (x{ratio:.1f})"}
    except Exception as e:  # noqa: BLE001
        b1 = {"error": f"{type(e).__name__}: {e}"}
    best = min(results, key=results.get)
    out = {
        "value": 1.0 / results[best], "unit": "frames/s", "cores": best, "kind": "port",
        "sample": f"whole {w}x{h} frames through the oracle path (C pre-proc, torch-CPU oneDNN FCN-ResNet50 incl. aux head, C "
                  f"up-sample + ColorCode): per thread count 1 warm-up + best of {dict(sorted(samples.items()))} timed frames; "
                  f"{spent():.1f} s of CPU wall time",
        "host_cpu": cpu_model_string(), "host_logical_cores": cores, "os_cpu_count": os.cpu_count(),
        "frames_per_s_by_threads": {str(k): 1.0 / v for k, v in sorted(results.items())},
        "all_cores": all_cores,
        "c_oracle_3_threads": b1,
        "which": "value / cores = B2 (torch-CPU oneDNN port) at its best measured thread count; frames_per_s_by_threads['3'] = B2 at the "
                 "reference's ORT setting; c_oracle_3_threads = B1 (plain C restatement)",
        "reference_note": "the reference pins ONNX Runtime to 3 intra-op threads (predict_onnx.rs:292); the reference itself "
                          "cannot run here (no cargo; onnxruntime / the model file are picked up when present, see "
                          "`reference_runtime`)",
    }
    ort = ort_baseline(frame)
    if ort is not None:
        out["reference_runtime"] = ort
        if "frames_per_s" in ort:
            out.update(value=ort["frames_per_s"], cores=3, kind="reference-runtime")
    return out


def ort_baseline(frame):
    """Opportunistic: when `onnxruntime` imports and INFUR_ONNX_MODEL names the zoo's
    fcn-resnet50-12.onnx, time the reference's own runtime configured as the reference configures it
    (predict_onnx.rs:289-293: 3 intra-op threads, extended graph optimisations) on the bench frame."""
    path = os.environ.get("INFUR_ONNX_MODEL")
    if not path:
        return None
    try:
        import numpy as np
        import onnxruntime as ort
    except ImportError:
        return {"skipped": "INFUR_ONNX_MODEL is set but onnxruntime does not import"}
    if not os.path.exists(path):
        return {"skipped": f"{path} does not exist"}
    so = ort.SessionOptions()
    so.intra_op_num_threads = 3
    so.graph_optimization_level = ort.GraphOptimizationLevel.ORT_ENABLE_EXTENDED
    sess = ort.InferenceSession(path, so, providers=["CPUExecutionProvider"])
    mean = np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]
    std1 = (np.float32(1.0) / np.array([0.229, 0.224, 0.225], np.float32))[:, None, None]
    x = (np.ascontiguousarray(frame[..., ::-1].transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0) - mean) * std1
    name = sess.get_inputs()[0].name
    sess.run(None, {name: x[None]})  # warm-up
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        sess.run(None, {name: x[None]})
        ts.append(time.perf_counter() - t0)
    return {"frames_per_s": 1.0 / min(ts), "threads": 3, "model": os.path.basename(path), "onnxruntime": ort.__version__,
            "note": "session.run only (the reference's pre-proc and ColorCode run outside it)"}


def cpu_probe(spec):
    """Child of cpu_baseline: `threads,w,h` -> seconds for one probe frame through the torch-CPU network (after one
    warm-up frame), printed on stdout."""
    import torch

    from infur_amd import weights as W
    from oracle.infur_oracle import COracle, TorchModel

    th, w, h = (int(x) for x in spec.split(","))
    torch.set_num_threads(th)
    co = COracle(threads=min(th, 16))
    tm = TorchModel(W.synth_blob())
    fr = W.synth_frame(h, w)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        lo, _ = tm.forward_lowres(co.pack_normalize(fr))
        co.colorcode(co.upsample_bilinear(lo.numpy(), h, w))
        ts.append(time.perf_counter() - t0)
    print(ts[-1], flush=True)


LINE_LIMIT = 6000  # bytes: the driver recovers the JSON line from an 8 KB stdout tail (round 5's 20 KB line came back `parsed: null`)
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
SIDE_KEYS = ("f32_split_mode", "f32_split_fp8_mode", "f16hl_mode_1080p", "f16_mode_1080p", "int8_quantised_model", "configs2_stream_scale05",
             "configs4_r101_f16_4k", "configs3_batch64_group", "configs3_batch64_group_f16hl")


def _sig(x, digits=5):
    """floats to `digits` significant figures, recursively (the line is a record, not a measurement instrument)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _clip(x, n=120):
    """no string of the line longer than n characters (dict keys included: a kernel name as a key must not blow the budget)."""
    if isinstance(x, str):
        return x[:n]
    if isinstance(x, dict):
        return {str(k)[:n]: _clip(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clip(v, n) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(out):
    """The ONE stdout line: headline + `config` + `roofline` + `cpu_baseline` numbers only, <= LINE_LIMIT bytes.
    Every note string, the per-kernel tables (`other_kernels`, `winograd`, `conv3x3`, tile-configuration times) and the side objects go to
    bench_detail.json next to this file (and to stderr); `config.side_rates` keeps each side object's [frames/s, roofline fraction
    (executed_frac where Winograd runs), PCIe-inclusive frames/s]."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data"))
    cfg = out.get("config") or {}
    c = _pick(cfg, ("workload", "frames_per_step_per_gpu", "contexts_per_gpu", "frames_per_s_one_context", "sharding", "backend", "weights_load_ms",
                    "weights_bcast_ms", "ranks_agree_on_frame0_mask", "oversubscribed", "ms_per_frame_per_gpu", "conv_gflop_per_frame",
                    "effective_conv_tflops", "pcie_inclusive_frames_per_s", "pcie_inclusive_zero_copy_frames_per_s"))
    side = {}
    for key in SIDE_KEYS:
        o = out.get(key)
        if isinstance(o, dict) and "value" in o:
            r = o.get("roofline") or {}
            frac = r.get("executed_frac", r.get("frac"))
            side[key] = [round(o["value"], 1), None if frac is None else round(frac, 3), o.get("pcie_inclusive_frames_per_s")]
        elif isinstance(o, dict) and "error" in o:
            side[key] = str(o["error"])[:60]
    if side:
        c["side_rates"] = side
    c["detail"] = "bench_detail.json (next to bench.py; also on stderr): notes, per-kernel tables, side objects"
    line["config"] = c
    r = out.get("roofline")
    if isinstance(r, dict):
        rr = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "flops_per_launch",
                       "algorithmic_bytes_per_launch", "frame_kernel_ms"))
        rr["kernel"] = str(r.get("kernel", "")).split(" (")[0][:80]
        if isinstance(r.get("all_convs"), dict):
            rr["all_convs"] = _pick(r["all_convs"], ("achieved", "frac", "ms"))
        w = r.get("winograd")
        if isinstance(w, dict):
            rr["winograd"] = _pick(w, ("launches", "gemm_ms", "transform_ms", "executed_frac"))
        if "executed_frac" in r:
            rr["executed_frac"] = r["executed_frac"]
        line["roofline"] = rr
    b = out.get("cpu_baseline")
    if isinstance(b, dict):
        bb = _pick(b, ("value", "unit", "cores", "kind", "host_cpu", "host_logical_cores", "frames_per_s_by_threads"))
        bb["sample"] = str(b.get("sample", ""))[:160]
        c3 = b.get("c_oracle_3_threads")
        if isinstance(c3, dict):
            bb["c_oracle_3_threads"] = _pick(c3, ("frames_per_s", "threads", "error"))
        if isinstance(b.get("reference_runtime"), dict):
            bb["reference_runtime"] = _pick(b["reference_runtime"], ("frames_per_s", "threads", "model", "onnxruntime", "skipped"))
        line["cpu_baseline"] = bb
    text = json.dumps(_clip(_sig(line)), separators=(",", ":"))
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes (limit {LINE_LIMIT}): move fields to bench_detail.json"
    return text


def emit(out):
    """bench_detail.json + stderr get the full record; stdout gets the one compact line, last."""
    detail = json.dumps(out, indent=1, default=str)
    try:
        with open(DETAIL_FILE, "w") as f:
            f.write(detail + "\n")
    except OSError as e:
        sys.stderr.write(f"bench.py: cannot write {DETAIL_FILE}: {e}\n")
    sys.stderr.write(json.dumps(out, default=str) + "\n")
    sys.stderr.flush()
    print(compact_line(out), flush=True)


def main():
    a = parse()
    if a.cpu_probe:
        return cpu_probe(a.cpu_probe)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    import hashlib

    import numpy as np
    import torch
    import torch.distributed as dist

    from infur_amd import dist as idist
    from infur_amd import weights as W
    from infur_amd.processors import Context, FramePath

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # INFUR_BENCH_FORCE_COLLECTIVES=1 under a launcher with ONE rank: the process group is initialised and every
    # collective of the N > 1 path (broadcast of the blob, barrier, max-reduce, all-gather of the mask hashes) runs
    # over a one-rank communicator -- this is how the RCCL branch executes on a 1-GPU box (tests/test_gpu_multi.py)
    multi = world > 1 or (os.environ.get("INFUR_BENCH_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ)
    ndev = torch.cuda.device_count()
    dev = local_rank if a.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)
    coll_dev = f"cuda:{dev}" if a.backend == "nccl" else "cpu"
    if multi:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev}"))
        else:
            dist.init_process_group("gloo")

    # K contexts per GPU, each with its own stream and activation arena, take the frames of a step in turn: kernels
    # of different frames overlap where one of them leaves CUs idle (tails of a launch, HBM-bound next to MFMA-bound)
    K = max(1, a.contexts_per_gpu)
    streams = [torch.cuda.Stream() for _ in range(K)]
    # --dtype i8: the QUANTISED model (INFURQ01: u8 x s8 on the i8 MFMA); a quantised model defines its own arithmetic, the
    # context's compute dtype does not matter to it
    ctxs = [Context(device=dev, compute_aux=not a.no_aux, profile=not a.no_profile, stream=st.cuda_stream, dtype="f32" if a.dtype == "i8" else a.dtype,
                    winograd_min_cin=a.winograd_min_cin & 0xFFFFFFFF, winograd_tile=a.winograd_tile) for st in streams]
    ctx = ctxs[0]

    # ---- weights: rank 0 synthesises, RCCL broadcast over xGMI, every rank repacks locally; the rank's other
    #      contexts get the repacked arena through the C ABI's group call (device-to-device on one GPU) ----
    if a.dtype == "i8":
        from infur_amd import quantize

        blob = quantize.synth_qblob(depth=a.depth) if rank == 0 else None
    else:
        blob = W.synth_blob(depth=a.depth) if rank == 0 else None
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbytes, bcast_ms = idist.load_model_everywhere(ctx, blob, coll_device=coll_dev)
    if K > 1:
        from infur_amd.processors import Group

        with Group(ctxs) as grp:
            grp.weights_broadcast(0)
    torch.cuda.synchronize()
    load_ms = (time.perf_counter() - t0) * 1e3

    # ---- synthetic frames, resident in HBM ----
    H, Wd, B = a.height, a.width, a.frames_per_step
    frames_np = [W.synth_frame(H, Wd, index=rank * B + i) for i in range(B)]
    d_frames = [torch.from_numpy(f).cuda() for f in frames_np]
    rc_w, rc_h = idims(ctx, Wd, H, a.scale)
    d_masks = [torch.empty((rc_h, rc_w, 4), dtype=torch.uint8, device="cuda") for _ in range(B)]
    fps_ = [FramePath(c, a.scale_mode) for c in ctxs]
    fp = fps_[0]
    torch.cuda.synchronize()

    def sync_all():
        for c in ctxs:
            c.synchronize()

    def step(profile_last=False):
        for i in range(B):
            k = i % K
            if profile_last and i == B - 1:
                # the roofline sample: per-kernel HIP events for this frame only, and the frame runs ALONE -- context 0's
                # stream waits (on the GPU, no host round trip that would let the chip idle and clock down) for the frames
                # the other contexts have in flight, so that an event pair brackets one kernel and nothing else
                k = 0
                for st in streams[1:]:
                    streams[0].wait_stream(st)
                ctx.L.infur_profile_enable(ctx.h, 1)
            fps_[k].advance_dev(d_frames[i].data_ptr(), Wd, H, a.scale, d_masks[i].data_ptr(), d_masks[i].numel())

    for _ in range(a.warmup):
        step()
    sync_all()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # Per-kernel HIP events bracket every launch of the LAST frame of the timed region only (the
    # roofline sample); recording them on all frames costs ~2.5 % of throughput in event packets.
    for c in ctxs:
        c.L.infur_profile_enable(c.h, 0)
    for k in range(a.steps):
        step(profile_last=(not a.no_profile) and k == a.steps - 1)
    sync_all()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the roofline records: per-kernel HIP events of the timed region's LAST frame (every later frame resets them)
    recs_timed = ctx.profile() if (rank == 0 and not a.no_profile) else None
    for c in ctxs:
        c.L.infur_profile_enable(c.h, 0)

    # every rank decodes the SAME frame once more (outside the timed region): replicas must agree bit for bit
    agree = None
    if multi:
        f0 = torch.from_numpy(W.synth_frame(H, Wd, index=0)).cuda()
        fps_[K - 1].advance_dev(f0.data_ptr(), Wd, H, a.scale, d_masks[0].data_ptr(), d_masks[0].numel())
        sync_all()
        sha = hashlib.sha256(d_masks[0].cpu().numpy().tobytes()).hexdigest()
        shas = [None] * world
        dist.all_gather_object(shas, sha)
        agree = len(set(shas)) == 1

    # for reference: the same frames with strictly one frame at a time (context 0 only), outside the timed region,
    # without per-kernel events
    one_ctx_fps = None
    if K > 1 and world == 1:
        sync_all()
        t1 = time.perf_counter()
        for _ in range(2):
            for i in range(B):
                fp.advance_dev(d_frames[i].data_ptr(), Wd, H, a.scale, d_masks[i].data_ptr(), d_masks[i].numel())
        ctx.synchronize()
        one_ctx_fps = 2 * B / (time.perf_counter() - t1)

    frames_total = world * B * a.steps
    fps = frames_total / elapsed
    out = {
        "metric": "1080p frames/sec FCN-ResNet50-12 @1/2/4/8 MI355X; %MFMA roofline",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {
            "workload": f"{Wd}x{H} packed-BGR frame, FCN-ResNet{a.depth} {a.dtype} (aux head {'off' if a.no_aux else 'on'}), "
                        f"scale={a.scale}" + (" [BASELINE configs[1]]" if (Wd, H, a.scale, a.dtype, a.depth) == (1920, 1080, 1.0, "f32", 50) else ""),
            "frames_per_step_per_gpu": B, "contexts_per_gpu": K, "frames_per_s_one_context": one_ctx_fps, "sharding": f"frames x{world}, no data-path collective", "backend": a.backend if multi else None,
            "weights": f"synthetic seed {W.DEFAULT_SEED:#x}, {nbytes / 1e6:.1f} MB blob",
            "weights_load_ms": round(load_ms, 2), "weights_bcast_ms": round(bcast_ms, 3),
            "weights_note": "weights_load_ms = broadcast + per-rank repack into kernel layouts; weights_bcast_ms = the "
                            "collective alone (0 at N = 1); both outside the timed region",
            "ranks_agree_on_frame0_mask": agree, "oversubscribed": world > max(ndev, 1),
            "ms_per_frame_per_gpu": elapsed / (a.steps * B) * 1e3,
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel family from the HIP events of the last timed frame ----
        flops = W.conv_flops(rc_h, rc_w, depth=a.depth, aux=not a.no_aux)
        if not a.no_profile:
            recs = recs_timed
            k3 = {c.name for c in W.graph(a.depth) if c.k == 3}
            # f32s: three f16 MFMAs per product -> the ceiling for f32-equivalent FLOPs is a third of the f16 peak
            peak = {"f32": PEAK_F32_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS, "f32s": PEAK_F16_MFMA_TFLOPS / 3.0,
                    "f32x": PEAK_F16_MFMA_TFLOPS / 2.0,  # f32x: one f16 MFMA + one fp8 MX MFMA (half an f16 unit per term) per product
                    "f16hl": PEAK_F16_MFMA_TFLOPS / 2.0,  # the same two units on three-byte tensors
                    "i8": 2.0 * PEAK_F16_MFMA_TFLOPS}[a.dtype]  # dense i8 MFMA: twice the f16 rate (TOP/s; >= 3944 measured in the guide)
            conv = [r for r in recs if r["kernel"].startswith(CONV_KERNELS)]
            # dominant kernel = the tile configuration of conv_igemm that takes the most time in a frame over the
            # launches that execute the convolution directly (algorithmic FLOPs == executed FLOPs); the
            # Winograd-domain GEMM launches are reported under "winograd" with both views
            is_wino = lambda r: r["algo_flops"] > r["flops"] * 1.01  # noqa: E731
            by_cfg, by_cfg_direct = {}, {}
            for r in conv:
                by_cfg[r["kernel"]] = by_cfg.get(r["kernel"], 0.0) + r["ms"]
                if not is_wino(r):
                    by_cfg_direct[r["kernel"]] = by_cfg_direct.get(r["kernel"], 0.0) + r["ms"]
            dom_name = max(by_cfg_direct, key=by_cfg_direct.get)
            dom = [r for r in conv if r["kernel"] == dom_name and not is_wino(r)]
            wg = [r for r in conv if is_wino(r)]
            c3 = [r for r in conv if r["name"] in k3]
            c1 = [r for r in conv if r["name"] not in k3]
            wino = [r for r in recs if r["kernel"].startswith("wino_")]
            tf = lambda rs: sum(r["flops"] for r in rs) / max(sum(r["ms"] for r in rs), 1e-9) / 1e9  # noqa: E731
            ms = lambda rs: sum(r["ms"] for r in rs)  # noqa: E731
            ms_all = ms(recs)
            traffic = None
            tj = os.path.join(ROOT, "profiles", "traffic_latest.json" if a.dtype == "f32" else f"traffic_{a.dtype}.json")
            if os.path.exists(tj) and (Wd, H, a.scale, a.depth) == (1920, 1080, 1.0, 50) and "<" in dom_name:
                parts = dom_name.split("<")[1].rstrip(">").split(",")  # "64,64" or "64,64,1buf" / "256,256,1frag"
                bm, bn = parts[0], parts[1]
                nbuf = {"1buf": "1", "1frag": "3", "dma": "4", "dmai": "5"}.get(parts[2], "2") if len(parts) > 2 else "2"
                waves = {"128,256": "2, 4", "256,128": "4, 2", "256,32": "4, 1", "256,256": "2, 4"}.get(f"{bm},{bn}", "2, 2")
                el = {"f16": "_Float16, _Float16", "i8": "signed char, unsigned char"}.get(a.dtype, "float, float")
                split = {"f32": "false", "f16": "false", "f32s": "true", "f32x": "true", "i8": "false", "f16hl": "false"}[a.dtype]
                # all instantiations of this tile (plain / 1x1-GEMM addressing / residual prefetch), launch-weighted
                pre = f"conv_igemm_kernel<{el}, {bm}, {bn}, {waves}, {nbuf}, {split}"
                ts = [t for k, t in json.load(open(tj))["kernels"].items() if k.startswith(pre)]
                n = sum(t["launches"] for t in ts)
                if n:
                    traffic = sum((t["read_bytes_per_launch"] + t["write_bytes_per_launch"]) * t["launches"] for t in ts) / n
            others = {}
            for r in recs:
                if r["kernel"].startswith(CONV_KERNELS):
                    continue
                o = others.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "bytes": 0.0})
                o["launches"] += 1
                o["ms"] += r["ms"]
                o["bytes"] += r["bytes"]
            for o in others.values():
                o["GB/s"] = o["bytes"] / max(o["ms"], 1e-9) / 1e6
                o["frac_hbm"] = o["GB/s"] / PEAK_HBM_GBS
                o["frac_copy"] = o["GB/s"] / COPY_CEILING_GBS
            # what actually bounds each of them (LAB_NOTES.md 3.1b / 3.2): only the Winograd transforms are HBM kernels
            bound_of = {"wino_input": ("hbm", "a hand-written 16 B / lane streaming copy of 0.5-4 GB sustains 5.3-5.6 TB/s read + write on this part (profiles/r03_copy_ceiling.md): frac_copy is GB/s over 5.45 TB/s"),
                        "wino_output": ("hbm", "see wino_input"),
                        "stem_pool": ("mfma", "7x7 stem as an implicit GEMM fused with the max-pool: 11.2 GFLOP executed per 1080p frame; its bytes are "
                                              "the 6.2 MB frame + the 33 MB pooled tensor, so GB/s says nothing about it"),
                        "upsample_argmax_shade": ("valu", "about 180 VALU instructions per 64 pixels (packed multiplies / adds of the 4-term bilerp in the "
                                                          "reference's operation order + compare / select per class): ~9.5 us of pure issue at 1080p; "
                                                          "its 11 MB are the compulsory bytes (PMC: 7.9 MB read + 8.3 MB written)")}
            for k, o in others.items():
                if k in bound_of:
                    o["bound"], o["note"] = bound_of[k]
            if "stem_pool" in others:
                st = [r for r in recs if r["kernel"] == "stem_pool"]
                others["stem_pool"]["TFLOP/s"] = sum(r["flops"] for r in st) / max(sum(r["ms"] for r in st), 1e-9) / 1e9
                others["stem_pool"]["frac_mfma"] = others["stem_pool"]["TFLOP/s"] / peak
            algo3 = sum(r["algo_flops"] for r in c3)
            algo_dom = sum(r["algo_flops"] for r in dom)
            ach = algo_dom / max(ms(dom), 1e-9) / 1e9
            out["roofline"] = {
                "bound": "mfma", "kernel": f"{dom_name} ({len(dom)} direct launches of the {len(conv)} conv launches of a frame; tile "
                                            f"configurations per layer shape are picked by measurement: {dict((k, round(v, 3)) for k, v in by_cfg.items())} ms)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": traffic, "traffic_note": "HBM bytes per launch from separate rocprofv3 --pmc passes "
                                                    "(profiles/traffic_latest.json); null if not collected for this shape",
                "launches": len(dom), "avg_launch_ms": ms(dom) / max(len(dom), 1),
                "avg_launch_note": "HIP events around each launch of the LAST frame of the timed region, which runs alone (the other "
                                   "context is drained on the GPU first): compare with profiles/rNN_solo_kernel_stats.csv (rocprofv3 of "
                                   "`bench.py --contexts-per-gpu 1`); in the default two-frames-in-flight trace a kernel shares the chip "
                                   "with the other frame's kernel and its wall time is longer",
                "flops_per_launch": algo_dom / max(len(dom), 1),
                "algorithmic_bytes_per_launch": sum(r["bytes"] for r in dom) / max(len(dom), 1),
                "note": "achieved = ALGORITHMIC (direct-convolution, BASELINE.md section 4) FLOPs of the layers these launches "
                        "compute / their HIP-event time; these launches run the convolution directly, so algorithmic == executed. "
                        "Stride-1 3x3 convs with Cin >= 128 run as Winograd F(6x6,3x3)-domain GEMMs instead: see `winograd`",
                "winograd": {"launches": len(wg), "gemm_ms": ms(wg), "transform_ms": ms(wino),
                             "executed_tflops": tf(wg), "executed_frac": tf(wg) / peak,
                             "algorithmic_tflops_incl_transforms": sum(r["algo_flops"] for r in wg) / max(ms(wg) + ms(wino), 1e-9) / 1e9,
                             "note": "algorithmic = the direct 3x3 convolution's FLOPs (4.7x the executed GEMM FLOPs with F(6x6,3x3) tiles) over GEMM + "
                                     "transform time: may exceed the MFMA peak, that is the point of the transform"},
                "all_convs": {"achieved": tf(conv), "frac": tf(conv) / peak, "ms": ms(conv)},
                "conv3x3": {"achieved": tf(c3), "frac": tf(c3) / peak, "ms": ms(c3), "winograd_transform_ms": ms(wino),
                            "direct_equivalent_tflops": algo3 / max(ms(c3) + ms(wino), 1e-9) / 1e9},
                "conv1x1": {"achieved": tf(c1), "frac": tf(c1) / peak, "ms": ms(c1)},
                "mfma_pipe": None if a.dtype != "f32s" else {
                    "note": "f32s issues three v_mfma_f32_32x32x16_f16 per f32 product; this is the matrix-pipe view of all conv launches "
                            "against the dense f16 peak (the measured shader clock under this load is ~1.87 GHz of 2.4, profiles/)",
                    "achieved": 3.0 * tf(conv), "peak": PEAK_F16_MFMA_TFLOPS, "frac": 3.0 * tf(conv) / PEAK_F16_MFMA_TFLOPS},
                "frame_kernel_ms": ms_all,
                "other_kernels": others,
            }
            if a.kernels:
                for r in recs:
                    sys.stderr.write(f"{r['name']:40s} {r['kernel']:24s} {r['ms']:8.3f} ms "
                                     f"{r['flops'] / max(r['ms'], 1e-9) / 1e9:8.1f} TF/s {r['bytes'] / max(r['ms'], 1e-9) / 1e6:9.1f} GB/s\n")
        out["config"]["conv_gflop_per_frame"] = flops["total"] / 1e9
        out["config"]["effective_conv_tflops"] = flops["total"] * fps / world / 1e12

        # the headline's contexts (streams, arenas, weights) are released before the side measurements start
        for c in ctxs:
            c.close()
        if world == 1 and a.dtype == "f32" and not a.no_split:
            out["f32_split_mode"] = split_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H)
            out["f32_split_fp8_mode"] = split_fp8_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H)
            try:
                out["f16hl_mode_1080p"] = f16hl_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H)
            except Exception as e:  # noqa: BLE001
                out["f16hl_mode_1080p"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["f16_mode_1080p"] = f16_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H)
            except Exception as e:  # noqa: BLE001
                out["f16_mode_1080p"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["int8_quantised_model"] = int8_mode_rate(a, dev, d_frames, d_masks, Wd, H)
            except Exception as e:  # noqa: BLE001
                out["int8_quantised_model"] = {"error": f"{type(e).__name__}: {e}"}
        default_workload = (Wd, H, a.scale, a.dtype, a.depth) == (1920, 1080, 1.0, "f32", 50)
        if world == 1 and not a.no_side:
            try:
                out["config"]["pcie_inclusive_frames_per_s"] = pcie_inclusive_rate(a, dev, blob, frames_np, a.dtype)
                out["config"]["pcie_inclusive_zero_copy_frames_per_s"] = pcie_inclusive_rate(a, dev, blob, frames_np, a.dtype, zero_copy=True)
                out["config"]["pcie_inclusive_note"] = ("the same frames from HOST memory through the infur_stream ring (H2D + forward + decode + mask D2H, wall "
                                                        "clock): pageable buffers through the copying submit / collect, and produced / consumed in place in the "
                                                        "ring's pinned slots (acquire / commit / collect_view / release, ABI 5); `value` is HBM-resident as the "
                                                        "bench contract asks")
            except Exception as e:  # noqa: BLE001
                out["config"]["pcie_inclusive_frames_per_s"] = None
                out["config"]["pcie_inclusive_note"] = f"{type(e).__name__}: {e}"
        if world == 1 and default_workload and not a.no_side:
            out["configs2_stream_scale05"] = stream_scale05_rate(a, dev, blob, frames_np)
            out["configs4_r101_f16_4k"] = r101_f16_4k_rate(a, dev)
            try:  # a side measurement must never cost the line its headline
                out["configs3_batch64_group"] = group_batch64_rate(a, blob)
            except Exception as e:  # noqa: BLE001
                out["configs3_batch64_group"] = {"error": f"{type(e).__name__}: {e}"}
            try:  # the same batch in the fast compliant mode: where host staging, not the GPU, would bound an 8-GPU node
                out["configs3_batch64_group_f16hl"] = group_batch64_rate(a, blob, "f16hl")
            except Exception as e:  # noqa: BLE001
                out["configs3_batch64_group_f16hl"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(blob, frames_np[0], a.cpu_seconds)
        emit(out)

    for c in ctxs:
        c.close()
    if multi:
        dist.destroy_process_group()


F16_CONTEXTS = 2  # the f16-rate modes: two frames in flight fill the dispatch gaps; a third only splits the CUs of kernels that
                  # fill the chip on their own (r4, same box: 1080p 411.7 / 406.7 / 398.9 frames/s with 2 / 3 / 4 contexts, 4K
                  # FCN-ResNet101 67.2 / 69.1 / 68.4 / 68.4 with 1 / 2 / 3 / 4: profiles/r04_contexts_sweep.log)


def resident_rate(a, dev, dtype, blob, d_frames, d_masks, Wd, H, scale, n_frames, contexts=None):
    """frames/s of the fused path on HBM-resident frames with a.contexts_per_gpu contexts (or `contexts`) taking the frames in turn
    (weights loaded once, replicated to the other contexts through infur_group_weights_broadcast)."""
    from infur_amd.processors import Context, FramePath, Group, Model, ModelCmd

    import torch

    K = max(1, contexts if contexts else a.contexts_per_gpu)
    # the contexts run on streams of torch's pool, as the headline's do (streams a context creates for itself one after the other
    # can share a hardware queue, and two frames in flight then behave like one: the side measurements read 8-10 % low)
    streams = [torch.cuda.Stream() for _ in range(K)]
    ctxs = [Context(device=dev, compute_aux=not a.no_aux, profile=False, dtype=dtype, stream=st.cuda_stream) for st in streams]
    Model(ctxs[0]).control(ModelCmd.LoadBlob(blob))
    if K > 1:
        with Group(ctxs) as g:
            g.weights_broadcast(0)
    fps_ = [FramePath(c, a.scale_mode) for c in ctxs]
    B = len(d_frames)

    def run(n):
        for i in range(n):
            fps_[i % K].advance_dev(d_frames[i % B].data_ptr(), Wd, H, scale, d_masks[i % B].data_ptr(), d_masks[i % B].numel())
        for c in ctxs:
            c.synchronize()

    # arenas, tile configurations, and the clocks back up after the idle gap: a side measurement starts after seconds of host work
    # (blob synthesis) with the GPU idle, and 16 frames of a fast mode (27 ms of the quantised model) do not bring the package
    # back to its steady clock -- warm up for at least 0.3 s of frames, like the headline's own warm-up steps do for it
    tw = time.perf_counter()
    run(max(4 * K, 16) if n_frames >= 32 else 2 * K)
    while n_frames >= 32 and time.perf_counter() - tw < 0.3:
        run(4 * K)
    t0 = time.perf_counter()
    run(n_frames)
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return n_frames / dt, dt / n_frames * 1e3


def executed_gflop_per_frame(a, dev, dtype, blob, frame_np, scale):
    """GFLOP the MFMA pipe actually executes for one frame in this mode (sum over the kernel records of one profiled frame:
    Winograd-domain GEMMs count their own FLOPs, not the direct convolution's) -- the denominator-side twin of
    conv_gflop_per_frame, so that a rate quoted against the direct-convolution FLOPs can be read as a roofline fraction."""
    from infur_amd.processors import Context, FramePath, Model, ModelCmd

    c = Context(device=dev, compute_aux=not a.no_aux, profile=True, dtype=dtype)
    try:
        Model(c).control(ModelCmd.LoadBlob(blob))
        fp = FramePath(c, a.scale_mode)
        fp.advance(frame_np, scale)
        fp.advance(frame_np, scale)
        return sum(r["flops"] for r in c.profile()) / 1e9
    finally:
        c.close()


def with_executed(roof, executed_gflop, fps):
    """adds the executed view to a whole-frame roofline object whose `frac` is quoted on algorithmic FLOPs"""
    if executed_gflop:
        roof["executed_tflops"] = executed_gflop * fps / 1e3
        roof["executed_frac"] = executed_gflop * fps / 1e3 / roof["peak"]
        roof["executed_note"] = ("executed = FLOPs the kernels of one frame actually issue (Winograd F(6x6) GEMMs count 1 / 5.06 of their "
                                 "convolution); `frac` is on ALGORITHMIC direct-convolution FLOPs and may exceed 1, `executed_frac` is the "
                                 "matrix-pipe fraction and cannot")
    return roof


def split_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H):
    """The same frames through INFUR_DTYPE_F32_SPLIT (f32 tensors, conv GEMMs on the f16 matrix cores with every
    operand split into an f16 hi+lo pair, f32 accumulation): reported NEXT TO the native-f32 headline, not as it.
    Its logits match the f32 CPU oracle as closely as the native f32 MFMA path does (tests/test_gpu_split.py)."""
    fps, ms = resident_rate(a, dev, "f32s", blob, d_frames, d_masks, Wd, H, a.scale, len(d_frames) * a.steps)
    flops = None
    try:
        from infur_amd import weights as W

        flops = W.conv_flops(H, Wd, depth=a.depth, aux=not a.no_aux)["total"]
    except Exception:
        pass
    out = {"value": fps, "unit": "frames/s", "dtype": "f32s", "ms_per_frame": ms, "contexts_per_gpu": max(1, a.contexts_per_gpu),
           "parity": "logits within 3e-5 of the f32 oracle enforced in tests/test_gpu_split.py (measured 2.5e-6 .. 4e-6, the "
                     "native f32 MFMA mode measures 3e-6 .. 4e-6); class maps identical outside a 3e-5 band",
           "run": "python bench.py --dtype f32s"}
    if flops:
        ceil = PEAK_F16_MFMA_TFLOPS / 3.0
        out["roofline"] = with_executed(
            {"bound": "mfma", "achieved": flops * fps / 1e12, "peak": ceil, "unit": "TFLOP/s", "frac": flops * fps / 1e12 / ceil,
             "note": "whole-frame algorithmic (direct-conv) FLOPs x frames/s against a third of the dense f16 MFMA peak "
                     "(three f16 MFMAs per f32 product); 14 convs run as Winograd F(6x6)"},
            executed_gflop_per_frame(a, dev, "f32s", blob, d_frames[0].cpu().numpy(), a.scale), fps)
    return out


def split_fp8_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H):
    """The same frames through INFUR_DTYPE_F32_SPLIT_FP8 ("f32x"): the split mode with its two cross terms hi*lo on the bf8
    (OCP e5m2) MX MFMA -- 2 MFMA units per product instead of 3, no tensor-level scales (e5m2 has f16's exponent range).  The
    cheapest arithmetic in the tree that stays inside north_star's 1e-3 on HOSTILE parameters too (tests/test_gpu_hostile.py:
    1.4e-4 max-abs, 1.0e-2 per element at 1080p; the f16 mode: 1.9e-3 / 1.3e-1).  A side measurement like f32_split_mode."""
    fps, ms = resident_rate(a, dev, "f32x", blob, d_frames, d_masks, Wd, H, a.scale, len(d_frames) * a.steps)
    out = {"value": fps, "unit": "frames/s", "dtype": "f32x", "ms_per_frame": ms, "contexts_per_gpu": max(1, a.contexts_per_gpu),
           "parity": "logits within 5e-4 of the f32 oracle enforced in tests/test_gpu_split.py (synthetic weights: measured 2-3e-4); "
                     "hostile parameters (heavy tails, per-channel scales over 3.2 decades) against a float64 reference, 1920x1080: "
                     "1.4e-4 max-abs / 1.0e-2 worst per-element with the F(6x6) default, 1.1e-4 / 7.0e-3 with winograd_tile = 4 (tests/test_gpu_hostile.py, "
                     "profiles/r04_hostile_probe.log); "
                     "round 3's e4m3 cross terms: 7.1e-4 / 5.1e-2; f16 mode: 1.9e-3 / 1.3e-1",
           "winograd_tile": "F(6x6) (F(4x4) costs 7.5 % of the frame -- 205 frames/s against 209 for f32s on one box -- for 1.1e-4 / 7.0e-3)",
           "run": "python bench.py --dtype f32x"}
    try:
        from infur_amd import weights as W

        flops = W.conv_flops(H, Wd, depth=a.depth, aux=not a.no_aux)["total"]
        ceil = PEAK_F16_MFMA_TFLOPS / 2.0
        out["roofline"] = with_executed(
            {"bound": "mfma", "achieved": flops * fps / 1e12, "peak": ceil, "unit": "TFLOP/s", "frac": flops * fps / 1e12 / ceil,
             "note": "whole-frame algorithmic (direct-conv) FLOPs x frames/s against half of the dense f16 MFMA peak (one f16 "
                     "MFMA + one bf8 MX MFMA of twice the depth at twice the rate per product); 14 convs run as Winograd F(6x6)"},
            executed_gflop_per_frame(a, dev, "f32x", blob, d_frames[0].cpu().numpy(), a.scale), fps)
    except Exception:
        pass
    return out


def f16hl_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H):
    """the headline's frames through INFUR_DTYPE_F16_HL ("f16hl", round 5) -- three-byte tensors (an f16 hi plane + an
    e5m2 lo plane per tensor, written by the producer's epilogue, staged by LDS-DMA), hi*hi on the f16 MFMA + both cross terms on the
    bf8 MX MFMA: two units per product like f32x, 3 bytes per element instead of 4 and no register staging.  The compliant fast mode:
    hostile parameters, 1920x1080, float64 reference: 1.5e-4 max-abs / 9.6e-3 per element (tests/test_gpu_hostile.py)."""
    fps, ms = resident_rate(a, dev, "f16hl", blob, d_frames, d_masks, Wd, H, a.scale, len(d_frames) * a.steps)
    out = {"value": fps, "unit": "frames/s", "dtype": "f16hl", "ms_per_frame": ms, "contexts_per_gpu": max(1, a.contexts_per_gpu),
           "parity": "logits within 6e-4 of the f32 oracle on the synthetic weights enforced in tests/test_gpu_hl.py (measured 0.9e-4 direct convs, "
                     "2.9-3.2e-4 with the F(6x6) default); hostile parameters against a float64 reference, 1920x1080: 1.5e-4 max-abs / 9.6e-3 worst "
                     "per-element (bars 1e-3 / 1e-2, tests/test_gpu_hostile.py); f32x 1.4e-4 / 1.0e-2, f16 1.9e-3 / 1.3e-1",
           "run": "python bench.py --dtype f16hl"}
    try:
        host = [f.cpu().numpy() for f in d_frames]
        out["pcie_inclusive_frames_per_s"] = pcie_inclusive_rate(a, dev, blob, host, "f16hl")
        out["pcie_inclusive_zero_copy_frames_per_s"] = pcie_inclusive_rate(a, dev, blob, host, "f16hl", zero_copy=True)
    except Exception as e:  # noqa: BLE001
        out["pcie_inclusive_note"] = f"{type(e).__name__}: {e}"
    try:
        from infur_amd import weights as W

        flops = W.conv_flops(H, Wd, depth=a.depth, aux=not a.no_aux)["total"]
        ceil = PEAK_F16_MFMA_TFLOPS / 2.0
        out["roofline"] = with_executed(
            {"bound": "mfma", "achieved": flops * fps / 1e12, "peak": ceil, "unit": "TFLOP/s", "frac": flops * fps / 1e12 / ceil,
             "note": "whole-frame algorithmic (direct-conv) FLOPs x frames/s against half of the dense f16 MFMA peak (per 32 channels: two f16 "
                     "MFMAs + one bf8 MX MFMA of four times their depth at twice their rate); 14 convs run as Winograd F(6x6)"},
            executed_gflop_per_frame(a, dev, "f16hl", blob, d_frames[0].cpu().numpy(), a.scale), fps)
    except Exception:
        pass
    return out


def f16_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H):
    """the headline's frames through INFUR_DTYPE_F16 (f16 tensors and operands, f32 accumulation -- configs[4]'s
    arithmetic on configs[1]'s workload).  A REDUCED-PRECISION mode: its logits are outside north_star's 1e-3 (1.5-2e-3), so it is
    reported for the record next to the compliant modes, never as `value`.  Every 3x3 runs directly (no Winograd in this mode), so
    algorithmic FLOPs == executed FLOPs and `frac` is the matrix-pipe fraction of the whole frame."""
    from infur_amd import weights as W
    from infur_amd.processors import Context, FramePath, Model, ModelCmd

    K = min(F16_CONTEXTS, max(1, a.contexts_per_gpu))
    fps, ms = resident_rate(a, dev, "f16", blob, d_frames, d_masks, Wd, H, a.scale, len(d_frames) * a.steps, contexts=K)
    flops = W.conv_flops(H, Wd, depth=a.depth, aux=not a.no_aux)["total"]
    out = {"value": fps, "unit": "frames/s", "dtype": "f16", "ms_per_frame": ms, "contexts_per_gpu": K,
           "workload": f"{Wd}x{H} frame, FCN-ResNet{a.depth}, f16 operands / f32 accumulation, scale {a.scale}, HBM resident",
           "parity": "NOT inside north_star's 1e-3: logits 1.5-2.1e-3 from the f32 oracle on the synthetic weights (tests state 5e-3: "
                     "tests/test_gpu_parity.py, tests/test_gpu_exporter.py), 1.9e-3 max-abs / 1.3e-1 worst per-element on hostile parameters "
                     "(tests/test_gpu_hostile.py); scripts/sim_f16_attribution.py: weights, branch tensors, trunk and head each carry "
                     "3e-4 .. 9e-4 of it, so no single tensor kept in f32 repairs it -- f32x is the compliant mode at this MFMA family",
           "run": "python bench.py --dtype f16",
           "roofline": {"bound": "mfma", "achieved": flops * fps / 1e12, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops * fps / 1e12 / PEAK_F16_MFMA_TFLOPS, "executed_frac": flops * fps / 1e12 / PEAK_F16_MFMA_TFLOPS,
                        "note": "whole-frame direct-convolution FLOPs x frames/s against the dense f16 MFMA peak; every conv runs directly, "
                                "so algorithmic == executed"}}
    # one profiled frame on one context: which kernel ran each layer, the fused conv3 -> conv1 pairs and the dominant kernel's rate
    try:
        c = Context(device=dev, compute_aux=not a.no_aux, profile=True, dtype="f16")
        try:
            Model(c).control(ModelCmd.LoadBlob(blob))
            fp = FramePath(c, a.scale_mode)
            fr = d_frames[0].cpu().numpy()
            for _ in range(3):
                fp.advance(fr, a.scale)
            recs = c.profile()
        finally:
            c.close()
        by = {}
        for r in recs:
            o = by.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "gflop": 0.0, "mbytes": 0.0})
            o["launches"] += 1
            o["ms"] += r["ms"]
            o["gflop"] += r["flops"] / 1e9
            o["mbytes"] += r["bytes"] / 1e6
        for o in by.values():
            o["TFLOP/s"] = o["gflop"] / max(o["ms"], 1e-9)
            o["GB/s"] = o["mbytes"] / max(o["ms"], 1e-9)
        dom = max((k for k in by if k.startswith(CONV_KERNELS)), key=lambda k: by[k]["ms"])
        out["roofline"]["dominant_kernel"] = {"kernel": dom, **by[dom], "frac": by[dom]["TFLOP/s"] / PEAK_F16_MFMA_TFLOPS}
        out["roofline"]["frame_kernel_ms_one_context"] = sum(r["ms"] for r in recs)
        out["kernels"] = by
        out["fused_pairs"] = sum(1 for r in recs if r["kernel"].startswith("conv1x1_b2b"))
        tj = os.path.join(ROOT, "profiles", "traffic_f16.json")
        if os.path.exists(tj) and (Wd, H, a.scale, a.depth) == (1920, 1080, 1.0, 50):
            out["roofline"]["traffic_file"] = "profiles/traffic_f16.json (HBM bytes per launch per kernel, separate --pmc passes)"
    except Exception as e:  # noqa: BLE001
        out["kernels"] = {"error": f"{type(e).__name__}: {e}"}
    try:  # the fast modes with the frames in HOST memory, copying and zero-copy (two lanes)
        host = [f.cpu().numpy() for f in d_frames]
        out["pcie_inclusive_frames_per_s"] = pcie_inclusive_rate(a, dev, blob, host, "f16", contexts=F16_CONTEXTS)
        out["pcie_inclusive_zero_copy_frames_per_s"] = pcie_inclusive_rate(a, dev, blob, host, "f16", zero_copy=True, contexts=F16_CONTEXTS)
    except Exception as e:  # noqa: BLE001
        out["pcie_inclusive_note"] = f"{type(e).__name__}: {e}"
    return out


def stream_scale05_rate(a, dev, blob, frames_np):
    """BASELINE configs[2]: the 1080p stream at scale 0.5 -- frames come from HOST memory through the depth-3
    infur_stream ring (H2D, scale + forward + decode, D2H on three HIP streams), i.e. PCIe inclusive, native f32.
    Reported next to the headline; never `value`.  roofline: direct-conv FLOPs of a 960x540 frame x frames/s
    against the f32 MFMA peak (Winograd layers make > 1 possible, as for the headline)."""
    from infur_amd import weights as W
    from infur_amd.app import StreamPath
    from infur_amd.processors import Context, FramePath, Model, ModelCmd

    from infur_amd.processors import Group

    K = min(2, max(1, a.contexts_per_gpu))  # (the host ring is measured with one or two compute lanes)
    lanes = [Context(device=dev, compute_aux=not a.no_aux) for _ in range(K)]
    ctx = lanes[0]
    Model(ctx).control(ModelCmd.LoadBlob(blob))
    if K > 1:
        with Group(lanes) as g:
            g.weights_broadcast(0)
    H, Wd = frames_np[0].shape[:2]
    sp = StreamPath(ctx, depth=3 if K == 1 else 4)
    for other in lanes[1:]:
        sp.add_lane(other)  # consecutive frames on alternating contexts
    n = 48
    frames = [(i, frames_np[i % len(frames_np)]) for i in range(n)]
    list(sp.run(frames[:6], 0.5))  # warm-up: ring allocation, tile configurations
    t0 = time.perf_counter()
    got = list(sp.run(frames, 0.5))
    dt = time.perf_counter() - t0
    sp.close()
    for c in lanes:
        c.close()
    # the same per-frame work with the frames already in HBM
    import torch

    oh, ow = got[0][1].shape[:2]
    d_in = [torch.from_numpy(f).cuda() for f in frames_np[:4]]
    d_out = [torch.empty((oh, ow, 4), dtype=torch.uint8, device="cuda") for _ in d_in]
    res_fps, _ = resident_rate(a, dev, "f32", blob, d_in, d_out, Wd, H, 0.5, 64)
    gflop = W.conv_flops(oh, ow, depth=50, aux=not a.no_aux)["total"] / 1e9
    fps = n / dt
    return {"value": fps, "unit": "frames/s", "dtype": "f32", "frames": n, "realtime_30fps_streams": fps / 30.0,
            "workload": f"{Wd}x{H} bgr24 frames from host memory -> scale 0.5 (nearest) -> {ow}x{oh} FCN-ResNet50 -> mask to host; "
                        f"infur_stream ring, {K} compute lane(s), PCIe inclusive",
            "hbm_resident_frames_per_s": res_fps, "conv_gflop_per_frame": gflop,
            "roofline": with_executed(
                {"bound": "mfma", "achieved": gflop * res_fps / 1e3, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": gflop * res_fps / 1e3 / PEAK_F32_MFMA_TFLOPS,
                 "note": "whole-frame algorithmic (direct-conv) FLOPs x HBM-resident frames/s; 14 convs run as Winograd F(6x6)"},
                executed_gflop_per_frame(a, dev, "f32", blob, frames_np[0], 0.5), res_fps)}


def pcie_inclusive_rate(a, dev, blob, frames_np, dtype="f32", zero_copy=False, contexts=None):
    """the headline workload with the frames in HOST memory -- H2D, forward, decode, mask D2H through the
    depth-3 infur_stream ring (two compute lanes when --contexts-per-gpu >= 2), wall-clock frames/s.  Reported next to
    `value` (which is HBM-resident by the bench contract), never as it.
    zero_copy (ABI 5): infur_stream_acquire / _commit / _collect_view / _release -- EVERY frame is produced into its pinned slot (one
    np.copyto per frame: what a decoder's read() into the slot costs the producer) and the masks are consumed in place; the copying form
    pays the same production into a pageable frame BEFORE the timed region (the frames exist) plus one pageable -> pinned memcpy per frame
    and one back per mask.  One mask of the zero-copy run is compared with the copying run's (frame ids and slot contents
    must belong together)."""
    from infur_amd.app import StreamPath
    from infur_amd.processors import Context, Group, Model, ModelCmd

    import numpy as np

    K = min(2, max(1, contexts if contexts else a.contexts_per_gpu))  # (one or two compute lanes)
    lanes = [Context(device=dev, compute_aux=not a.no_aux, dtype=dtype) for _ in range(K)]
    try:
        Model(lanes[0]).control(ModelCmd.LoadBlob(blob))
        if K > 1:
            with Group(lanes) as g:
                g.weights_broadcast(0)
        depth = 3 if K == 1 else 4
        sp = StreamPath(lanes[0], depth=depth)
        for other in lanes[1:]:
            sp.add_lane(other)
        n = 40
        frames = [(i, frames_np[i % len(frames_np)]) for i in range(n)]
        if not zero_copy:
            list(sp.run(frames[:6], a.scale))
            t0 = time.perf_counter()
            list(sp.run(frames, a.scale))
            dt = time.perf_counter() - t0
        else:
            check = {}

            def pump(items, keep=None):
                def drain():
                    fid, rgba, _ = sp.collect_view()
                    if keep is not None and fid == keep:
                        check["mask"] = rgba.copy()
                    sp.release()

                for fid, img in items:
                    if sp.pending() >= depth:
                        drain()
                    h, w = img.shape[:2]
                    np.copyto(sp.acquire(w, h, a.scale), img)  # the producer's own cost: every frame goes into its slot
                    sp.commit(w, h, a.scale, fid)
                while sp.pending():
                    drain()

            pump(frames[:8])
            t0 = time.perf_counter()
            pump(frames, keep=frames[-1][0])
            dt = time.perf_counter() - t0
            ref = dict(sp.run(frames[-1:], a.scale))[frames[-1][0]]  # the same frame through the copying calls
            if "mask" not in check or not (check["mask"] == ref).all():
                raise RuntimeError("zero-copy mask differs from the copying path's")
        sp.close()
        return n / dt
    finally:
        for c in lanes:
            c.close()


def int8_mode_rate(a, dev, d_frames, d_masks, Wd, H):
    """The same 1080p frames through a QUANTISED model -- the QOperator int8 form the reference's own tests load
    (fcn-resnet50-12-int8.onnx, predict_onnx.rs:357-381; here the seeded synthetic FCN-ResNet50 statically quantised by
    infur_amd/quantize.py): u8 activations x s8 weights on v_mfma_i32_32x32x32_i8, QLinearConv / QLinearAdd / DequantizeLinear
    arithmetic in the epilogues, bit-exact against the integer oracle (tests/test_gpu_quant.py).  A side measurement: another
    model file, not another way to compute the headline's."""
    from infur_amd import quantize
    from infur_amd import weights as W

    qblob = quantize.synth_qblob(depth=50)
    fps, ms = resident_rate(a, dev, "f32", qblob, d_frames, d_masks, Wd, H, a.scale, len(d_frames) * a.steps)
    gop = W.conv_flops(H, Wd, depth=50, aux=not a.no_aux)["total"] / 1e9
    peak = 2.0 * PEAK_F16_MFMA_TFLOPS  # dense i8 MFMA: twice the f16 rate (MI355X_MICROARCH.md: >= 3944 TOPS measured)
    out = {"value": fps, "unit": "frames/s", "dtype": "int8 (u8 x s8 -> i32)", "ms_per_frame": ms, "contexts_per_gpu": max(1, a.contexts_per_gpu),
           "model": "FCN-ResNet50, QOperator static quantisation of the synthetic weights (per-channel s8 weights, per-tensor u8 activations)",
           "parity": "every layer's u8 tensor, the dequantised logits and the mask bit-exact against oracle/infur_qoracle.py (tests/test_gpu_quant.py)",
           "roofline": {"bound": "mfma", "achieved": gop * fps / 1e3, "peak": peak, "unit": "TOP/s", "frac": gop * fps / 1e3 / peak,
                        "note": "direct-convolution integer ops (2 x MAC) x frames/s against the dense i8 MFMA peak"}}
    try:
        host = [f.cpu().numpy() for f in d_frames]
        out["pcie_inclusive_frames_per_s"] = pcie_inclusive_rate(a, dev, qblob, host, "f32")
        out["pcie_inclusive_zero_copy_frames_per_s"] = pcie_inclusive_rate(a, dev, qblob, host, "f32", zero_copy=True)
    except Exception as e:  # noqa: BLE001
        out["pcie_inclusive_note"] = f"{type(e).__name__}: {e}"
    return out


def r101_f16_4k_rate(a, dev):
    """BASELINE configs[4]: one 3840x2160 frame through FCN-ResNet101 on the f16 matrix cores (f16 operands, f32
    accumulation), scale 1.0, frame resident in HBM.  roofline: 14,278 GFLOP per frame x frames/s against the dense
    f16 MFMA peak (2.5 PFLOP/s)."""
    import torch

    from infur_amd import weights as W
    from infur_amd.processors import Context, FramePath, Model, ModelCmd

    H, Wd = 2160, 3840
    d_in = [torch.from_numpy(W.synth_frame(H, Wd, index=i)).cuda() for i in range(2)]
    d_out = [torch.empty((H, Wd, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    n = 24
    K = min(F16_CONTEXTS, max(1, a.contexts_per_gpu))
    fps, ms = resident_rate(a, dev, "f16", W.synth_blob(depth=101), d_in, d_out, Wd, H, 1.0, n, contexts=K)
    dt = ms * n / 1e3
    gflop = W.conv_flops(H, Wd, depth=101, aux=not a.no_aux)["total"] / 1e9
    return {"value": fps, "unit": "frames/s", "dtype": "f16", "ms_per_frame": dt / n * 1e3, "frames": n,
            "contexts_per_gpu": K,
            "workload": f"{Wd}x{H} frame, FCN-ResNet101 f16 operands / f32 accumulation, scale 1.0, HBM resident",
            "conv_gflop_per_frame": gflop,
            "roofline": {"bound": "mfma", "achieved": gflop * fps / 1e3, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": gflop * fps / 1e3 / PEAK_F16_MFMA_TFLOPS},
            "parity": "per-layer and whole-frame logits vs the f32 oracle at 5e-3 (tests/test_gpu_f16_r101.py); a "
                      "reduced-precision mode by definition, never the headline"}


def group_batch64_rate(a, blob, dtype="f32"):
    """BASELINE configs[3] through the C ABI's group calls, as a Rust host would run it (one process, one context per
    GPU): 64 distinct 1080p frames in HOST memory, `infur_group_weights_broadcast` (RCCL across
    distinct devices, device-to-device copies between contexts of one device), then `infur_group_batch_advance` -- a
    contiguous slice of the batch per context, one worker thread each, masks back in frame order, PCIe inclusive.
    Eight contexts over the GPUs this process can see (device i % n_visible): on the 1-GPU box all eight share the one
    GPU, so this is the code path of configs[3] and its single-GPU rate, not a scaling figure -- the line says which."""
    import torch

    from infur_amd import weights as W
    from infur_amd.processors import Context, FramePath, Group, Model, ModelCmd

    ndev = max(1, torch.cuda.device_count())
    n_ctx, n_frames = 8, 64
    from infur_amd.app import PinnedArray

    ctxs = [Context(device=i % ndev, compute_aux=not a.no_aux, dtype=dtype) for i in range(n_ctx)]
    pins = []
    try:
        t0 = time.perf_counter()
        Model(ctxs[0]).control(ModelCmd.LoadBlob(blob))
        t1 = time.perf_counter()
        with Group(ctxs) as g:
            g.weights_broadcast(0)
            t2 = time.perf_counter()
            frames = [W.synth_frame(a.height, a.width, index=1000 + i) for i in range(n_frames)]
            tf = time.perf_counter()
            g.advance_batch(frames[:2 * n_ctx], 1.0)  # arenas, rings and tile configurations of every context
            t3 = time.perf_counter()
            masks = g.advance_batch(frames, 1.0)
            dt = time.perf_counter() - t3
            # the same batch with frames and masks in caller-owned PINNED buffers (infur_host_alloc, ABI 5): DMA straight from / into
            # them, no pageable <-> pinned staging copies in the workers
            pin_in = [PinnedArray(f.shape) for f in frames]
            pin_out = [PinnedArray(m.shape) for m in masks]
            pins = pin_in + pin_out
            for p_, f in zip(pin_in, frames):
                p_.array[...] = f
            tp = time.perf_counter()
            got = g.advance_batch([p_.array for p_ in pin_in], 1.0, outs=[p_.array for p_ in pin_out])
            dt_pin = time.perf_counter() - tp
            pin_ok = all(bool((x == y).all()) for x, y in zip(got[::9], masks[::9]))
            uses_rccl = g.uses_rccl
            numa = g.worker_numa_nodes()
            # the same frames through the same eight contexts with the frames ALREADY in HBM (no rings, no PCIe): what is
            # left of the gap to the headline is the eight-way time-sharing of one GPU
            d_in = [torch.from_numpy(f).cuda() for f in frames[:16]]
            d_out = [torch.empty((a.height, a.width, 4), dtype=torch.uint8, device="cuda") for _ in d_in]
            fps8 = [FramePath(c) for c in ctxs]
            for rep in range(2):
                if rep == 1:
                    t4 = time.perf_counter()
                for i in range(n_frames):
                    fps8[i % n_ctx].advance_dev(d_in[i % 16].data_ptr(), a.width, a.height, 1.0, d_out[i % 16].data_ptr(), d_out[i % 16].numel())
                for c in ctxs:
                    c.synchronize()
            dt_res = time.perf_counter() - t4
        # frame order and replica agreement: a frame from the middle of another context's slice, recomputed on context 0
        k = 5 * (n_frames // n_ctx) + 3
        solo, _ = FramePath(ctxs[0]).advance(frames[k], 1.0)
        ok = bool((solo == masks[k]).all()) and len(masks) == n_frames
    finally:
        for p_ in pins:
            p_.close()
        for c in ctxs:
            c.close()
    return {"value": n_frames / dt, "unit": "frames/s", "dtype": dtype, "frames": n_frames, "contexts": n_ctx, "devices": min(ndev, n_ctx),
            "pinned_caller_buffers": {"frames_per_s": n_frames / dt_pin, "batch_ms": round(dt_pin * 1e3, 1), "masks_equal": pin_ok,
                                      "note": "frames and masks in memory from infur_host_alloc: the workers DMA them directly (no staging memcpy)"},
            "rccl_broadcast": bool(uses_rccl), "weights_load_ms": round((t1 - t0) * 1e3, 2), "weights_broadcast_ms": round((t2 - t1) * 1e3, 2),
            "masks_in_frame_order_and_equal_to_one_context": ok,
            "split": {"first_16_frames_ms_incl_ring_setup_and_tuning": round((t3 - tf) * 1e3, 1), "steady_batch_ms": round(dt * 1e3, 1),
                      "same_frames_hbm_resident_ms": round(dt_res * 1e3, 1), "hbm_resident_frames_per_s": n_frames / dt_res,
                      "pcie_and_host_copies_ms": round((dt - dt_res) * 1e3, 1), "bytes_per_frame_h2d_plus_d2h": a.width * a.height * 7,
                      "note": "the rings are persistent per context (created in the first batch); steady - resident = H2D + D2H + the "
                              "pageable <-> pinned memcpys that do not hide behind compute"},
            "worker_numa_nodes": numa,
            "workload": f"64 x {a.width}x{a.height} frames from host memory, infur_group_batch_advance over {n_ctx} contexts on "
                        f"{min(ndev, n_ctx)} visible GPU(s), slices of {n_frames // n_ctx} frames, masks to host (PCIe inclusive)",
            "note": "with one visible GPU the eight contexts share it: the path of BASELINE configs[3], not its scaling"}


def idims(ctx, w, h, factor):
    import ctypes as C

    ow, oh = C.c_uint32(0), C.c_uint32(0)
    rc = ctx.L.infur_scale_out_dims(w, h, factor, C.byref(ow), C.byref(oh))
    if rc:
        raise SystemExit(f"scale dims error {rc}")
    return ow.value, oh.value


if __name__ == "__main__":
    main()
