#!/usr/bin/env python3
"""bench.py -- 1080p frames/s of the FCN-ResNet50 segmentation path on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (packed BGR frame -> [scale] -> fused pre-proc + stem ->
FCN-ResNet50 incl. aux head -> bilinear up-sample + argmax + shade -> RGBA mask) over one batch
of ``--frames-per-step`` distinct synthetic frames per GPU, inputs and outputs resident in HBM.
N > 1 is launched by torch.distributed.run, one rank per GPU: the weight blob is broadcast
once over RCCL (timed separately, outside the region), frames are sharded with no data-path
collective ("weak" scaling: per-GPU work is fixed).  Rank 0 prints ONE JSON line.

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
    ap.add_argument("--no-aux", action="store_true", help="skip the aux head (the ONNX graph always evaluates it)")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="CPU baseline budget")
    ap.add_argument("--kernels", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--no-split", action="store_true", help="skip the f32_split_mode side measurement")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "f32s"],
                    help="conv-stack arithmetic: f32 (BASELINE configs[1], the default and the parity mode) or f16 "
                         "operands with f32 accumulation (configs[4]'s mode)")
    ap.add_argument("--winograd-min-cin", type=int, default=0,
                    help="f32 stride-1 3x3 convs with Cin >= this run as Winograd F(2x2,3x3); 0 = library default, -1 = never")
    ap.add_argument("--winograd-tile", type=int, default=0, choices=[0, 2, 4], help="Winograd output tile (0 = default)")
    ap.add_argument("--depth", type=int, default=50, choices=[50, 101], help="backbone: FCN-ResNet50 (default) / 101")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise the "
                         "multi-process path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def cpu_baseline(blob, frame, budget_s):
    """The oracle's whole path on the host cores: pre-proc (C) + FCN-ResNet50 (torch-CPU /
    oneDNN) + up-sample + ColorCode (C).  Bounded sample: one whole 1080p frame per thread
    count.  oneDNN's best thread count on a many-core host is far below the core count (256
    threads is ~100x slower than 16 on the GPU box), so a short sweep picks the fastest; the
    reference's own setting (ONNX Runtime pinned to 3 intra-op threads, predict_onnx.rs:292)
    is timed too and reported beside it."""
    import torch

    from oracle.infur_oracle import COracle, TorchModel

    cores = os.cpu_count() or 1
    co = COracle(threads=min(cores, 16))
    tm = TorchModel(blob)
    h, w = frame.shape[:2]

    def one_frame():
        t0 = time.perf_counter()
        chw = co.pack_normalize(frame)
        lo, _ = tm.forward_lowres(chw)
        full = co.upsample_bilinear(lo.numpy(), h, w)
        co.colorcode(full)
        return time.perf_counter() - t0

    results, spent = {}, 0.0
    for th in (16, 32, 3):
        if th > cores or spent > budget_s:
            continue
        torch.set_num_threads(th)
        results[th] = one_frame()
        spent += results[th]
    best = min(results, key=results.get)
    return {
        "value": 1.0 / results[best], "unit": "frames/s", "cores": best, "kind": "port",
        "sample": f"one whole {w}x{h} frame through the oracle path (C pre-proc, torch-CPU oneDNN FCN-ResNet50 incl. aux "
                  f"head, C up-sample + ColorCode) per thread count {sorted(results)}; {spent:.1f} s of CPU wall time; "
                  f"host has {cores} logical cores",
        "frames_per_s_by_threads": {str(k): 1.0 / v for k, v in sorted(results.items())},
        "reference_note": "the reference pins ONNX Runtime to 3 intra-op threads (predict_onnx.rs:292); the reference "
                          "itself cannot run here (no cargo / onnxruntime / model file)",
    }


def main():
    a = parse()
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
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    ndev = torch.cuda.device_count()
    dev = local_rank if a.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)
    coll_dev = f"cuda:{dev}" if a.backend == "nccl" else "cpu"
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev}"))
        else:
            dist.init_process_group("gloo")

    stream = torch.cuda.Stream()
    ctx = Context(device=dev, compute_aux=not a.no_aux, profile=not a.no_profile, stream=stream.cuda_stream, dtype=a.dtype,
                  winograd_min_cin=a.winograd_min_cin & 0xFFFFFFFF, winograd_tile=a.winograd_tile)

    # ---- weights: rank 0 synthesises, RCCL broadcast over xGMI, every rank repacks locally ----
    blob = W.synth_blob(depth=a.depth) if rank == 0 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbytes = idist.load_model_everywhere(ctx, blob, coll_device=coll_dev)
    torch.cuda.synchronize()
    load_ms = (time.perf_counter() - t0) * 1e3

    # ---- synthetic frames, resident in HBM ----
    H, Wd, B = a.height, a.width, a.frames_per_step
    frames_np = [W.synth_frame(H, Wd, index=rank * B + i) for i in range(B)]
    d_frames = [torch.from_numpy(f).cuda() for f in frames_np]
    rc_w, rc_h = idims(ctx, Wd, H, a.scale)
    d_masks = [torch.empty((rc_h, rc_w, 4), dtype=torch.uint8, device="cuda") for _ in range(B)]
    fp = FramePath(ctx, a.scale_mode)
    torch.cuda.synchronize()

    def step(profile_last=False):
        for i in range(B):
            if profile_last and i == B - 1:
                ctx.L.infur_profile_enable(ctx.h, 1)  # per-kernel HIP events for this frame only
            fp.advance_dev(d_frames[i].data_ptr(), Wd, H, a.scale, d_masks[i].data_ptr(), d_masks[i].numel())

    for _ in range(a.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # Per-kernel HIP events bracket every launch of the LAST frame of the timed region only (the
    # roofline sample); recording them on all frames costs ~2.5 % of throughput in event packets.
    ctx.L.infur_profile_enable(ctx.h, 0)
    for k in range(a.steps):
        step(profile_last=(not a.no_profile) and k == a.steps - 1)
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
            "frames_per_step_per_gpu": B, "sharding": f"frames x{world}, no data-path collective", "backend": a.backend if world > 1 else None,
            "weights": f"synthetic seed {W.DEFAULT_SEED:#x}, {nbytes / 1e6:.1f} MB blob",
            "weights_load_ms": round(load_ms, 2), "ms_per_frame_per_gpu": elapsed / (a.steps * B) * 1e3,
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel family from the HIP events of the last timed frame ----
        flops = W.conv_flops(rc_h, rc_w, depth=a.depth, aux=not a.no_aux)
        if not a.no_profile:
            recs = ctx.profile()
            k3 = {c.name for c in W.graph(a.depth) if c.k == 3}
            # f32s: three f16 MFMAs per product -> the ceiling for f32-equivalent FLOPs is a third of the f16 peak
            peak = {"f32": PEAK_F32_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS, "f32s": PEAK_F16_MFMA_TFLOPS / 3.0}[a.dtype]
            conv = [r for r in recs if r["kernel"].startswith("conv_igemm_")]
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
            if os.path.exists(tj) and (Wd, H, a.scale, a.depth) == (1920, 1080, 1.0, 50):
                parts = dom_name.split("<")[1].rstrip(">").split(",")  # "64,64" or "64,64,1buf" / "256,256,1frag"
                bm, bn = parts[0], parts[1]
                nbuf = {"1buf": "1", "1frag": "3"}.get(parts[2], "2") if len(parts) > 2 else "2"
                waves = {"128,256": "2, 4", "256,128": "4, 2", "256,32": "4, 1", "256,256": "2, 4"}.get(f"{bm},{bn}", "2, 2")
                el = "_Float16, _Float16" if a.dtype == "f16" else "float, float"
                split = {"f32": "false", "f16": "false", "f32s": "true"}[a.dtype]
                # all instantiations of this tile (plain / 1x1-GEMM addressing / residual prefetch), launch-weighted
                pre = f"conv_igemm_kernel<{el}, {bm}, {bn}, {waves}, {nbuf}, {split}"
                ts = [t for k, t in json.load(open(tj))["kernels"].items() if k.startswith(pre)]
                n = sum(t["launches"] for t in ts)
                if n:
                    traffic = sum((t["read_bytes_per_launch"] + t["write_bytes_per_launch"]) * t["launches"] for t in ts) / n
            others = {}
            for r in recs:
                if r["kernel"].startswith("conv_igemm"):
                    continue
                o = others.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "bytes": 0.0})
                o["launches"] += 1
                o["ms"] += r["ms"]
                o["bytes"] += r["bytes"]
            for o in others.values():
                o["GB/s"] = o["bytes"] / max(o["ms"], 1e-9) / 1e6
                o["frac_hbm"] = o["GB/s"] / PEAK_HBM_GBS
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
                "flops_per_launch": algo_dom / max(len(dom), 1),
                "algorithmic_bytes_per_launch": sum(r["bytes"] for r in dom) / max(len(dom), 1),
                "note": "achieved = ALGORITHMIC (direct-convolution, BASELINE.md section 4) FLOPs of the layers these launches "
                        "compute / their HIP-event time; these launches run the convolution directly, so algorithmic == executed. "
                        "Stride-1 3x3 convs with Cin >= 256 run as Winograd F(4x4,3x3)-domain GEMMs instead: see `winograd`",
                "winograd": {"launches": len(wg), "gemm_ms": ms(wg), "transform_ms": ms(wino),
                             "executed_tflops": tf(wg), "executed_frac": tf(wg) / peak,
                             "algorithmic_tflops_incl_transforms": sum(r["algo_flops"] for r in wg) / max(ms(wg) + ms(wino), 1e-9) / 1e9,
                             "note": "algorithmic = the direct 3x3 convolution's FLOPs (4x the executed GEMM FLOPs) over GEMM + "
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

        if world == 1 and a.dtype == "f32" and not a.no_split:
            out["f32_split_mode"] = split_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(blob, frames_np[0], a.cpu_seconds)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def split_mode_rate(a, dev, blob, d_frames, d_masks, Wd, H):
    """The same frames through INFUR_DTYPE_F32_SPLIT (f32 tensors, conv GEMMs on the f16 matrix cores with every
    operand split into an f16 hi+lo pair, f32 accumulation): reported NEXT TO the native-f32 headline, not as it.
    Its logits match the f32 CPU oracle as closely as the native f32 MFMA path does (tests/test_gpu_split.py)."""
    import torch

    from infur_amd.processors import Context, FramePath, Model, ModelCmd

    stream = torch.cuda.Stream()
    ctx = Context(device=dev, compute_aux=not a.no_aux, profile=False, stream=stream.cuda_stream, dtype="f32s")
    Model(ctx).control(ModelCmd.LoadBlob(blob))
    fp = FramePath(ctx, a.scale_mode)
    B = len(d_frames)

    def step():
        for i in range(B):
            fp.advance_dev(d_frames[i].data_ptr(), Wd, H, a.scale, d_masks[i].data_ptr(), d_masks[i].numel())

    step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.close()
    return {"value": B * a.steps / dt, "unit": "frames/s", "dtype": "f32s", "ms_per_frame": dt / (B * a.steps) * 1e3,
            "parity": "logits within 3e-5 of the f32 oracle enforced in tests/test_gpu_split.py (measured 2.5e-6 .. 4e-6, the "
                      "native f32 MFMA mode measures 3e-6 .. 4e-6); class maps identical outside a 3e-5 band",
            "run": "python bench.py --dtype f32s"}


def idims(ctx, w, h, factor):
    import ctypes as C

    ow, oh = C.c_uint32(0), C.c_uint32(0)
    rc = ctx.L.infur_scale_out_dims(w, h, factor, C.byref(ow), C.byref(oh))
    if rc:
        raise SystemExit(f"scale dims error {rc}")
    return ow.value, oh.value


if __name__ == "__main__":
    main()
