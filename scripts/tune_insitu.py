#!/usr/bin/env python3
"""In-situ refinement of the tuning database (round 4).  `pick_cfg` / scripts/tune.py time every candidate configuration of a layer
shape IN ISOLATION -- the same launch repeated, its operands warm in L2 / Infinity Cache -- and that is not always the order
the candidates have inside a frame (configuration 20 of the f16 mode won classifier.0 in isolation by 2-4 % and lost it in the
frame by 12 %: LAB_NOTES, round 4).  This script measures where it matters: one context with the current database, whole frames
with per-kernel HIP events; then, one shape at a time, every configuration is imported for that shape alone and the frame is run
again -- the layers whose kernel changed are compared with their baseline times (median of N frames) and the shape keeps the
configuration with the smallest in-frame time, if it beats the database's by more than MIN_GAIN.  Coordinate descent, one pass.

    python scripts/tune_insitu.py [dtype ...]      (on an MI355X; rewrites infur_amd/conv_tune_gfx950.txt in place)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

JOBS = [("f32", 50, (1920, 1080), 1.0), ("f16", 50, (1920, 1080), 1.0), ("f16", 101, (3840, 2160), 1.0),
        ("f32s", 50, (1920, 1080), 1.0), ("f32x", 50, (1920, 1080), 1.0), ("i8", 50, (1920, 1080), 1.0),
        ("f32@0.5", 50, (1920, 1080), 0.5), ("f32@480", 50, (640, 480), 1.0), ("f32s@0.5", 50, (1920, 1080), 0.5)]
MODE = {"f32": "0", "f16": "1", "f32s": "2", "f32x": "3", "i8": "4"}
# a change must win 3 % on the layers it touches (INSITU_MIN_GAIN / INSITU_FRAMES / INSITU_CFGS="19,20": a closer look at few candidates)
FRAMES, MIN_GAIN = int(os.environ.get("INSITU_FRAMES", 7)), float(os.environ.get("INSITU_MIN_GAIN", 0.03))
NCFG = 22
CANDS = [int(x) for x in os.environ["INSITU_CFGS"].split(",")] if os.environ.get("INSITU_CFGS") else list(range(NCFG))


def frame_times(c, fp, fr, factor):
    per = {}
    for it in range(FRAMES + 2):
        fp.advance(fr, factor)
        if it < 2:
            continue
        for i, r in enumerate(c.profile()):
            per.setdefault((i, r["name"]), []).append((r["kernel"], r["ms"]))
    return {k: (v[0][0], float(np.median([m for _, m in v]))) for k, v in per.items()}


def main():
    only = set(sys.argv[1:])
    db_lines = [ln for ln in open(P.TUNE_DB) if ln.strip() and not ln.startswith("#")]
    db = {" ".join(ln.split()[:13]): ln.split()[13] for ln in db_lines}
    changed = 0
    for job, depth, (w, h), factor in JOBS:
        if only and job not in only:
            continue
        dtype = job.split("@")[0]
        if dtype == "i8":
            from infur_amd import quantize

            blob = quantize.synth_qblob(depth=depth)
        else:
            blob = W.synth_blob(depth=depth)
        fr = W.synth_frame(h, w)
        # the shapes this job touches: one frame on a context WITHOUT the database (it exports exactly what it tuned)
        os.rename(P.TUNE_DB, P.TUNE_DB + ".off")
        try:
            c0 = P.Context(device=0, dtype="f32" if dtype == "i8" else dtype)
            P.Model(c0).control(P.ModelCmd.LoadBlob(blob))
            P.FramePath(c0).advance(fr, factor)
            keys = [" ".join(ln.split()[:13]) for ln in c0.tuning_text().splitlines()]
            c0.close()
        finally:
            os.rename(P.TUNE_DB + ".off", P.TUNE_DB)
        c = P.Context(device=0, dtype="f32" if dtype == "i8" else dtype, profile=True)
        P.Model(c).control(P.ModelCmd.LoadBlob(blob))
        fp = P.FramePath(c)
        base = frame_times(c, fp, fr, factor)
        cur = {" ".join(ln.split()[:13]): ln.split()[13] for ln in c.tuning_text().splitlines()}
        total0 = sum(m for _, m in base.values())
        print(f"== {dtype} R{depth} {w}x{h}: {len(keys)} shapes, frame kernels {total0:.3f} ms", flush=True)
        for key in keys:
            if key.split()[10] == "3" or key not in cur:  # (the fused conv3 -> conv1 decision changes the record list: scripts/b2b_ab.py)
                continue
            best, best_delta = cur[key], 0.0
            for k in CANDS:
                if str(k) == cur[key]:
                    continue
                txt = f"{key} {k}\n".encode()
                c.check(c.L.infur_tune_import(c.h, txt, len(txt)))
                try:
                    got = frame_times(c, fp, fr, factor)
                except P.InfurError:
                    got = None
                now = {" ".join(ln.split()[:13]): ln.split()[13] for ln in c.tuning_text().splitlines()}
                if got is None or now.get(key) != str(k) or set(got) != set(base):  # not a candidate for this shape (the library re-tuned it)
                    continue
                touched = [x for x in base if got[x][0] != base[x][0]]
                if not touched:
                    continue
                t0, t1 = sum(base[x][1] for x in touched), sum(got[x][1] for x in touched)
                delta = (t1 - t0) / t0
                if delta < best_delta - 1e-9 and delta < -MIN_GAIN:
                    best, best_delta = str(k), delta
            txt = f"{key} {best}\n".encode()
            c.check(c.L.infur_tune_import(c.h, txt, len(txt)))
            if best != cur[key]:
                print(f"   {key}: {cur[key]} -> {best}  ({best_delta:+.1%} on the layers it runs)", flush=True)
                cur[key] = best
                changed += 1
                base = frame_times(c, fp, fr, factor)  # the new baseline
            db[key] = best
        total1 = sum(m for _, m in frame_times(c, fp, fr, factor).values())
        print(f"   frame kernels {total0:.3f} -> {total1:.3f} ms", flush=True)
        c.close()
    with open(P.TUNE_DB, "w") as f:
        f.write("# conv_igemm tile configuration per shape: H W Cin OH OW Cout KH stride dil batch res mode outf32 cfg\n")
        f.write("# measured on MI355X (gfx950) by scripts/tune.py (isolated launches, majority of 3) and refined in situ by scripts/tune_insitu.py;\n")
        f.write("# results are bit-identical for every configuration\n")
        for key in sorted(db, key=lambda k: [int(x) for x in k.split()]):
            f.write(f"{key} {db[key]}\n")
    print("wrote", P.TUNE_DB, len(db), "shapes,", changed, "changed")


if __name__ == "__main__":
    main()
