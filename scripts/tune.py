#!/usr/bin/env python3
"""Measure the conv tile configurations for the BASELINE shapes on this GPU and write the tuning
database infur_amd/conv_tune_gfx950.txt (run on an MI355X:  python scripts/tune.py).
Each shape is tuned ROUNDS times in fresh contexts; the configuration that wins most often is kept."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import processors as P  # noqa: E402
from infur_amd import weights as W  # noqa: E402

ROUNDS = 3
JOBS = [  # (dtype, depth, (w, h), factor)
    ("f32", 50, (1920, 1080), 1.0), ("f32", 50, (1920, 1080), 0.5), ("f32", 50, (640, 480), 1.0), ("f32", 50, (320, 240), 1.0),
    ("f16", 50, (1920, 1080), 1.0), ("f16", 101, (3840, 2160), 1.0),
    ("f32s", 50, (1920, 1080), 1.0), ("f32s", 50, (1920, 1080), 0.5), ("f32s", 50, (640, 480), 1.0),
    ("f32x", 50, (1920, 1080), 1.0), ("f32x", 50, (1920, 1080), 0.5),
    ("i8", 50, (1920, 1080), 1.0), ("i8", 50, (1920, 1080), 0.5), ("i8", 50, (640, 480), 1.0),  # the quantised model (INFURQ01)
    ("f16hl", 50, (1920, 1080), 1.0), ("f16hl", 50, (1920, 1080), 0.5), ("f16hl", 50, (640, 480), 1.0), ("f16hl", 101, (3840, 2160), 1.0),
]
MODE = {"f32": "0", "f16": "1", "f32s": "2", "f32x": "3", "i8": "4", "f16hl": "5"}  # the `mode` column of the database


def main():
    # python scripts/tune.py [dtype ...]: re-measure only the given modes, keep the other lines
    only = set(sys.argv[1:])
    db = P.TUNE_DB
    votes = collections.defaultdict(collections.Counter)
    if os.path.exists(db):
        if only:
            redo = {MODE[d] for d in only}
            for ln in open(db):
                if ln.strip() and not ln.startswith("#") and ln.split()[11] not in redo:
                    *key, cfg = ln.split()
                    votes[" ".join(key)][cfg] += 1000
        os.rename(db, db + ".old")  # contexts created below must not import it
    blobs = {}
    for dtype, depth, (w, h), factor in JOBS:
        if only and dtype not in only:
            continue
        if dtype == "i8":
            from infur_amd import quantize

            blob = blobs.setdefault(("q", depth), quantize.synth_qblob(depth=depth))
        else:
            blob = blobs.setdefault(depth, W.synth_blob(depth=depth))
        fr = W.synth_frame(h, w)
        for _ in range(ROUNDS):
            c = P.Context(device=0, dtype="f32" if dtype == "i8" else dtype)
            P.Model(c).control(P.ModelCmd.LoadBlob(blob))
            P.FramePath(c).advance(fr, factor)
            for ln in c.tuning_text().splitlines():
                *key, cfg = ln.split()
                votes[" ".join(key)][cfg] += 1
            c.close()
        print(f"tuned {dtype} R{depth} {w}x{h} x{factor}: {len(votes)} shapes so far", flush=True)
    with open(db, "w") as f:
        f.write("# conv_igemm tile configuration per shape: H W Cin OH OW Cout KH stride dil batch res mode outf32 cfg\n")
        f.write("# measured on MI355X (gfx950) by scripts/tune.py; results are bit-identical for every configuration\n")
        for key in sorted(votes, key=lambda k: [int(x) for x in k.split()]):
            f.write(f"{key} {votes[key].most_common(1)[0][0]}\n")
    if os.path.exists(db + ".old"):
        os.remove(db + ".old")
    print("wrote", db, len(votes), "shapes")


if __name__ == "__main__":
    main()
