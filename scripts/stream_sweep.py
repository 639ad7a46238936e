"""BASELINE configs[2] (1080p frames from host memory, scale 0.5) through the infur_stream ring: ring depth x compute
lanes, next to the HBM-resident rate of the same work -- where the host path's 15 % go."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infur_amd import weights as W
from infur_amd.app import StreamPath
from infur_amd.processors import Context, Group, Model, ModelCmd

blob = W.synth_blob()
frames_np = [W.synth_frame(1080, 1920, index=i) for i in range(8)]
n = 96
for lanes_n in (1, 2, 3):
    for depth in (3, 4, 6, 8):
        lanes = [Context(device=0, profile=False) for _ in range(lanes_n)]
        Model(lanes[0]).control(ModelCmd.LoadBlob(blob))
        if lanes_n > 1:
            with Group(lanes) as g:
                g.weights_broadcast(0)
        sp = StreamPath(lanes[0], depth=depth)
        for o in lanes[1:]:
            sp.add_lane(o)
        fr = [(i, frames_np[i % 8]) for i in range(n)]
        list(sp.run(fr[:12], 0.5))
        t0 = time.perf_counter()
        list(sp.run(fr, 0.5))
        dt = time.perf_counter() - t0
        # host-side cost alone: time spent inside submit() calls
        ts = 0.0
        t1 = time.perf_counter()
        for i, f in fr[:depth]:
            a = time.perf_counter(); sp.submit(f, 0.5, i); ts += time.perf_counter() - a
        while sp.pending():
            sp.collect()
        print(f"lanes {lanes_n} depth {depth}: {n / dt:7.1f} frames/s   submit() {ts / depth * 1e3:.2f} ms/frame", flush=True)
        sp.close()
        for c in lanes:
            c.close()
