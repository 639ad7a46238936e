"""PCIe-inclusive rates of the host-pointer entry points at 1080p (for DESIGN.md):  python scripts/host_path_rates.py [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infur_amd import weights as W
from infur_amd.app import StreamPath
from infur_amd.processors import Context, FramePath, Model, ModelCmd
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
c = Context(device=0, dtype=dtype); Model(c).control(ModelCmd.LoadBlob(W.synth_blob()))
frames = [W.synth_frame(1080, 1920, index=i) for i in range(4)]
fp = FramePath(c)
for f in frames[:2]: fp.advance(f, 1.0)
t = time.perf_counter(); n = 16
for i in range(n): fp.advance(frames[i % 4], 1.0)
print(f"[{dtype}] infur_frame_advance (host buffers, synchronous) 1080p scale 1.0: {n / (time.perf_counter() - t):.1f} frames/s")
for depth in (2, 3):
    sp = StreamPath(c, depth=depth)
    list(sp.run([(i, frames[i % 4]) for i in range(4)], 1.0))
    t = time.perf_counter(); n = 48
    out = list(sp.run([(i, frames[i % 4]) for i in range(n)], 1.0))
    print(f"infur_stream depth {depth} 1080p scale 1.0: {n / (time.perf_counter() - t):.1f} frames/s")
    sp.close()
m = fp.advance_batch([frames[i % 4] for i in range(16)], 1.0)
t = time.perf_counter(); m = fp.advance_batch([frames[i % 4] for i in range(32)], 1.0)
print(f"infur_batch_advance 32 x 1080p: {32 / (time.perf_counter() - t):.1f} frames/s")
# two contexts of the device: a second compute lane for the stream, a 2-context group for the batch
from infur_amd.processors import Group
c2 = Context(device=0, dtype=dtype)
with Group([c, c2]) as g:
    g.weights_broadcast(0)
    sp = StreamPath(c, depth=4); sp.add_lane(c2)
    list(sp.run([(i, frames[i % 4]) for i in range(8)], 1.0))
    t = time.perf_counter(); n = 64
    out = list(sp.run([(i, frames[i % 4]) for i in range(n)], 1.0))
    print(f"infur_stream depth 4, two lanes, 1080p scale 1.0: {n / (time.perf_counter() - t):.1f} frames/s")
    sp.close()
    g.advance_batch([frames[i % 4] for i in range(16)], 1.0)
    t = time.perf_counter(); g.advance_batch([frames[i % 4] for i in range(64)], 1.0)
    print(f"infur_group_batch_advance (2 contexts on one GPU) 64 x 1080p: {64 / (time.perf_counter() - t):.1f} frames/s")
