"""Soak: many frames of changing sizes / scales through one context and one stream ring; device memory in use must
settle (arena trimming) and results must stay identical for repeated inputs."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infur_amd import weights as W
from infur_amd.app import StreamPath
from infur_amd.processors import Context, FramePath, Model, ModelCmd

DT = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("i8",) else "f32"  # f32 | f32s | f32x | f16 | f16hl | i8
c = Context(device=0, dtype=DT, profile=False)
if len(sys.argv) > 1 and sys.argv[1] == "i8":  # the quantised model through the same soak
    from infur_amd import quantize

    Model(c).control(ModelCmd.LoadBlob(quantize.synth_qblob()))
else:
    Model(c).control(ModelCmd.LoadBlob(W.synth_blob()))
fp = FramePath(c)
sizes = [(1080, 1920), (480, 640), (720, 1280), (97, 161)]
frames = {s: W.synth_frame(*s, index=7) for s in sizes}
ref, used = {}, []
t0 = time.time()
for it in range(400):
    s = sizes[(it // 25) % len(sizes)]
    f = 0.5 if (it // 100) % 2 else 1.0
    rgba, _ = fp.advance(frames[s], f)
    h = hashlib.sha1(rgba.tobytes()).hexdigest()
    assert ref.setdefault((s, f), h) == h, (it, s, f)
    if it % 25 == 24:
        free, total = torch.cuda.mem_get_info()
        used.append((total - free) / 2**20)
print("sync path: %d frames in %.1f s; device MiB in use every 25 frames: %s" % (400, time.time() - t0, [round(u) for u in used]))
sp = StreamPath(c, depth=3)
n = 0
for rounds in range(4):
    for s in sizes:
        for fid, rgba in sp.run([(i, frames[s]) for i in range(40)], 1.0):
            assert hashlib.sha1(rgba.tobytes()).hexdigest() == ref[(s, 1.0)]
            n += 1
free, total = torch.cuda.mem_get_info()
print("stream path: %d frames, all equal to the synchronous path; device MiB in use %d" % (n, (total - free) / 2**20))
sp.close(); c.close()
free, total = torch.cuda.mem_get_info()
print("after close: device MiB in use %d" % ((total - free) / 2**20))
