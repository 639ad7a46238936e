import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-side']
import bench
import torch
from infur_amd import weights as W
a = bench.parse()
blob = W.synth_blob()
H, Wd = 1080, 1920
d_frames = [torch.from_numpy(W.synth_frame(H, Wd, index=i)).cuda() for i in range(8)]
d_masks = [torch.empty((H, Wd, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]
for dt in ("f32", "f32s", "f32s", "f32", "f32s"):
    print(dt, bench.resident_rate(a, 0, dt, blob, d_frames, d_masks, Wd, H, 1.0, 48), flush=True)
# now with two idle f32 contexts alive (as in the bench main path)
from infur_amd.processors import Context, Model, ModelCmd, FramePath
idle = [Context(device=0) for _ in range(2)]
for c in idle:
    Model(c).control(ModelCmd.LoadBlob(blob))
    FramePath(c).advance_dev(d_frames[0].data_ptr(), Wd, H, 1.0, d_masks[0].data_ptr(), d_masks[0].numel()); c.synchronize()
print("with idle ctxs", bench.resident_rate(a, 0, "f32s", blob, d_frames, d_masks, Wd, H, 1.0, 48), flush=True)
idle2 = [Context(device=0, profile=True, stream=torch.cuda.Stream().cuda_stream) for _ in range(2)]
print("with idle ctxs + torch streams", bench.resident_rate(a, 0, "f32s", blob, d_frames, d_masks, Wd, H, 1.0, 48), flush=True)
