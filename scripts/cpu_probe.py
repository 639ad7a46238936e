"""Probe torch-CPU oracle speed vs thread count on the GPU box's host (bench cpu_baseline tuning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infur_amd import weights as W
from oracle.infur_oracle import COracle, TorchModel
blob = W.synth_blob()
co = COracle()
tm = TorchModel(blob)
fr = W.synth_frame(540, 960)
chw = co.pack_normalize(fr)
print("cpus", os.cpu_count(), torch.__config__.parallel_info().splitlines()[:3])
for th in (3, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    tm.forward_lowres(chw)
    t = time.perf_counter(); tm.forward_lowres(chw); el = time.perf_counter() - t
    print(f"threads {th}: 960x540 forward {el:.2f} s  ({W.conv_flops(540,960)['total']/el/1e9:.0f} GFLOP/s)", flush=True)
