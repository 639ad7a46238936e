"""dev: per-layer error vs torch for a given frame size / options"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infur_amd import weights as W
from infur_amd.processors import Context, Model, ModelCmd
from oracle.infur_oracle import COracle, TorchModel
blob = W.synth_blob(); tm = TorchModel(blob); co = COracle()
for (w, h) in [(1, 1), (7, 5), (33, 17), (8, 8), (130, 66)]:
    for mc in (0, 0xFFFFFFFF):
        c2 = Context(device=0, keep_activations=True, winograd_min_cin=mc)
        m = Model(c2).control(ModelCmd.LoadBlob(blob))
        fr = W.synth_frame(h, w, index=w + h)
        out = []; m.advance(fr, out)
        taps = {}; tm.forward_lowres(co.pack_normalize(fr), taps=taps)
        first = None; worst = 0
        for i, spec in enumerate(W.graph(50)):
            ref = taps[spec.name].numpy(); buf = np.empty(ref.shape, np.float32)
            a, b, d = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
            c2.check(c2.L.infur_debug_read_activation(c2.h, i, buf.ctypes.data, buf.size, C.byref(a), C.byref(b), C.byref(d)))
            e = np.abs(buf - ref).max() / max(np.abs(ref).max(), 1e-30)
            worst = max(worst, e)
            if e > 1e-4 and first is None: first = (spec.name, ref.shape, float(e))
        print(f"{w}x{h} wino_min_cin={mc:#x}: worst {worst:.2e} first bad {first}", flush=True)
        c2.close()
