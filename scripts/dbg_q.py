import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from infur_amd import weights as W
from infur_amd.processors import Context, FramePath, Model, ModelCmd
from oracle.infur_oracle import COracle
from oracle import infur_qoracle as Q
qblob = Q.synth_qblob()
co = COracle()
h, w = 96, 128
fr = W.synth_frame(h, w, index=h)
taps = {}
ref_lo, ref_aux = Q.qforward(qblob, co.pack_normalize(fr), taps)
c = Context(device=0, keep_activations=True)
m = Model(c).control(ModelCmd.LoadBlob(qblob))
rgba, _ = FramePath(c).advance(fr, 1.0)
buf = np.empty(64 << 18, np.float32)
for i, spec in enumerate(W.graph(50)):
    ref = taps[spec.name].astype(np.float32)
    cc, hh, ww = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    c.check(c.L.infur_debug_read_activation(c.h, i, buf.ctypes.data, buf.size, C.byref(cc), C.byref(hh), C.byref(ww)))
    full = buf[: cc.value * hh.value * ww.value].reshape(cc.value, hh.value, ww.value)
    if spec.role in ('cls','auxcls'): continue
    got = full[: ref.shape[0]]
    if (full[ref.shape[0]:] != 0).any(): print('   PAD nonzero', spec.name, full.shape, np.unique(full[ref.shape[0]:])[:8])
    d = got - ref
    if (d != 0).any() or i < 2: print(spec.name, got.shape, 'mismatch', int((d != 0).sum()), 'max|d|', np.abs(d).max(), 'interior mismatch', int((d[:, 4:-4, 4:-4] != 0).sum()),
          'per-channel mismatching', int(((d != 0).sum(axis=(1, 2)) > 0).sum()), 'mean d', d.mean())
    if i == 0:
        print(' got[0,:3,:6]', got[0, :3, :6], '\n ref', ref[0, :3, :6])
        print(' ch with errors:', np.nonzero((d != 0).sum(axis=(1, 2)))[0][:20])
lo, la = m.lowres()
d = lo - ref_lo
print('logits mismatch', int((d != 0).sum()), 'of', d.size, 'max', np.abs(d).max(), 'lo[:3,0,0]', lo[:3, 0, 0], ref_lo[:3, 0, 0])
meta, convs, adds = W.unpack_qblob(qblob)
cq = [c_ for c_ in convs if c_.name == 'classifier.4'][0]
print('cls4 y_scale', cq.y_scale, 'zp', cq.y_zp, 'ratio', (d[d != 0] / cq.y_scale)[:10])
print('per-channel mismatches', (d != 0).sum(axis=(1, 2)))
print('q got ch1..', np.rint(lo[:, 0, 0] / cq.y_scale + cq.y_zp), 'ref', np.rint(ref_lo[:, 0, 0] / cq.y_scale + cq.y_zp))
