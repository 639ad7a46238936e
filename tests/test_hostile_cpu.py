"""CPU twin of tests/test_gpu_hostile.py: on the hostile parameter set (heavy-tailed weights, per-channel scales over > 3
decades, always-on channels, a frame with saturated regions -- tests/hostile.py) the two f32 oracles (plain C, torch-CPU)
must still agree with a float64 evaluation of the same network, otherwise they could not grade the HIP path there."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostile as H  # noqa: E402


@pytest.fixture(scope="module")
def hostile_blob():
    return H.hostile_blob()


def test_hostile_set_is_hostile_and_finite(hostile_blob):
    from infur_amd import weights as W
    from oracle.infur_oracle import COracle, TorchModel

    co = COracle()
    x = co.pack_normalize(H.saturated_frame(96, 128))
    taps = {}
    lo, la = TorchModel(hostile_blob, float64=True).forward_lowres(x, taps=taps)
    assert np.isfinite(lo.numpy()).all() and 0.5 < float(lo.abs().max()) < 50.0  # the reparametrisation keeps the logits O(1)
    cm = taps["backbone.layer3.5.conv3"].abs().amax(dim=(1, 2)).numpy()
    assert cm[cm > 0].max() / cm[cm > 0].min() > 1e3  # neighbouring channels three decades apart
    _, ts = W.unpack_blob(hostile_blob)
    w = dict((n, w_) for n, w_, _ in ts)["backbone.layer2.1.conv2"]
    assert np.abs(w).max() / np.median(np.abs(w)) > 300  # heavy tails and scales: a U(-a, a) tensor has max / median = 2


def test_f32_oracles_agree_with_float64_on_hostile_parameters(hostile_blob):
    from oracle.infur_oracle import COracle, TorchModel

    co = COracle()
    assert co.model_load(hostile_blob) == 0
    x = co.pack_normalize(H.saturated_frame(72, 104, index=1))
    ref, ref_aux = TorchModel(hostile_blob, float64=True).forward_lowres(x)
    t32, t32_aux = TorchModel(hostile_blob).forward_lowres(x)
    c32 = co.model_forward(x, full=False)
    for name, got, want in (("torch f32 out", t32.numpy(), ref.numpy()), ("torch f32 aux", t32_aux.numpy(), ref_aux.numpy()),
                            ("C out", c32["out_low"], ref.numpy()), ("C aux", c32["aux_low"], ref_aux.numpy())):
        e_max, e_rel = H.errors(got, want)
        print(f"{name}: max-abs/max-abs {e_max:.2e}, per-element (|ref| > 1e-2 max) {e_rel:.2e}")
        assert e_max < 2e-5 and e_rel < 1e-3


def test_every_seed_of_the_distribution_is_hostile_finite_and_distinct():
    """tests/hostile.py::SEEDS (round 6): the six parameter sets the fast modes' accuracy is stated over are six DIFFERENT sets, each a
    finite O(1) function with channels three decades apart, and the torch f32 oracle agrees with float64 on each (so it can grade there)."""
    import hashlib

    from oracle.infur_oracle import COracle, TorchModel

    co = COracle()
    x = co.pack_normalize(H.saturated_frame(72, 104, index=2))
    seen = set()
    for seed, base in H.SEEDS:
        blob = H.hostile_blob(seed=seed, base_seed=base)
        seen.add(hashlib.sha256(blob).hexdigest())
        taps = {}
        ref, ref_aux = TorchModel(blob, float64=True).forward_lowres(x, taps=taps)
        assert np.isfinite(ref.numpy()).all() and 0.5 < float(ref.abs().max()) < 50.0, hex(seed)
        cm = taps["backbone.layer3.5.conv3"].abs().amax(dim=(1, 2)).numpy()
        assert cm[cm > 0].max() / cm[cm > 0].min() > 1e3, hex(seed)
        t32, t32_aux = TorchModel(blob).forward_lowres(x)
        for got, want in ((t32.numpy(), ref.numpy()), (t32_aux.numpy(), ref_aux.numpy())):
            e_max, e_rel = H.errors(got, want)
            assert e_max < 2e-5 and e_rel < 1e-3, (hex(seed), e_max, e_rel)
    assert len(seen) == len(H.SEEDS)
