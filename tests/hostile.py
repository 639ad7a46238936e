"""Hostile FCN-ResNet parameters and frames: what pretrained, BatchNorm-folded weights look like and
the friendly U(-a, a) synthetic set does not.  Test infrastructure only.

Starting from the seeded synthetic tensors (infur_amd/weights.py) the set is made hostile in ways that keep the network a
finite, well-defined function:
  * heavy tails: 1 % of every conv's weights are replaced by +-30 sigma outliers, then each output channel is rescaled to
    its old L2 norm (the variance through 50 layers is unchanged, the distribution is not);
  * per-channel scales over >= 3 decades: every internal channel c of the network is multiplied by s_c = 10^U(-1.6, 1.6)
    (log-uniform: 3.2 decades) on the producing conv (weights + bias) and divided out on every consumer's input channel --
    an exact reparametrisation of the same function (ReLU and max-pool commute with positive scales), so the logits stay
    O(1) while the activations of neighbouring channels differ by three orders of magnitude: the dynamic range Winograd's
    transforms and the f16 hi/lo splits have to survive.  The residual stream of a stage shares one scale vector (identity
    adds need it), the 1x1 / 3x3 branch channels get their own;
  * large-mean channels: a few internal channels get a bias of +25 output standard deviations (always-on channels).
The frame has saturated regions (0 and 255 blocks) next to noise."""
import numpy as np

from infur_amd import weights as W


def _rng(seed, stream, n):
    return W.uniform01(seed, 900_000 + stream, n).astype(np.float64)


def hostile_tensors(depth=50, seed=0xBADC0DE, decades=3.2, outlier_frac=0.01, outlier_sigma=30.0, big_mean=25.0, base_seed=None):
    """seed: the hostile modifications (which weights become outliers, the per-channel scales); base_seed: the synthetic tensors they
    start from (default: the library's).  SEEDS below = the sets the accuracy claims of the reduced-width modes are graded over."""
    specs = W.graph(depth)
    ts = [(c, w.astype(np.float64), b.astype(np.float64)) for c, w, b in W.synth_tensors(depth, seed=W.DEFAULT_SEED if base_seed is None else base_seed)]
    # ---- heavy-tailed weights, channel norms kept ----
    for i, (c, w, b) in enumerate(ts):
        if c.role in ("cls", "auxcls"):
            continue
        flat = w.reshape(c.cout, -1)
        norm0 = np.sqrt((flat ** 2).sum(1))
        u = _rng(seed, 10 * i, flat.size).reshape(flat.shape)
        sg = flat.std()
        mask = u < outlier_frac
        sign = np.where(_rng(seed, 10 * i + 1, flat.size).reshape(flat.shape) < 0.5, -1.0, 1.0)
        flat = np.where(mask, sign * outlier_sigma * sg, flat)
        flat *= (norm0 / np.sqrt((flat ** 2).sum(1)))[:, None]
        ts[i] = (c, flat.reshape(w.shape), b)
    # ---- always-on channels ----
    for i, (c, w, b) in enumerate(ts):
        if c.role in ("conv1", "conv2"):
            k = (np.arange(3) * 7 + i) % c.cout
            b = b.copy()
            b[k] += big_mean * np.sqrt((w.reshape(c.cout, -1) ** 2).sum(1))[k] * 0.6  # ~ output std for O(1) inputs
            ts[i] = (c, w, b)
    # ---- per-channel scales over `decades` decades: an exact reparametrisation ----
    def scales(stream, n):
        return 10.0 ** ((_rng(seed, stream, n) - 0.5) * decades)

    def scale_out(i, s):
        c, w, b = ts[i]
        ts[i] = (c, w * s[:, None, None, None], b * s)

    def scale_in(i, s):
        c, w, b = ts[i]
        ts[i] = (c, w / s[None, :, None, None], b)

    idx = {c.name: i for i, (c, _, _) in enumerate(ts)}
    stream_s = scales(1, 64)  # stem output (max-pool commutes with positive scales)
    scale_out(idx["backbone.conv1"], stream_s)
    i = 1
    stage = None
    l3_s = None
    while specs[i].role == "conv1":
        has_down = specs[i + 3].role == "down"
        if has_down:
            new_s = scales(100 + i, specs[i + 2].cout)
        a, b_ = scales(200 + i, specs[i].cout), scales(300 + i, specs[i + 1].cout)
        scale_in(i, stream_s)
        scale_out(i, a)
        scale_in(i + 1, a)
        scale_out(i + 1, b_)
        scale_in(i + 2, b_)
        if has_down:
            scale_in(i + 3, stream_s)
            scale_out(i + 3, new_s)
            stream_s = new_s
        scale_out(i + 2, stream_s)
        name = specs[i].name
        i += 4 if has_down else 3
        if name.startswith("backbone.layer3.") and specs[i].name.startswith("backbone.layer4."):
            l3_s = stream_s
    h = scales(7, specs[i].cout)
    scale_in(i, stream_s); scale_out(i, h); scale_in(i + 1, h)
    if i + 2 < len(specs):
        h2 = scales(8, specs[i + 2].cout)
        scale_in(i + 2, l3_s); scale_out(i + 2, h2); scale_in(i + 3, h2)
    return [(c, w.astype(np.float32), b.astype(np.float32)) for c, w, b in ts]


def hostile_blob(depth=50, **kw):
    return W.pack_blob(hostile_tensors(depth, **kw), depth, W.NUM_CLASSES, True)


# (seed, base_seed) of the hostile parameter sets a mode's accuracy is stated over (a distribution, not a sample):
# the first is the set every earlier round used; the others change the outlier positions and per-channel scales (seed) and, for the
# last three, the underlying synthetic tensors as well (base_seed)
SEEDS = ((0xBADC0DE, None), (0x5EED0001, None), (0x5EED0002, None), (0x5EED0003, 0x1F0A2027), (0x5EED0004, 0x00C0FFEE), (0x5EED0005, 0x0BADF00D))


def saturated_frame(h, w, index=0):
    """noise + gradient with hard 0 / 255 blocks (letterbox bars, blown highlights)"""
    fr = W.synth_frame(h, w, index=index).copy()
    fr[: h // 6] = 0
    fr[h - h // 7 :, : w // 2] = 255
    fr[h // 3 : h // 2, w // 4 : w // 2] = 255
    fr[h // 2 : 2 * h // 3, w // 2 : 3 * w // 4, 1] = 0
    return fr


def errors(got, ref):
    """(max-abs / max-abs, worst per-element relative error over |ref| > 1e-2 max |ref|)"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    m = np.abs(ref).max()
    d = np.abs(got - ref)
    big = np.abs(ref) > 1e-2 * m
    return float(d.max() / max(m, 1e-300)), float((d[big] / np.abs(ref[big])).max()) if big.any() else 0.0
