"""The pixel-pair arrangement of the quantised layer1 (infur_amd/csrc/infur_quant_model.cpp: model_load_q_dev / forward_q) as
integer arithmetic in numpy: a 64-channel NHWC tensor viewed as (H, W/2, 128) and convolved with the pair-arranged weights gives,
viewed back, exactly the plain convolution.  This is the index formula the loader implements -- the GPU tests check the bytes of the
whole network against the oracle; this one pins the formula itself, 1x1 and 3x3, without a GPU."""
import numpy as np
import pytest


def conv_nhwc(x, w, pad):
    """x [H][W][Cin] int, w [Cout][KH][KW][Cin] int -> [H][W][Cout] (stride 1, zero padding)"""
    H, W, _ = x.shape
    co, kh, kw, _ = w.shape
    xp = np.pad(x, ((pad, pad), (pad, pad), (0, 0)))
    y = np.zeros((H, W, co), np.int64)
    for ky in range(kh):
        for kx in range(kw):
            y += np.einsum("hwc,oc->hwo", xp[ky:ky + H, kx:kx + W, :], w[:, ky, kx, :])
    return y


def pair_weights(w):
    """[Cout][K][K][Cin] -> [2 Cout][K][K][2 Cin]: output row p * Cout + o, input column q * Cin + i; pair-column kx' holds tap
    kx = 2 (kx' - 1) + q - p + 1 of a 3x3 (a 1x1: the block diagonal)"""
    co, k, _, ci = w.shape
    w2 = np.zeros((2 * co, k, k, 2 * ci), w.dtype)
    for p in range(2):
        for q in range(2):
            for kxp in range(k):
                kx = (0 if p == q else -1) if k == 1 else 2 * (kxp - 1) + q - p + 1
                if 0 <= kx < k:
                    w2[p * co:(p + 1) * co, :, kxp, q * ci:(q + 1) * ci] = w[:, :, kx, :]
    return w2


@pytest.mark.parametrize("k,cin,cout,hw", [(1, 64, 64, (3, 8)), (1, 256, 64, (2, 6)), (1, 64, 256, (4, 2)), (3, 64, 64, (5, 8)), (3, 64, 64, (1, 2)), (3, 8, 4, (6, 10))])
def test_pair_arranged_conv_equals_the_plain_one(k, cin, cout, hw):
    rng = np.random.default_rng(k * 1000 + cin + cout)
    H, W = hw
    x = rng.integers(0, 256, (H, W, cin)).astype(np.int64) - 128  # the kernel's signed operand
    w = rng.integers(-128, 128, (cout, k, k, cin)).astype(np.int64)
    want = conv_nhwc(x, w, pad=k // 2)
    xv = x.reshape(H, W // 2, 2 * cin)  # two neighbouring pixels = one row
    got = conv_nhwc(xv, pair_weights(w), pad=k // 2).reshape(H, W, cout)  # pad 1 in PAIR units for the 3x3
    assert (got == want).all()
    # the row sums that go into the folded bias are the channel's own, once per pixel of the pair
    assert (pair_weights(w).sum(axis=(1, 2, 3)) == np.tile(w.sum(axis=(1, 2, 3)), 2)).all()
