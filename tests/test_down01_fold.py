"""Host logic of the folded first two down-convs (hobot_stereonet_amd/csrc/sn_down01.hpp): the nine weight classes
sn_dbg_compose_down01 produces, applied as ONE 13x13 stride-4 convolution with the class chosen per output pixel, must
reproduce the oracle's two layers (oracle/stereonet_oracle.c so_features: conv 5x5 s2 p2, conv 5x5 s2 p2, no activation
between them — the network behind DnnNode::Run, stereonet_infer/src/stereonet_node.cpp:812).  CPU only: the fold runs
on the host at model load."""
import numpy as np
import pytest

from hobot_stereonet_amd import api


def _folded_reference(x, weff, beff):
    """x (3, Hp, Wp) float -> (32, Hp/4, Wp/4): per-pixel class = 3 * {first, inner, last row} + {first, inner, last col}"""
    _, hp, wp = x.shape
    ho, wo = hp // 4, wp // 4
    xp = np.zeros((3, hp + 12, wp + 12), np.float64)
    xp[:, 6:6 + hp, 6:6 + wp] = x                      # window of (oy, ox) starts at image (4 oy - 6, 4 ox - 6)
    out = np.empty((32, ho, wo), np.float64)
    for oy in range(ho):
        rc = 0 if oy == 0 else 2 if oy == ho - 1 else 1
        for ox in range(wo):
            cc = 0 if ox == 0 else 2 if ox == wo - 1 else 1
            cls = 3 * rc + cc
            win = xp[:, 4 * oy:4 * oy + 13, 4 * ox:4 * ox + 13]
            out[:, oy, ox] = np.tensordot(weff[cls].astype(np.float64), win, axes=3) + beff[cls]
    return out


@pytest.mark.parametrize("hp,wp,seed", [(16, 16, 0), (32, 48, 1), (48, 32, 2)])
def test_fold_matches_the_two_layers(oracle, hp, wp, seed):
    rng = np.random.default_rng(seed)
    w0 = (rng.standard_normal((32, 3, 5, 5)) / 8.0).astype(np.float32)
    b0 = rng.standard_normal(32).astype(np.float32)
    w1 = (rng.standard_normal((32, 32, 5, 5)) / 28.0).astype(np.float32)
    b1 = rng.standard_normal(32).astype(np.float32)
    x = (rng.integers(-128, 128, (3, hp, wp)).astype(np.float32) / 128.0)
    ref = oracle.conv2d(oracle.conv2d(x, w0, b0, 2, 2, 1), w1, b1, 2, 2, 1)
    weff, beff = api.compose_down01(w0, b0, w1, b1)
    assert weff.shape == (9, 32, 3, 13, 13) and beff.shape == (9, 32)
    got = _folded_reference(x, weff, beff)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    # the classes really differ where down-conv 1 pads: using the inner class on the border must NOT reproduce it
    inner = np.tensordot(weff[4].astype(np.float64), np.pad(x, ((0, 0), (6, 6), (6, 6)))[:, 0:13, 0:13], axes=3) + beff[4]
    assert np.abs(inner - ref[:, 0, 0]).max() > 1e-3


def test_inner_class_is_the_plain_composition():
    """Weff[inner][o, i, u, v] = sum over c and (ky1, ky0), (kx1, kx0) with 2 ky1 + ky0 = u, 2 kx1 + kx0 = v."""
    rng = np.random.default_rng(7)
    w0 = rng.standard_normal((32, 3, 5, 5)).astype(np.float32)
    w1 = rng.standard_normal((32, 32, 5, 5)).astype(np.float32)
    b0 = rng.standard_normal(32).astype(np.float32)
    b1 = rng.standard_normal(32).astype(np.float32)
    weff, beff = api.compose_down01(w0, b0, w1, b1)
    want = np.zeros((32, 3, 13, 13), np.float64)
    for ky1 in range(5):
        for kx1 in range(5):
            want[:, :, 2 * ky1:2 * ky1 + 5, 2 * kx1:2 * kx1 + 5] += np.einsum("oc,cikl->oikl", w1[:, :, ky1, kx1].astype(np.float64),
                                                                              w0.astype(np.float64))
    assert np.abs(weff[4] - want).max() <= 1e-5 * np.abs(want).max()
    assert np.abs(beff[4] - (b1 + w1.astype(np.float64).sum(axis=(2, 3)) @ b0)).max() <= 1e-5 * np.abs(beff[4]).max()
    # first-row class: down-conv 1's taps ky1 = 0, 1 are dropped, so window rows 0..3 carry no weight
    assert np.abs(weff[1][:, :, :4, :]).max() == 0.0 and np.abs(weff[7][:, :, 12:, :]).max() == 0.0
