"""Host logic: the sum-preserving fp16 rounding of the refinement towers' 3x3 weights (SN_PREC_F16, model load;
hobot_stereonet_amd/csrc/stereonet_hip.hip round_kernel_sum_preserving).  The model file is opaque to the reference
(stereonet_infer/src/stereonet_node.cpp:131-136 only checks that it exists); how its float weights become fp16 operands
is this build's business, and this is the rule.  CPU only."""
import numpy as np

from hobot_stereonet_amd import api, weights


def _neighbours(w):
    n = w.astype(np.float16)
    nf = n.astype(np.float64)
    lo = np.where(nf <= w, nf, np.nextafter(n, np.float16(-np.inf)).astype(np.float64))
    hi = np.where(nf >= w, nf, np.nextafter(n, np.float16(np.inf)).astype(np.float64))
    return lo, hi


def test_every_weight_moves_to_a_neighbour_and_the_kernel_sums_survive():
    blob = weights.synthetic(6)
    w = weights.tensor(blob, "ref.res3.1.w").astype(np.float32)          # (32, 32, 3, 3)
    q = api.round_kernels_f16(w)
    assert q.shape == w.shape and np.array_equal(q, q.astype(np.float16).astype(np.float32))      # fp16 numbers
    lo, hi = _neighbours(w.astype(np.float64))
    assert np.all((q == lo) | (q == hi))                                   # never further than the enclosing pair
    rne = w.astype(np.float16).astype(np.float64)
    sum_sp = np.abs((q.astype(np.float64) - w).sum(axis=(2, 3)))
    sum_rne = np.abs((rne - w).sum(axis=(2, 3)))
    ulp = np.spacing(np.abs(w).max(axis=(2, 3)).astype(np.float16)).astype(np.float64)
    assert np.all(sum_sp <= sum_rne + 1e-12)
    assert sum_sp.mean() < 0.15 * sum_rne.mean()                            # the coherent part shrinks > 6x on average ...
    assert np.all(sum_sp < 0.5 * ulp) and sum_sp.mean() < 0.05 * ulp.mean()
    assert np.abs(q - w).max() <= np.abs(hi - lo).max()                     # ... for tap errors of at most one ulp


def test_exact_weights_are_left_alone_and_signs_zeros_are_handled():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((64, 3, 3)).astype(np.float16).astype(np.float32)
    assert np.array_equal(api.round_kernels_f16(w), w)
    z = np.zeros((2, 3, 3), np.float32)
    z[1, 0, 0] = -1e-9                    # below half of the smallest subnormal: 0 or -2^-24
    z[1, 1, 1] = 3e-8
    q = api.round_kernels_f16(z)
    assert np.array_equal(q[0], z[0]) and np.all(np.abs(q[1] - z[1]) <= 2.0 ** -24)
