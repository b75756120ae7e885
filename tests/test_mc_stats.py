"""The Monte-Carlo-error helper of the posterior checks (tests/mc_stats.py) on chains with a known autocorrelation time."""
import numpy as np

import mc_stats


def _ar1(phi, n, seed):
    rng = np.random.default_rng(seed)
    e = rng.standard_normal(n)
    x = np.zeros(n)
    for i in range(1, n):
        x[i] = phi * x[i - 1] + e[i]
    return x


def test_integrated_autocorrelation_time_of_ar1():
    for phi in (0.0, 0.5, 0.9):
        tau = mc_stats.tau_int(_ar1(phi, 100000, 3))
        assert abs(tau - (1 + phi) / (1 - phi)) < 0.08 * (1 + phi) / (1 - phi), (phi, tau)


def test_two_chains_of_one_law_agree_and_a_shifted_one_does_not():
    a = np.stack([_ar1(0.8, 40000, 1), _ar1(0.3, 40000, 2)], axis=1)
    b = np.stack([_ar1(0.8, 40000, 5), _ar1(0.3, 40000, 6)], axis=1)
    zm, zv, ea, eb = mc_stats.compare_chains(a, b)
    assert np.all(np.abs(zm) < 4) and np.all(np.abs(zv) < 4.5)
    assert 3000 < ea[0] < 6000 and 15000 < ea[1] < 28000          # n / tau = 40000 / 9, 40000 / 1.86
    c = b.copy()
    c[:, 0] += 0.1 * b[:, 0].std()                                  # a tenth of a sigma: far outside the Monte-Carlo error
    zm2, _, _, _ = mc_stats.compare_chains(a, c)
    assert abs(zm2[0]) > 4
    d = b.copy()
    d[:, 1] *= 1.1
    _, zv2, _, _ = mc_stats.compare_chains(a, d)
    assert abs(zv2[1]) > 4.5
