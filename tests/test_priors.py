"""Log-prior parity (row a1 of SURVEY section 8: generate_model = prior -> model -> likelihood).
CPU: the product's host priors (csrc/priors_impl.h, long double) against the oracle restatement and against analytic
values; parity unpinned by the reference (its tests never evaluate a prior)."""
import numpy as np
import pytest


def test_primitive_priors_known_answers(oracle):
    L = oracle.lib
    assert float(L.orc_logP_uniform(2.0, 6.0, 3.0)) == pytest.approx(-np.log(4.0), rel=1e-15)
    assert float(L.orc_logP_uniform(2.0, 6.0, 7.0)) == -np.inf
    assert float(L.orc_logP_uniform_abs(2.0, 6.0, -3.0)) == pytest.approx(-np.log(4.0), rel=1e-15)
    assert float(L.orc_logP_gaussian(1.0, 0.5, 1.7)) == pytest.approx(-np.log(np.sqrt(2 * np.pi) * 0.5) - 0.5 * (0.7 / 0.5) ** 2, rel=1e-14)
    # Jeffreys: (1/(h+hmin)) / ln((hmax+hmin)/hmin) on 0 < h < hmax
    assert float(L.orc_logP_jeffrey(1.0, 1e4, 12.0)) == pytest.approx(np.log((1 / 13.0) / np.log(10001.0)), rel=1e-14)
    assert float(L.orc_logP_jeffrey(1.0, 1e4, -1.0)) == -np.inf and float(L.orc_logP_jeffrey(1.0, 1e4, 2e4)) == -np.inf
    # GU: flat on [a,b], Gaussian tail below a; normalisation ln(|b-a| + sqrt(2 pi)/2 sigma)
    C = np.log(45.0 + 0.5 * np.sqrt(2 * np.pi) * 2.0)
    assert float(L.orc_logP_gaussian_uniform(0.0, 45.0, 2.0, 10.0)) == pytest.approx(-C, rel=1e-14)
    assert float(L.orc_logP_gaussian_uniform(0.0, 45.0, 2.0, -3.0)) == pytest.approx(-0.5 * 1.5 ** 2 - C, rel=1e-14)
    assert float(L.orc_logP_gaussian_uniform(0.0, 45.0, 2.0, 50.0)) == -np.inf


def test_host_priors_match_oracle(pkg, oracle, synth):
    from tamcmc_c_amd import sampler
    rng = np.random.default_rng(4)
    for star in (synth.make_c3_star(nx=2000, step=1.0), synth.make_c2_star(nx=1000)):
        idx = star.index_to_relax
        v0, st = sampler.log_prior(star)
        assert st == 0 and np.isfinite(v0)
        assert v0 == pytest.approx(oracle.call_prior(star), rel=1e-14)
        n_inf = 0
        for trial in range(200):
            p = star.params.copy()
            p[idx] *= 1.0 + 0.02 * rng.standard_normal(idx.size)
            if trial % 7 == 0:
                p[idx[rng.integers(idx.size)]] *= -1.0     # push something out of its support
            a, st = sampler.log_prior(star, p)
            b = oracle.call_prior(star, p)
            assert st == 0
            if np.isinf(b):
                assert a == b
                n_inf += 1
            else:
                assert a == pytest.approx(b, rel=1e-13, abs=1e-12)
        assert n_inf > 5


def test_smoothness_and_d02_terms(pkg, synth):
    """priors_MS_Global extras: second-difference smoothness of each degree's frequency list ~ N(0, scoef) and
    d02 ~ GU(0, Dnu/3, 0.015 Dnu) (priors_calc.cpp:281-313)."""
    from tamcmc_c_amd import sampler
    star = synth.make_c3_star(nx=2000, step=1.0)
    base, _ = sampler.log_prior(star)
    off = sampler.log_prior(_with_extra(star, 0, 0.0))[0]          # smoothness off
    nmax = int(star.plength[0])
    f = star.params[nmax + 3:nmax + 3 + 4 * nmax].reshape(4, nmax)
    d2 = np.zeros_like(f)
    d2[:, 1:-1] = f[:, 2:] - 2 * f[:, 1:-1] + f[:, :-2]
    d2[:, 0] = d2[:, 1]
    d2[:, -1] = d2[:, -2]
    want = np.sum(-np.log(np.sqrt(2 * np.pi) * 2.0) - 0.5 * (d2 / 2.0) ** 2)
    assert base - off == pytest.approx(want, rel=1e-12)


def _with_extra(star, i, v):
    import copy
    s = copy.copy(star)
    s.extra_priors = star.extra_priors.copy()
    s.extra_priors[i] = v
    return s
