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


def test_asymptotic_priors_match_oracle(pkg, oracle, synth):
    """Prior class io_asymptotic (priors_asymptotic, priors_calc.cpp:319-512) for the red-giant model: product host code vs oracle,
    incl. the hard constraints and the smoothness terms (which the reference switches on the prior id of the first noise
    parameter -- kept, see csrc/priors_impl.h)."""
    from tamcmc_c_amd import sampler
    star = synth.make_c5_star(nx=2000, nmax=7)
    o = np.cumsum([0] + list(star.plength))
    rng = np.random.default_rng(3)
    base = star.params.copy()
    cases = [base]
    for _ in range(6):
        p = base.copy()
        p[star.index_to_relax] *= 1 + 1e-3 * rng.standard_normal(star.nvars)
        cases.append(p)
    bad = {"negative visibility": (o[1], -0.1), "a3/a1 over the limit": (o[6] + 4, 0.5), "negative Wfactor": (o[3] + 6, -0.2),
           "negative Hfactor": (o[3] + 7, -0.2), "negative width-law parameter": (o[7] + 2, -1.0), "outside a uniform prior": (o[3] + 3, 1.5)}
    for k, (idx, val) in bad.items():
        p = base.copy()
        p[idx] = val
        cases.append(p)
    got = np.array([sampler.log_prior(star, p)[0] for p in cases])
    ref = np.array([oracle.call_prior(star, p) for p in cases])
    assert np.all(np.isfinite(ref[:7])) and np.all(ref[7:] == -np.inf)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref)) and np.allclose(got[:7], ref[:7], rtol=1e-14)
    # smoothness: active because the first noise parameter carries a Uniform prior (id 1); a kink in the l=0 ladder is penalised
    assert star.priors_switch[o[8]] == 1
    p = base.copy()
    p[o[2] + 3] += 0.4
    d = sampler.log_prior(star, p)[0] - got[0]
    sd = lambda y: np.array([y[2] - 2 * y[1] + y[0]] + list(y[2:] - 2 * y[1:-1] + y[:-2]) + [y[-1] - 2 * y[-2] + y[-3]])
    want = -0.5 * (np.sum(sd(p[o[2]:o[3]]) ** 2) - np.sum(sd(base[o[2]:o[3]]) ** 2)) / 2.0 ** 2
    assert np.isclose(d, want, rtol=1e-9)
