"""BASELINE config C1: the reference's own sample input test/inputs/TF_3443483_local-v3 (.data slice 1: 973 bins of a real
Kepler red-giant spectrum, committed as tests/golden/TF_3443483_local-v3_slice1.data; model_MS_local_basic, the two modes
of that slice as listed in the .model file: l=0 at 99.1056 muHz, l=2 at 97.7716 muHz), 4 tempered chains.
The first test assembles the parameter vector by hand from the .model's eigen/noise tables; test_c1_from_the_shipped_files goes
through the `.model`/`.data` loaders (include/tamcmc_io.h)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_data_ascii(path):
    """`.data` format (config.cpp:907-1060): '#' header, '!' labels, '*' units, then whitespace-separated columns."""
    rows = [ln.split() for ln in open(path) if ln.strip() and ln.lstrip()[0] not in "#!*"]
    a = np.array(rows, dtype=np.float64)
    return a[:, 0].copy(), a[:, 1].copy()


def c1_star(synth, oracle, x):
    # eigen table of the .model: l / nu / Gamma / H ; noise table A/B/p x3 + N0 -> local white level at the slice centre
    H0, G0, f0 = 1199.06221, 0.30864, 99.10560
    H2, G2, f2 = 635.50294, 0.30646, 97.77160
    nu = 0.5 * (x[0] + x[-1])
    N0 = 664.13440 / (1 + (1e-3 * 32.186165 * nu) ** 4.0) + 355.78922 / (1 + (1e-3 * 13.739317 * nu) ** 2.5) + 5.1845856
    a1, inc = 0.4, 45.0
    eta0 = oracle.lib.orc_eta0_from_dnu(10.6795)
    params = np.array([H0, H2, f0, f2, 0.0, eta0, 0.0, np.sqrt(a1) * np.cos(np.radians(inc)), np.sqrt(a1) * np.sin(np.radians(inc)), 0.0,
                       G0, G2, N0, 0.0, 10000.0, 0.0])
    plength = np.array([2, 0, 1, 0, 1, 0, 6, 2, 1, 1, 2], dtype=np.int32)
    assert params.size == plength.sum() == 16
    names = ["Height_l", "Height_l", "Frequency_l", "Frequency_l", "Splitting_a1", "Asphericity_eta", "Splitting_a3",
             "sqrt(splitting_a1).cosi", "sqrt(splitting_a1).sini", "Lorentzian_asymetry", "Width_l", "Width_l", "White_Noise_N0",
             "Inclination", "Truncation_parameter", "do_amp"]
    relax = np.zeros(16, dtype=np.int32)
    relax[[0, 1, 2, 3, 7, 8, 10, 11, 12]] = 1
    P = synth
    rules = {"Height_l": (P.P_JEFF, lambda v: (1.0, 1.0e4)), "Frequency_l": (P.P_UNIFORM, lambda v: (v - 0.6, v + 0.6)),
             "sqrt(splitting_a1).cosi": (P.P_UNIFORM, lambda v: (0.0, 1.3)), "sqrt(splitting_a1).sini": (P.P_UNIFORM, lambda v: (0.0, 1.3)),
             "Width_l": (P.P_JEFF, lambda v: (0.02, 5.0)), "White_Noise_N0": (P.P_UNIFORM, lambda v: (0.0, 1000.0))}
    pr, sw = P._prior_tables(names, params, relax, rules)
    extra = np.array([0.0, 0.0, 0.2, 0.0, 0, 0, 0, 0, 0, 0])
    return P.Star(P.MODEL_LOCAL, params, plength, x, relax, pr, sw, names, prior_class=3, extra_priors=extra)


def test_c1_real_spectrum_parity_and_sampling(pkg, oracle, synth):
    x, y = load_data_ascii(os.path.join(GOLD, "TF_3443483_local-v3_slice1.data"))
    assert x.size == 973 and 94.30 <= x[0] and x[-1] <= 102.20
    assert np.allclose(np.diff(x), x[1] - x[0], rtol=0, atol=2e-8)   # the regular grid set_imin_imax assumes
    star = c1_star(synth, oracle, x)
    star.y = y
    B = 4
    T = 3.5 ** np.arange(B)                       # config_default.cfg: lambda_temp = 3.5
    rng = np.random.default_rng(1)
    P = np.tile(star.params, (B, 1))
    P[1:, star.index_to_relax] *= 1 + 0.01 * rng.standard_normal((B - 1, star.nvars))
    ref, m_o, st_o = oracle.loglike_batch(star.model_id, P, star.plength, x, y, 1.0, T, want_model=True)
    assert (st_o == 0).all()
    for prec, tol_m, tol_l in ((pkg.PRECISION_STRICT, 0.0, 1e-12), (pkg.PRECISION_FAST, 1e-12, 1e-11)):
        c = pkg.HipContext(0, precision=prec)
        c.set_spectrum(x, y)
        logL, model, status = c.loglike_params_batch(star.model_id, P, star.plength, T, 1.0, want_model=True)
        assert (status == 0).all()
        assert np.max(np.abs(model - m_o) / m_o) <= tol_m
        for b in range(B):
            truth = oracle.chi22p_ld(y, m_o[b], 1) / T[b]
            assert abs(logL[b] - truth) <= tol_l * abs(truth)
        c.close()
    flat = -(y / y.mean() + np.log(y.mean())).sum()   # white-noise-only model: the fit must end up above it
    # 4 tempered chains, both engines: a short run stays finite and moves
    for eng in ("host", "device"):
        c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
        c.set_spectrum(x, y)
        s = pkg.Sampler(c, star, nchains=4, lambda_temp=3.5, engine=eng, seed=5, Nt_learn=(100, 600), periods_learn=(1,), c0=5.0)
        s.run(600, record=False)
        smp, stt = s.run(500, stats=True)
        assert np.all(np.isfinite(stt)) and np.all(np.isfinite(smp))
        acc = np.mean(np.any(smp[1:, 0] != smp[:-1, 0], axis=1))
        assert 0.03 < acc < 0.9, (eng, acc)
        # the catalogue values of the .model are initial guesses: sampling must improve on them and beat the flat model
        assert stt[:, 0, 0].mean() > ref[0] + 10 and stt[:, 0, 0].mean() > flat + 10, (eng, stt[:, 0, 0].mean(), ref[0], flat)
        post = smp[:, 0, :]
        fidx = [i for i, k in enumerate(star.index_to_relax) if star.names[k] == "Frequency_l"]
        dev = np.abs(post[:, fidx].mean(0) - star.params[star.index_to_relax][fidx])
        assert dev[0] < 0.1 and np.all(dev <= 0.6)   # the strong l=0 peak is pinned; the weak l=2 one stays inside its prior box
        s.close(); c.close()


def test_device_priors_match_host(pkg, oracle, synth):
    """log-prior evaluated by the device kernels (double, terms spread over lanes) vs the host (long double, reference order)."""
    from tamcmc_c_amd import sampler
    star = synth.make_c3_star(nx=20000, step=0.1)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    star.set_spectrum_from_model(m0, 3)
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    c.set_spectrum(star.x, star.y)
    rng = np.random.default_rng(8)
    idx = star.index_to_relax
    h = 1e-7 * np.maximum(np.abs(star.params[idx]), 1e-3)
    P = np.tile(star.params, (6, 1))
    P[1:, idx] *= 1 + 0.01 * rng.standard_normal((5, idx.size))
    P[5, idx[3]] = -5.0     # a negative height: Jeffreys prior -> -inf
    _, pr0, _ = c.fd_gradient_posterior(star, P, h, np.ones(6), 1.0)
    for b in range(6):
        want, st = sampler.log_prior(star, P[b])
        if np.isinf(want):
            assert pr0[b] == want
        else:
            assert pr0[b] == pytest.approx(want, rel=1e-13, abs=1e-11)
    c.close()


def test_c1_from_the_shipped_files(pkg, oracle):
    """BASELINE config C1 end to end from the reference's own files: `.model` (slice 0) + `.data` -> Input_Data
    (include/tamcmc_io.h) -> logL parity with the oracle at the file's starting point -> 4 tempered chains on the device."""
    from tamcmc_c_amd import inputs
    star, inp = inputs.load_local_star(os.path.join(GOLD, "TF_3443483_local-v3.model"),
                                       os.path.join(GOLD, "TF_3443483_local-v3_slice1.data"), 0)
    assert star.x.size == 973 and list(inp.plength) == [2, 0, 1, 0, 1, 0, 6, 2, 1, 1, 2]
    T = 3.5 ** np.arange(4)
    P = np.tile(star.params, (4, 1))
    ref, _, st_o = oracle.loglike_batch(star.model_id, P, star.plength, star.x, star.y, 1.0, T)
    assert (st_o == 0).all()
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    got, _, st = ctx.loglike_params_batch(star.model_id, P, star.plength, T)
    assert (st == 0).all() and np.allclose(got, ref, rtol=1e-11, atol=0)
    for eng in ("host", "device"):
        s = pkg.Sampler(ctx, star, nchains=4, lambda_temp=3.5, seed=5, engine=eng, Nt_learn=(20, 400), periods_learn=(1,))
        st0 = s.state()
        assert np.isfinite(st0["logPrior"]).all() and np.allclose(st0["logL"], ref, rtol=1e-11)
        smp, stat = s.run(600, stats=True)
        st1 = s.state()
        assert st1["iteration"] == 600 and np.isfinite(stat).all()
        assert stat[-200:, 0, 2].mean() > st0["logPost"][0] - 5.0        # the cold chain stays in the posterior's bulk
        lo, hi = star.priors[0, star.index_to_relax], star.priors[1, star.index_to_relax]
        un = [k for k, i in enumerate(star.index_to_relax) if inp.prior_names[i] == "Uniform"]
        je = [k for k, i in enumerate(star.index_to_relax) if inp.prior_names[i] == "Jeffreys"]
        assert np.all(smp[:, 0][:, un] >= lo[un]) and np.all(smp[:, 0][:, un] <= hi[un])   # hard-bounded priors are respected
        assert np.all(smp[:, 0][:, je] >= 0) and np.all(smp[:, 0][:, je] <= hi[je])       # (modified Jeffreys: support [0, hmax])
        s.close()
    ctx.close()


def test_global_aj_model_file_on_the_device(pkg, oracle):
    """The reference's Sun sample `.model` (model_MS_Global_aj_HarveyLike, 90 parameters, 71 free) through the loader; the
    reference ships no `.data` for it, so the spectrum is the model of the file's starting point times Exp(1) noise."""
    from tamcmc_c_amd import inputs
    step = 0.0317
    x = 2330.0 + step * np.arange(int((3660.0 - 2330.0) / step))
    inp = inputs.GlobalInputs(os.path.join(GOLD, "Sun_19992002_incfix_fast_Priorevalrange.model"), step)
    star = inputs.star_from_inputs(inp, x)
    st_m, m0 = oracle.call_model(star.model_id, star.params, star.plength, x)
    assert st_m == 0 and np.all(m0 > 0)
    y = star.set_spectrum_from_model(m0, 3)
    T = 1.5 ** np.arange(6)
    P = np.tile(star.params, (6, 1))
    ref, _, st_o = oracle.loglike_batch(star.model_id, P, star.plength, x, y, 1.0, T)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(x, y)
    got, _, st = ctx.loglike_params_batch(star.model_id, P, star.plength, T)
    assert (st == 0).all() and (st_o == 0).all() and np.allclose(got, ref, rtol=1e-11, atol=0)
    s = pkg.Sampler(ctx, star, nchains=6, lambda_temp=1.5, seed=9, engine="device", Nt_learn=(10, 300), periods_learn=(1,))
    st0 = s.state()
    assert np.isfinite(st0["logPrior"]).all() and np.allclose(st0["logL"], ref, rtol=1e-11)
    smp, stat = s.run(400, stats=True)
    assert np.isfinite(stat).all() and s.state()["iteration"] == 400
    assert (smp[:, 0] != smp[0, 0]).any()            # the cold chain moves
    s.close(); ctx.close()
