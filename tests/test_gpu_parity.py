"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Tolerances (fp64):
  STRICT model row : bit-exact when the Harvey pow() terms are inactive; <= 1e-15 relative otherwise (device pow vs libm)
  FAST   model row : <= 1e-12 relative per bin
  logL             : <= 1e-12 (STRICT) / 1e-11 (FAST) relative against the 80-bit-accumulated oracle sum;
                     the reference's own acceptance bound ||dM||_2 <= 1e-8 (test_build_l_mode.cpp:104,134) is the envelope.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs(pkg):
    c = {"strict": pkg.HipContext(0, precision=pkg.PRECISION_STRICT), "fast": pkg.HipContext(0, precision=pkg.PRECISION_FAST),
         "fast_direct": pkg.HipContext(0, precision=pkg.PRECISION_FAST_DIRECT)}
    yield c
    for v in c.values():
        v.close()


def _spectrum(oracle, star, seed=1):
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    return star.set_spectrum_from_model(m0, seed)


def _perturbed(star, B, rng, scale=0.01):
    P = np.tile(star.params, (B, 1))
    idx = star.index_to_relax
    P[1:, idx] *= 1.0 + scale * rng.standard_normal((B - 1, idx.size))
    return P


GEOMS = [(256, 1), (256, 2), (256, 4), (64, 4), (64, 8), (64, 16)]   # (workgroup size, bins per thread)


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("geom", GEOMS)
def test_reference_recipe_strict_bit_exact(pkg, oracle, synth, ctxs, seed, geom):
    wg, K = geom
    rng = np.random.default_rng(200 + seed)
    lmax = int(rng.integers(2, 4))
    p, pl = synth.make_params_aj_model(rng, lmax=lmax, nfreqs=5, dnu=rng.uniform(129, 130), epsilon=rng.uniform(0, 0.05),
                                       d0l=rng.uniform(-2.6, 0))
    nx = int(np.ceil((5 * 130 + 250) / synth.KEPLER_4YR_RESOL))
    x = np.linspace(0.0, 5 * 130 + 250.0, nx)
    st, m_o = oracle.call_model(23, p, pl, x)
    y = m_o * np.random.default_rng(seed).exponential(1.0, nx)
    c = ctxs["strict"]
    c.set_option(pkg.OPT_WORKGROUP, wg)
    c.set_option(pkg.OPT_BINS_PER_THREAD, K)
    c.set_spectrum(x, y)
    logL, model, status = c.loglike_params_batch(23, p, pl, want_model=True)
    assert status[0] == 0
    assert np.array_equal(model[0], m_o), float(np.max(np.abs(model[0] - m_o)))
    ref = oracle.chi22p_ld(y, m_o, 1)
    assert abs(logL[0] - ref) <= 1e-12 * abs(ref)
    assert np.linalg.norm(model[0] - m_o) <= 1e-8
    for name in ("fast", "fast_direct"):
        f = ctxs[name]
        f.set_option(pkg.OPT_WORKGROUP, wg)
        f.set_option(pkg.OPT_BINS_PER_THREAD, K)
        f.set_spectrum(x, y)
        logLf, modelf, _ = f.loglike_params_batch(23, p, pl, want_model=True)
        assert np.max(np.abs(modelf[0] - m_o) / m_o) <= 1e-12, name
        assert abs(logLf[0] - ref) <= 1e-11 * abs(ref), name
        assert np.linalg.norm(modelf[0] - m_o) <= 1e-12 * np.linalg.norm(m_o), name


def test_c2_local_batch_tempered(pkg, oracle, synth, ctxs):
    star = synth.make_c2_star()
    y = _spectrum(oracle, star)
    rng = np.random.default_rng(9)
    B = 10
    P = _perturbed(star, B, rng)
    T = 1.7 ** np.arange(B)
    ref, m_o, st_o = oracle.loglike_batch(star.model_id, P, star.plength, star.x, y, 1.0, T, want_model=True)
    for name, tol_m, tol_l in (("strict", 0.0, 1e-12), ("fast", 1e-12, 1e-11), ("fast_direct", 1e-12, 1e-11)):
        c = ctxs[name]
        c.set_option(pkg.OPT_WORKGROUP, 256)
        c.set_option(pkg.OPT_BINS_PER_THREAD, 2)
        c.set_spectrum(star.x, y)
        logL, model, status = c.loglike_params_batch(star.model_id, P, star.plength, T, 1.0, want_model=True)
        assert (status == 0).all() and (st_o == 0).all()
        assert np.max(np.abs(model - m_o) / m_o) <= tol_m
        for b in range(B):
            truth = oracle.chi22p_ld(y, m_o[b], 1) / T[b]
            assert abs(logL[b] - truth) <= tol_l * abs(truth)
            assert abs(logL[b] - ref[b]) <= 1e-10 * abs(ref[b])


def test_c3_like_global_with_harvey_and_asymmetry(pkg, oracle, synth, ctxs):
    star = synth.make_c3_star(nx=30000, step=2000.0 / 30000)
    o = star.plength[0] + star.plength[1] + star.plength[2:6].sum()
    star.params[o + 13] = 25.0   # asymmetry
    star.params[o + 12] = 1.0    # eta0 on
    y = _spectrum(oracle, star)
    rng = np.random.default_rng(10)
    B = 5
    P = _perturbed(star, B, rng, 0.003)
    T = 1.4 ** np.arange(B)
    ref, m_o, st_o = oracle.loglike_batch(star.model_id, P, star.plength, star.x, y, 1.0, T, want_model=True)
    assert (st_o == 0).all()
    for name, tol_m, tol_l in (("strict", 1e-15, 1e-12), ("fast", 1e-12, 1e-11), ("fast_direct", 1e-12, 1e-11)):
        c = ctxs[name]
        c.set_spectrum(star.x, y)
        logL, model, status = c.loglike_params_batch(star.model_id, P, star.plength, T, 1.0, want_model=True)
        assert (status == 0).all()
        assert np.max(np.abs(model - m_o) / m_o) <= tol_m, name
        for b in range(B):
            truth = oracle.chi22p_ld(y, m_o[b], 1) / T[b]
            assert abs(logL[b] - truth) <= tol_l * abs(truth)


def test_classic_model_and_p2(pkg, oracle, synth, ctxs):
    rng = np.random.default_rng(11)
    p, pl = synth.make_params_aj_model(rng, lmax=3, nfreqs=6, asym=-30.0, n_first=12)
    pc, plc = synth.aj_to_classic(p, pl)
    x = synth.grid(50001, 1400.0, 0.02)   # ragged: not a multiple of any tile size
    _, m_o = oracle.call_model(3, pc, plc, x)
    y = m_o * np.random.default_rng(2).exponential(1.0, x.size)
    c = ctxs["strict"]
    c.set_spectrum(x, y)
    logL, model, status = c.loglike_params_batch(3, pc, plc, None, 2.0, want_model=True)
    assert np.array_equal(model[0], m_o)
    truth = oracle.chi22p_ld(y, m_o, 2)
    assert abs(logL[0] - truth) <= 1e-12 * abs(truth)


def test_table_level_entry_and_empty_table(pkg, oracle, synth, ctxs):
    """tamcmc_hip_loglike_batch with explicit tables; an evaluation with zero multiplets = background only."""
    star = synth.make_c2_star(nx=3000)
    y = _spectrum(oracle, star)
    st, mults, noise, nh = pkg.build_mode_table(star.model_id, star.params, star.plength, star.x)
    c = ctxs["strict"]
    c.set_spectrum(star.x, y)
    offsets = np.array([0, len(mults), len(mults)], dtype=np.int32)
    noise2 = np.stack([noise, noise])
    logL, model = c.loglike_batch(mults, offsets, noise2, [nh, nh], [noise.size, noise.size], None, 1.0, want_model=True)
    _, m_o = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    assert np.array_equal(model[0], m_o)
    assert np.array_equal(model[1], np.full(star.x.size, noise[-1]))
    truth = -(y / noise[-1] + np.log(noise[-1])).sum()
    assert logL[1] == pytest.approx(truth, rel=1e-13)
    with pytest.raises(pkg.TamcmcError):
        bad = mults.copy()
        bad["i1"][0] = star.x.size + 5   # window beyond the grid must be refused on the host, never launched
        c.loglike_batch(bad, offsets, noise2, [nh, nh], [noise.size, noise.size])


def test_failed_window_gives_nan_and_status(pkg, oracle, synth, ctxs):
    star = synth.make_c2_star(nx=3000)
    y = _spectrum(oracle, star)
    P = np.tile(star.params, (3, 1))
    P[1, 18] = np.nan
    c = ctxs["fast"]
    c.set_spectrum(star.x, y)
    logL, _, status = c.loglike_params_batch(star.model_id, P, star.plength)
    assert status.tolist() == [0, pkg.ERR_NAN_WINDOW, 0]
    assert np.isnan(logL[1]) and np.isfinite(logL[0]) and logL[0] == logL[2]


def test_fd_gradient_matches_oracle(pkg, oracle, synth, ctxs):
    """The finite-difference gradient has no reference counterpart (MALA::D_MALA is a stub, MALA.cpp:321-328);
    its oracle is the same forward difference of the CPU log-likelihood."""
    star = synth.make_c2_star(nx=4000)
    y = _spectrum(oracle, star)
    idx = star.index_to_relax
    h = 1e-6 * np.maximum(np.abs(star.params[idx]), 1.0)
    _, l0_o, g_o = oracle.fd_gradient(star.model_id, star.params, star.plength, idx, h, star.x, y, 1.0, 1.7)
    for name, tol in (("strict", 1e-6), ("fast", 1e-5)):
        c = ctxs[name]
        c.set_spectrum(star.x, y)
        l0, g = c.fd_gradient(star.model_id, star.params, star.plength, idx, h, [1.7], 1.0)
        assert abs(l0[0] - l0_o) <= 1e-11 * abs(l0_o)
        scale = np.max(np.abs(g_o))
        assert np.max(np.abs(g[0] - g_o)) <= tol * scale, name


def test_full_size_c3_properties(pkg, oracle, synth, ctxs):
    """BASELINE full size (1e5 bins x 111 params x 20 chains): oracle on 2 chains, plus size-independent properties:
    identical chains give identical logL (determinism), tempering scales exactly, STRICT vs FAST agree to 1e-11,
    zero-height modes reduce the model to the background."""
    star = synth.make_c3_star()
    y = _spectrum(oracle, star)
    rng = np.random.default_rng(12)
    B = 20
    P = _perturbed(star, B, rng, 0.002)
    P[7] = P[3]
    T = 1.35 ** np.arange(B)
    ref, _, _ = oracle.loglike_batch(star.model_id, P[:2], star.plength, star.x, y, 1.0, T[:2])
    out = {}
    for name in ("strict", "fast", "fast_direct"):
        c = ctxs[name]
        c.set_option(pkg.OPT_WORKGROUP, 64 if name == "fast" else 256)   # the library defaults of each mode
        c.set_option(pkg.OPT_BINS_PER_THREAD, 8 if name == "fast" else 4)
        c.set_spectrum(star.x, y)
        logL, _, status = c.loglike_params_batch(star.model_id, P, star.plength, T, 1.0)
        logL1, _, _ = c.loglike_params_batch(star.model_id, P, star.plength, None, 1.0)
        assert (status == 0).all()
        assert np.allclose(logL[:2], ref, rtol=1e-11, atol=0)
        assert logL1[7] == logL1[3]
        assert np.allclose(logL, logL1 / T, rtol=4e-16, atol=0)
        again, _, _ = c.loglike_params_batch(star.model_id, P, star.plength, T, 1.0)
        assert np.array_equal(again, logL)
        out[name] = logL
    assert np.max(np.abs(out["fast"] - out["strict"]) / np.abs(out["strict"])) <= 1e-11
    assert np.max(np.abs(out["fast_direct"] - out["strict"]) / np.abs(out["strict"])) <= 1e-11
    P0 = star.params.copy()
    P0[:14] = 0.0
    c = ctxs["strict"]
    _, model, _ = c.loglike_params_batch(star.model_id, P0, star.plength, want_model=True)
    nz = np.abs(star.params[star.plength[:8].sum():star.plength[:9].sum()])
    bg = nz[0] / (1 + (1e-3 * nz[1] * star.x) ** nz[2]) + nz[3] / (1 + (1e-3 * nz[4] * star.x) ** nz[5]) + nz[6]
    assert np.max(np.abs(model[0] - bg) / bg) < 1e-14


def test_fd_gradient_posterior_device_batch(pkg, oracle, synth, ctxs):
    """tamcmc_hip_fd_gradient_posterior: tables AND priors of the Nvars+1 perturbed vectors are built on the device.
    Likelihood part against the oracle's forward differences; prior part against central differences of the host prior
    (host prior == reference restatement in long double)."""
    star = synth.make_c3_star(nx=20000, step=0.1)
    y = _spectrum(oracle, star)
    idx = star.index_to_relax
    h = 1e-7 * np.maximum(np.abs(star.params[idx]), 1e-3)
    T = 1.3
    _, l0_o, g_o = oracle.fd_gradient(star.model_id, star.params, star.plength, idx, h, star.x, y, 1.0, T)
    c = ctxs["fast"]
    c.set_spectrum(star.x, y)
    l0, g_like = c.fd_gradient(star.model_id, star.params, star.plength, idx, h, [T], 1.0)
    assert abs(l0[0] - l0_o) <= 1e-11 * abs(l0_o)
    scale = np.max(np.abs(g_o))
    # forward differences: logL is a sum of Nx O(1) terms, two implementations differ by ~1e-15 Nx -> gradient error ~ that / h_k
    assert np.all(np.abs(g_like[0] - g_o) <= 5e-15 * star.x.size / h + 1e-6 * scale)
    l0p, pr0, g_post = c.fd_gradient_posterior(star, star.params, h, [T], 1.0)
    assert l0p[0] == l0[0] and np.isfinite(pr0[0])
    g_prior = g_post[0] - g_like[0]
    # smoothness + Jeffreys + Gaussian priors have O(1) gradients; they must be finite and mostly non-zero for the frequencies
    assert np.all(np.isfinite(g_prior))
    fsel = [i for i, k in enumerate(idx) if star.names[k] == "Frequency_l"]
    assert np.count_nonzero(np.abs(g_prior[fsel]) > 1e-6) > len(fsel) // 2
    # a vector at the edge of a uniform prior: forward point outside the support -> backward difference, finite gradient
    P = star.params.copy()
    k_inc = [i for i, k in enumerate(idx) if star.names[k] == "Inclination"][0]
    P[idx[k_inc]] = 90.0 - 0.25 * h[k_inc]
    _, _, g_edge = c.fd_gradient_posterior(star, P, h, [T], 1.0)
    assert np.all(np.isfinite(g_edge))


def test_windowed_fd_matches_full_fd(pkg, oracle, synth, ctxs):
    """Windowed finite differences (delta tables: only the multiplets a perturbation changes, on their windows, against the
    stored base model row) against the brute-force batch (every perturbed model rebuilt on all bins) and the oracle."""
    star = synth.make_c3_star(nx=40000, step=0.05)
    o = star.plength[0] + star.plength[1] + star.plength[2:6].sum()
    star.params[o + 13] = 15.0   # asymmetry on: exercises the asymmetric far field with negative heights
    y = _spectrum(oracle, star)
    idx = star.index_to_relax
    h = 1e-6 * np.maximum(np.abs(star.params[idx]), 1e-2)
    T = np.array([1.0, 1.3, 2.2])
    rng = np.random.default_rng(5)
    P = np.tile(star.params, (3, 1))
    P[1:, idx] *= 1 + 0.003 * rng.standard_normal((2, idx.size))
    for name in ("fast", "fast_direct"):
        c = ctxs[name]
        c.set_option(pkg.OPT_WORKGROUP, 64 if name == "fast" else 256)
        c.set_option(pkg.OPT_BINS_PER_THREAD, 8 if name == "fast" else 4)
        c.set_spectrum(star.x, y)
        c.set_option(pkg.OPT_FD_WINDOWED, 0)
        l0_f, pr_f, g_f = c.fd_gradient_posterior(star, P, h, T, 1.0)
        c.set_option(pkg.OPT_FD_WINDOWED, 1)
        l0_w, pr_w, g_w = c.fd_gradient_posterior(star, P, h, T, 1.0)
        assert np.allclose(l0_w, l0_f, rtol=1e-12) and np.array_equal(pr_w, pr_f)
        scale = np.max(np.abs(g_f), axis=1, keepdims=True)
        # the brute-force difference of two ~Nx-term sums carries ~1e-15 Nx / h of cancellation noise; the windowed one does not
        tol = 5e-15 * star.x.size / h[None, :] + 1e-6 * scale
        assert np.all(np.abs(g_w - g_f) <= tol), name
    # against the CPU oracle (chain 0, likelihood part only)
    c = ctxs["fast"]
    _, g_like = c.fd_gradient(star.model_id, star.params, star.plength, idx, h, [1.0], 1.0)
    _, l0_o, g_o = oracle.fd_gradient(star.model_id, star.params, star.plength, idx, h, star.x, y, 1.0, 1.0)
    # (the oracle differences two SEQUENTIALLY summed ~Nx-term sums: ~sqrt(Nx) eps |partial sums| of cancellation noise each)
    assert np.all(np.abs(g_like[0] - g_o) <= 5e-14 * star.x.size / h + 1e-6 * np.max(np.abs(g_o)))
    # windowed differences have no cancellation: halving the step changes the gradient only by the O(h) truncation term
    _, g_half = c.fd_gradient(star.model_id, star.params, star.plength, idx, 0.5 * h, [1.0], 1.0)
    # (parameters that move a truncation window -- a1, widths, frequencies -- make logL piecewise discontinuous: a window edge
    #  crossing a bin between h and h/2 shows up as a jump; that is the model's property, build_lorentzian.cpp:645-649)
    okc = np.abs(g_half[0] - g_like[0]) <= 3e-3 * np.max(np.abs(g_o)) + 1e-2 * np.abs(g_like[0])
    assert okc.mean() >= 0.95


def test_full_table_fd_evaluations(pkg, oracle, synth, ctxs):
    """A perturbation that moves most multiplets (splitting coefficients, the asymmetry) is evaluated by the delta launch as the WHOLE
    perturbed table minus the stored base model row (d_flags bit 1, loglike_tile.h) instead of +new / -old row pairs.  Every splitting
    coefficient and the asymmetry as variables, beside parameters that take the pair tables: against the brute-force batch (every
    perturbed model rebuilt on all bins) and the oracle's forward differences."""
    star = synth.make_c3_star(nx=40000, step=0.05)
    o = star.plength[0] + star.plength[1] + star.plength[2:6].sum()
    star.params[o + 13] = 12.0   # asymmetry on
    star.params[o + 1] = 0.02    # a1 slope, so that its neighbours are not at zero either
    y = _spectrum(oracle, star)
    idx = np.concatenate([np.arange(o, o + 12), [o + 13], star.index_to_relax[[0, 15, 30, 60, 92]]]).astype(np.int32)
    h = 1e-6 * np.maximum(np.abs(star.params[idx]), 1e-2)
    T = np.array([1.0, 1.6])
    P = np.tile(star.params, (2, 1))
    P[1, star.index_to_relax] *= 1 + 0.002 * np.random.default_rng(8).standard_normal(star.nvars)
    c = ctxs["fast"]
    c.set_option(pkg.OPT_WORKGROUP, 64)
    c.set_option(pkg.OPT_BINS_PER_THREAD, 8)
    c.set_spectrum(star.x, y)
    c.set_option(pkg.OPT_FD_WINDOWED, 0)
    l0_f, g_f = c.fd_gradient(star.model_id, P, star.plength, idx, h, T, 1.0)
    c.set_option(pkg.OPT_FD_WINDOWED, 1)
    c.set_option(pkg.OPT_TIMING, 1)
    c.reset_kernel_stats()
    l0_w, g_w = c.fd_gradient(star.model_id, P, star.plength, idx, h, T, 1.0)
    n_full = c.fd_full_tables()
    c.set_option(pkg.OPT_TIMING, 0)
    # a1 and a2 (constant and slope) move every l >= 1 multiplet, the asymmetry every multiplet: five full tables per vector;
    # a3, a4 (l >= 2) and a5, a6 (l = 3) change half the rows or fewer and keep their pair tables, like the single-mode parameters
    assert n_full == 5 * P.shape[0], n_full
    assert np.allclose(l0_w, l0_f, rtol=1e-12)
    scale = np.max(np.abs(g_f), axis=1, keepdims=True)
    tol = 5e-15 * star.x.size / h[None, :] + 1e-6 * scale
    assert np.all(np.abs(g_w - g_f) <= tol), np.max(np.abs(g_w - g_f) / tol)
    assert np.all(np.abs(g_w[:, :13]) > 0)   # every splitting coefficient and the asymmetry move the likelihood
    _, l0_o, g_o = oracle.fd_gradient(star.model_id, star.params, star.plength, idx, h, star.x, y, 1.0, 1.0)
    assert np.all(np.abs(g_w[0] - g_o) <= 5e-14 * star.x.size / h + 1e-6 * np.max(np.abs(g_o)))


def test_far_only_tiles_come_from_the_moments(pkg, oracle, synth, ctxs):
    """A perturbed frequency moves two or three multiplets whose windows span many tiles, and is in the far field of most of them: those
    tiles are taken from moments of the base point (k_fd_moments / k_fd_far, kernels.hip) and no bin of theirs is walked.  Frequencies,
    widths and heights as variables: the gradient equals the brute-force batch and the oracle's differences as before, and the bins the
    delta launch walks are a small part of the spectrum (they were the affected ranges, half of it, before the moments)."""
    star = synth.make_c3_star(nx=40000, step=0.05)
    y = _spectrum(oracle, star)
    names = np.array(star.names)[star.index_to_relax]
    idx = star.index_to_relax[np.isin(names, ["Frequency_l", "Width_l0", "Height_l0"])]
    h = 1e-6 * np.maximum(np.abs(star.params[idx]), 1e-2)
    T = np.array([1.0, 1.5, 2.5])
    P = np.tile(star.params, (3, 1))
    P[1:, star.index_to_relax] *= 1 + 0.002 * np.random.default_rng(4).standard_normal((2, star.nvars))
    c = ctxs["fast"]
    c.set_option(pkg.OPT_WORKGROUP, 64)
    c.set_option(pkg.OPT_BINS_PER_THREAD, 8)
    c.set_spectrum(star.x, y)
    c.set_option(pkg.OPT_FD_WINDOWED, 0)
    l0_f, g_f = c.fd_gradient(star.model_id, P, star.plength, idx, h, T, 1.0)
    c.set_option(pkg.OPT_FD_WINDOWED, 1)
    c.set_option(pkg.OPT_TIMING, 1)
    c.reset_kernel_stats()
    l0_w, g_w = c.fd_gradient(star.model_id, P, star.plength, idx, h, T, 1.0)
    bins, evals = c.fd_stats()
    c.set_option(pkg.OPT_TIMING, 0)
    assert evals == 3 * (idx.size + 1) and 0 < bins < 0.2 * evals * star.x.size, (bins, evals)
    assert np.allclose(l0_w, l0_f, rtol=1e-12)
    scale = np.max(np.abs(g_f), axis=1, keepdims=True)
    tol = 5e-15 * star.x.size / h[None, :] + 1e-6 * scale
    assert np.all(np.abs(g_w - g_f) <= tol), np.max(np.abs(g_w - g_f) / tol)
    _, l0_o, g_o = oracle.fd_gradient(star.model_id, star.params, star.plength, idx, h, star.x, y, 1.0, 1.0)
    assert np.all(np.abs(g_w[0] - g_o) <= 5e-14 * star.x.size / h + 1e-6 * np.max(np.abs(g_o)))


def test_many_multiplets_multiple_chunks(pkg, oracle, synth, ctxs):
    """More than 64 multiplets per evaluation (the kernel stages them in chunks of 64): 30 radial orders x l<=3 = 120
    multiplets (BASELINE config C5 has O(100-300)); every mode and geometry family, plus the windowed gradient."""
    rng = np.random.default_rng(21)
    p, pl = synth.make_params_aj_model(rng, lmax=3, nfreqs=30, dnu=40.0, epsilon=0.3, d0l=-0.4, asym=8.0, n_first=8,
                                       noise=np.array([2.0, 30.0, 2.0, 1.0, 3.0, 2.5, 0.05]), dl_shift=(0, 0, -1, -1))
    o = pl[0] + pl[1] + pl[2:6].sum()
    p[o] = 0.6          # a1
    p[o + 14:o + 14 + 30] = rng.uniform(0.1, 0.4, 30)   # narrow widths
    x = synth.grid(60000, 250.0, 0.025)   # 250..1750 muHz
    st, m_o = oracle.call_model(23, p, pl, x)
    assert st == 0
    y = m_o * np.random.default_rng(3).exponential(1.0, x.size)
    ref = oracle.chi22p_ld(y, m_o, 1)
    for name, wg, K, tol_m in (("strict", 256, 2, 1e-15), ("strict", 64, 8, 1e-15), ("fast", 64, 8, 1e-12), ("fast", 256, 4, 1e-12),
                               ("fast_direct", 256, 4, 1e-12)):
        c = ctxs[name]
        c.set_option(pkg.OPT_WORKGROUP, wg)
        c.set_option(pkg.OPT_BINS_PER_THREAD, K)
        c.set_spectrum(x, y)
        logL, model, status = c.loglike_params_batch(23, p, pl, want_model=True)
        assert status[0] == 0
        assert np.max(np.abs(model[0] - m_o) / m_o) <= tol_m, (name, wg, K)
        assert abs(logL[0] - ref) <= 1e-11 * abs(ref)
    # windowed vs brute-force gradient over a handful of variables (heights, one frequency of each degree, a1, widths, noise)
    idx = np.array([0, 7, o - 90, o - 60, o - 30, o - 1, o, o + 14, o + 20, o + 14 + 30 + 3, o + 14 + 30 + 6], dtype=np.int32)
    h = 1e-6 * np.maximum(np.abs(p[idx]), 1e-2)
    c = ctxs["fast"]
    c.set_option(pkg.OPT_WORKGROUP, 64)
    c.set_option(pkg.OPT_BINS_PER_THREAD, 8)
    c.set_spectrum(x, y)
    c.set_option(pkg.OPT_FD_WINDOWED, 0)
    _, g_f = c.fd_gradient(23, p, pl, idx, h, [1.0], 1.0)
    c.set_option(pkg.OPT_FD_WINDOWED, 1)
    _, g_w = c.fd_gradient(23, p, pl, idx, h, [1.0], 1.0)
    assert np.all(np.abs(g_w - g_f) <= 5e-14 * x.size / h + 1e-6 * np.max(np.abs(g_f)))


def _numpy_table_model(x, mults, noise, nh):
    """The path's arithmetic written out in numpy (build_lorentzian.cpp:131-246 per component on its window, noise_models.cpp:15-39)."""
    m = np.zeros_like(x)
    for r in mults:
        sl = slice(int(r["i0"]), int(r["i1"]))
        xs = x[sl]
        res = np.zeros_like(xs)
        for k in range(2 * int(r["l"]) + 1):
            u = 4.0 * (xs - r["nu"][k]) ** 2 / r["gamma"] ** 2
            prof = r["hv"][k] / (1.0 + u)
            if r["asym"] != 0.0:
                prof = prof * ((1.0 + r["asym"] * (xs / r["fc"] - 1.0)) ** 2 + (0.5 * r["gamma"] * r["asym"] / r["fc"]) ** 2)
            res += prof
        m[sl] += res
    for k in range(nh):
        if noise[3 * k + 1] != 0:
            m += noise[3 * k] / (1.0 + (1e-3 * noise[3 * k + 1] * x) ** noise[3 * k + 2])
    return m + noise[-1]


@pytest.mark.parametrize("nx", [3, 64, 511, 512, 513, 1025, 4097])
def test_random_tables_on_awkward_grids(pkg, ctxs, nx):
    """Table-level entry (tamcmc_hip_loglike_batch) against the formula written out in numpy: grids shorter than a wave, one bin
    either side of a tile boundary, windows clipped at both ends or one bin wide, zero to 150 multiplets per evaluation (more than
    one staging chunk), widths from a fraction of a bin to the whole grid, asymmetric profiles, 0-3 Harvey terms, several
    evaluations of different lengths in one launch -- all three arithmetic modes and both kernel geometries."""
    rng = np.random.default_rng(nx)
    step = 0.05
    x = 1000.0 + step * np.arange(nx)
    B = 5
    counts = [0, 1, int(rng.integers(2, 9)), 70, 150]
    tabs, offsets, noises, nhs = [], [0], [], []
    for b in range(B):
        t = np.zeros(counts[b], dtype=pkg.MULT_DTYPE)
        for r in t:
            l = int(rng.integers(0, 4))
            fc = x[0] + rng.uniform(-0.1, 1.1) * (x[-1] - x[0] + step)
            gam = float(np.exp(rng.uniform(np.log(0.3 * step), np.log(max(nx * step, step)))))
            i0 = int(rng.integers(0, nx))
            i1 = int(rng.integers(i0 + 1, nx + 1))
            if rng.random() < 0.3:
                i0, i1 = 0, nx
            r["l"], r["i0"], r["i1"], r["fc"], r["gamma"] = l, i0, i1, fc, gam
            r["asym"] = rng.choice([0.0, 0.0, rng.uniform(-30, 30)])
            for k in range(2 * l + 1):
                r["nu"][k] = fc + (k - l) * rng.uniform(0.0, 0.4)
                r["hv"][k] = rng.uniform(0.1, 20.0)
        nh = int(rng.integers(0, 4))
        nz = np.zeros(10)
        for k in range(nh):
            nz[3 * k:3 * k + 3] = [rng.uniform(0.5, 5), rng.uniform(0.2, 2.0), rng.uniform(1.0, 4.0)]
        nz = np.concatenate([nz[:3 * nh], [rng.uniform(0.2, 2.0)]])
        noises.append(np.pad(nz, (0, 10 - nz.size)))
        nhs.append(nh)
        tabs.append(t)
        offsets.append(offsets[-1] + t.size)
    mults = np.concatenate(tabs)
    noise = np.stack(noises)
    nn = [3 * h + 1 for h in nhs]
    ref_m = np.stack([_numpy_table_model(x, tabs[b], noises[b][:nn[b]], nhs[b]) for b in range(B)])
    y = ref_m[2] * rng.exponential(1.0, nx)
    T = 1.3 ** np.arange(B)
    ref_l = np.array([-(y / ref_m[b] + np.log(ref_m[b])).sum() / T[b] for b in range(B)])
    for name, tol_m, tol_l in (("strict", 1e-12, 1e-12), ("fast_direct", 1e-11, 1e-11), ("fast", 1e-10, 1e-10)):
        for wg in (64, 256):
            c = ctxs[name]
            c.set_option(pkg.OPT_WORKGROUP, wg)
            c.set_spectrum(x, y)
            logL, model = c.loglike_batch(mults, np.array(offsets, dtype=np.int32), noise, nhs, nn, T, 1.0, want_model=True)
            assert np.max(np.abs(model - ref_m) / ref_m) < tol_m, (name, wg, np.max(np.abs(model - ref_m) / ref_m))
            assert np.allclose(logL, ref_l, rtol=tol_l, atol=1e-9), (name, wg, logL, ref_l)
