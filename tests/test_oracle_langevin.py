"""Known-answer tests of the Langevin-step oracle (oracle/sampler_oracle.c, second half).  The reference holds no such step (D_MALA and
multinormal_logpdf are stubs, MALA.cpp:321-337; use_drift = 1 is fatal, :496-500), so the oracle is pinned here by what the step must
satisfy by definition: the multivariate-normal density against scipy, the drift by hand, the finite-difference gradient against
central differences, the whole iteration against the same step assembled from numpy / scipy pieces, and DETAILED BALANCE of the
Metropolis-Hastings kernel  pi(x) q(x'|x) a(x -> x') = pi(x') q(x|x') a(x' -> x)  checked by making the reverse move explicitly.
No GPU, no product code."""
import numpy as np
from scipy.stats import multivariate_normal

LAM = 1.6


def _star(oracle, synth, nx=1536, seed=4):
    star = synth.make_c2_star(nx=nx)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    y = star.set_spectrum_from_model(m0, seed=seed)
    return star, y


def _state(oracle, star, y, vars_, T):
    C_ = len(T)
    params = np.tile(star.params, (C_, 1))
    params[:, star.index_to_relax] = vars_
    logL = oracle.loglike_batch(star.model_id, params, star.plength, star.x, y, 1.0, T)[0]
    logPr = np.array([oracle.call_prior(star, p) for p in params])
    return dict(params=params, vars=np.array(vars_, copy=True), logL=logL, logPrior=logPr, logPost=logL + logPr)


def test_mvn_logpdf_against_scipy(oracle):
    rng = np.random.default_rng(1)
    for n, cond in ((1, 1.0), (2, 10.0), (7, 1e3), (40, 1e5)):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        M = (Q * np.geomspace(1.0, cond, n)) @ Q.T
        M = (M + M.T) / 2
        v, mean = rng.standard_normal(n), rng.standard_normal(n)
        assert np.isclose(oracle.mvn_logpdf(v, mean, M), multivariate_normal.logpdf(v, mean, M), rtol=1e-9, atol=1e-9)
    # a proposal covariance as the sampler holds it: scales from 1e-5 (frequencies) to 1e2 (heights) -> condition number ~1e14.
    # M = D R D with a well-conditioned correlation matrix R: log N(v; mean, M) = log N(D^-1 (v - mean); 0, R) - sum log D_ii exactly
    n = 93
    A = rng.standard_normal((n, n)) * 0.1
    R = np.eye(n) + A @ A.T
    R = R / np.sqrt(np.outer(np.diag(R), np.diag(R)))
    D = np.geomspace(1e-5, 1e2, n)[rng.permutation(n)]
    M = R * np.outer(D, D)
    v, mean = rng.standard_normal(n) * D, rng.standard_normal(n) * D
    ref = multivariate_normal.logpdf((v - mean) / D, np.zeros(n), R) - np.log(D).sum()
    assert np.isclose(oracle.mvn_logpdf(v, mean, M), ref, rtol=1e-11)
    # by hand: N(1; 0, 4) = exp(-1/8) / sqrt(8 pi)
    assert np.isclose(oracle.mvn_logpdf([1.0], [0.0], [[4.0]]), -0.125 - 0.5 * np.log(8 * np.pi), rtol=1e-15)
    assert np.isnan(oracle.mvn_logpdf([1.0, 1.0], [0.0, 0.0], [[1.0, 1.0], [1.0, 1.0]]))      # singular


def test_drift_by_hand(oracle):
    """drift = (1/2) sigma (covarmat + eps2 I) grad min(1, delta/|grad|).  covarmat = [[2,1],[1,3]], eps2 = 1, sigma = 0.5, grad = (3,4):
    (Sigma + I) grad = (13, 19) -> drift = 0.25 (13, 19); with delta = 1 the gradient is scaled by 1/5."""
    cov, g = np.array([[2.0, 1.0], [1.0, 3.0]]), np.array([3.0, 4.0])
    assert np.allclose(oracle.langevin_drift(cov, 0.5, 1.0, 0.0, g), [3.25, 4.75], rtol=1e-15)
    assert np.allclose(oracle.langevin_drift(cov, 0.5, 1.0, 1.0, g), [0.65, 0.95], rtol=1e-15)
    assert np.allclose(oracle.langevin_drift(cov, 0.5, 1.0, 10.0, g), [3.25, 4.75], rtol=1e-15)     # |grad| below delta: untouched
    assert np.array_equal(oracle.langevin_drift(cov, 0.5, 1.0, 0.0, [np.inf, 1.0]), [0.0, 0.0])      # no finite norm: no drift


def test_fd_gradient_against_central_differences(oracle, synth):
    star, y = _star(oracle, synth)
    T = 1.6
    th = star.params.copy()
    idx = star.index_to_relax
    h = 1e-6 * np.maximum(np.abs(th[idx]), 1e-3)
    st, g, gp = oracle.fd_gradient_posterior(star, y, th, T, h)
    assert st == 0 and np.all(np.isfinite(g))

    def post_parts(p):
        l = oracle.loglike_batch(star.model_id, p, star.plength, star.x, y, 1.0, [T])[0][0]
        return l, oracle.call_prior(star, p)

    l0, pr0 = post_parts(th)
    for k in range(idx.size):
        # (1) the same forward difference from two rounded sums (noise ~1e-12 / h): equal to 1e-5 of the gradient's scale
        pp = th.copy()
        pp[idx[k]] += h[k]
        happ = pp[idx[k]] - th[idx[k]]
        lp, prp = post_parts(pp)
        scale = max(abs(g[k]), 1e-3 * np.abs(g).max())
        assert abs(g[k] - (lp + prp - l0 - pr0) / happ) < 1e-5 * scale, (k, g[k], (lp + prp - l0 - pr0) / happ)
        assert abs(gp[k] - (prp - pr0) / happ) < 1e-5 * scale + 1e-9
        # (2) a central difference of twice the step: the forward difference's own truncation error is ~h/(2 sigma_k), percent level for
        #     the frequencies -- the two must agree to that
        pm, pp2 = th.copy(), th.copy()
        pp2[idx[k]] += 2 * h[k]
        pm[idx[k]] -= 2 * h[k]
        (lp2, prp2), (lm, prm) = post_parts(pp2), post_parts(pm)
        gc = (lp2 + prp2 - lm - prm) / (pp2[idx[k]] - pm[idx[k]])
        assert abs(g[k] - gc) < 0.05 * scale, (k, g[k], gc)
    # the temperature divides the likelihood share only
    _, g2, gp2 = oracle.fd_gradient_posterior(star, y, th, 2 * T, h)
    assert np.allclose(gp2, gp, rtol=0, atol=0) and np.allclose(g2 - gp2, (g - gp) / 2, rtol=1e-12, atol=1e-12 * np.abs(g).max())


def test_prior_share_at_the_edge_of_the_support(oracle, synth):
    """A variable with a uniform prior sitting closer than h to the upper bound: the forward point is outside the support, the backward
    difference is used; boxed in on both sides the share is zero."""
    star, y = _star(oracle, synth)
    idx = star.index_to_relax
    uni = [k for k in range(idx.size) if star.priors_switch[idx[k]] == 1]       # primepriors_ctrl.list: 1 = Uniform
    assert uni
    k = uni[0]
    j = idx[k]
    lo, hi = star.priors[0, j], star.priors[1, j]
    th = star.params.copy()
    h = 1e-6 * np.maximum(np.abs(th[idx]), 1e-3)
    th[j] = hi - 0.25 * h[k]
    assert np.isfinite(oracle.call_prior(star, th))
    st, g, gp = oracle.fd_gradient_posterior(star, y, th, 1.0, h)
    assert np.isfinite(g[k]) and gp[k] == 0.0 and g[k] != 0.0     # uniform density: flat inside; the likelihood share survives
    hh = h.copy()
    hh[k] = 2 * (hi - lo)                                          # both neighbours outside
    st, g2, gp2 = oracle.fd_gradient_posterior(star, y, th, 1.0, hh)
    assert gp2[k] == 0.0


def _numpy_step(oracle, star, y, T, st0, law, z, u, fd_step_rel, delta, eps2=1e-12):
    """The Langevin iteration of chain-by-chain numpy / scipy pieces (no swap, no adaptation)."""
    mu, cov, sigma = law
    idx = star.index_to_relax
    h = fd_step_rel * np.maximum(np.abs(mu[0]), 1e-3)
    out = []
    for m in range(len(T)):
        M = (cov[m] + eps2 * np.eye(idx.size)) * sigma[m]

        def drift(p):
            _, g, _ = oracle.fd_gradient_posterior(star, y, p, T[m], h)
            n = np.linalg.norm(g)
            return 0.5 * M @ (g * (min(1.0, delta / n) if delta > 0 else 1.0))

        x = st0["vars"][m]
        d0 = drift(st0["params"][m])
        xp = x + d0 + np.linalg.cholesky(M) @ z[m]
        p_new = st0["params"][m].copy()
        p_new[idx] = xp
        pr = oracle.call_prior(star, p_new)
        if pr == -np.inf:
            out.append((xp, 0.0, False))
            continue
        l = oracle.loglike_batch(star.model_id, p_new, star.plength, star.x, y, 1.0, [T[m]])[0][0]
        d1 = drift(p_new)
        lq_fwd = multivariate_normal.logpdf(xp, x + d0, M)
        lq_rev = multivariate_normal.logpdf(x, xp + d1, M)
        r = min(1.0, np.exp(l + pr - st0["logPost"][m] + lq_rev - lq_fwd))
        out.append((xp, r, u[m] <= r))
    return out


def test_whole_iteration_against_numpy_pieces(oracle, synth):
    star, y = _star(oracle, synth)
    C_, Nv = 4, star.nvars
    T = LAM ** np.arange(C_)
    rng = np.random.default_rng(12)
    v0 = np.tile(star.params[star.index_to_relax], (C_, 1))
    st0 = _state(oracle, star, y, v0, T)
    err = 2e-4 * np.maximum(np.abs(v0[0]), 1.0)
    A = rng.standard_normal((C_, Nv, Nv)) * 0.3
    cov = np.array([np.diag(err) @ (np.eye(Nv) + a @ a.T) @ np.diag(err) for a in A])      # correlated proposal laws
    law = (v0.copy(), cov, 2.38 ** 2 * T ** 0.2 / Nv)
    z, u = rng.standard_normal((C_, Nv)), np.array([1e-9, 0.3, 0.6, 1.0 - 1e-12])
    for delta in (0.0, 50.0):
        st, _, rc = oracle.sampler_iteration(star, y, T, st0["logL"], st0, law, i=7, z=z, u_mh=u, use_drift=True, fd_step_rel=1e-6, delta=delta)
        assert rc == 0
        ref = _numpy_step(oracle, star, y, T, st0, law, z, u, 1e-6, delta)
        for m, (xp, r, acc) in enumerate(ref):
            assert np.allclose(st["prop_vars"][m], xp, rtol=1e-12, atol=1e-14)
            assert np.isclose(st["Pmove"][m], r, rtol=1e-7, atol=1e-300), (m, st["Pmove"][m], r)
            assert bool(st["moved"][m]) == bool(acc)
            assert np.array_equal(st["vars"][m], st["prop_vars"][m] if acc else st0["vars"][m])
        assert st["diag"][:, 2].min() > 0                       # a drift was applied
        assert 0 < st["moved"].sum() < C_ or delta > 0
    # with fd gradients switched to zero drift (delta tiny -> drift ~ 0) the step degenerates to the random-walk step of MALA.cpp:339-369
    st_rw, _, _ = oracle.sampler_iteration(star, y, T, st0["logL"], st0, law, i=7, z=z, u_mh=u)
    st_d, _, _ = oracle.sampler_iteration(star, y, T, st0["logL"], st0, law, i=7, z=z, u_mh=u, use_drift=True, fd_step_rel=1e-6, delta=1e-30)
    assert np.allclose(st_d["prop_vars"], st_rw["prop_vars"], rtol=1e-13) and np.allclose(st_d["Pmove"], st_rw["Pmove"], rtol=1e-6, atol=1e-300)


def test_detailed_balance_of_the_langevin_kernel(oracle, synth):
    """pi(x) q(x'|x) a(x,x') = pi(x') q(x|x') a(x',x): the forward move x -> x' from draws z; the reverse move x' -> x is made by the
    draws z' = L^-1 (x - x' - drift(x')), so the second call proposes x exactly; the two sides are compared in logs.  A swapped pair of
    densities, a wrong sign of the drift in either density or a dropped term breaks the identity at O(1)."""
    star, y = _star(oracle, synth)
    C_, Nv = 3, star.nvars
    T = LAM ** np.arange(C_)
    rng = np.random.default_rng(5)
    v0 = np.tile(star.params[star.index_to_relax], (C_, 1))
    v0 *= 1 + 1e-4 * rng.standard_normal(v0.shape)
    st0 = _state(oracle, star, y, v0, T)
    err = 3e-4 * np.maximum(np.abs(v0[0]), 1.0)
    cov = np.tile(np.diag(err ** 2), (C_, 1, 1))
    sigma = 2.38 ** 2 * T ** 0.2 / Nv
    law = (np.tile(star.params[star.index_to_relax], (C_, 1)), cov, sigma)
    z = rng.standard_normal((C_, Nv))
    fwd, _, _ = oracle.sampler_iteration(star, y, T, st0["logL"], st0, law, i=3, z=z, u_mh=np.ones(C_), use_drift=True, fd_step_rel=1e-6)
    assert np.all(fwd["prop_stats"][:, 2] > -np.inf)
    # reverse: start AT the proposals
    st1 = _state(oracle, star, y, fwd["prop_vars"], T)
    assert np.allclose(st1["logPost"], fwd["prop_stats"][:, 2], rtol=1e-13)
    idx = star.index_to_relax
    h = 1e-6 * np.maximum(np.abs(law[0][0]), 1e-3)
    zr = np.zeros_like(z)
    for m in range(C_):
        _, g, _ = oracle.fd_gradient_posterior(star, y, st1["params"][m], T[m], h)
        M = (cov[m] + 1e-12 * np.eye(Nv)) * sigma[m]
        d1 = 0.5 * M @ g
        zr[m] = np.linalg.solve(np.linalg.cholesky(M), v0[m] - fwd["prop_vars"][m] - d1)
    rev, _, _ = oracle.sampler_iteration(star, y, T, st0["logL"], st1, law, i=3, z=zr, u_mh=np.ones(C_), use_drift=True, fd_step_rel=1e-6)
    assert np.allclose(rev["prop_vars"], v0, rtol=1e-9, atol=1e-12)                 # the reverse move lands on x
    # densities seen from both ends agree, and the balance holds
    assert np.allclose(rev["diag"][:, 0], fwd["diag"][:, 1], rtol=1e-7, atol=1e-6)  # q(x|x') as forward density of the second call
    assert np.allclose(rev["diag"][:, 1], fwd["diag"][:, 0], rtol=1e-7, atol=1e-6)
    lhs = st0["logPost"] + fwd["diag"][:, 0] + np.log(fwd["Pmove"])
    rhs = st1["logPost"] + rev["diag"][:, 0] + np.log(rev["Pmove"])
    assert np.allclose(lhs, rhs, rtol=0, atol=1e-5), lhs - rhs
    assert np.any(fwd["Pmove"] < 1) and np.all(np.minimum(fwd["Pmove"], rev["Pmove"]) < 1)
    assert np.all(np.maximum(fwd["Pmove"], rev["Pmove"]) == 1.0)                    # min(1, r) and min(1, 1/r): one of them is 1


def test_swap_and_adaptation_follow_the_random_walk_iteration(oracle, synth):
    """The Langevin iteration shares the swap (MALA.cpp:397-461) and the Robbins-Monro update (:296-319) with the random-walk one: with
    the same post-test state both produce the same swap and the same law."""
    star, y = _star(oracle, synth)
    C_, Nv = 4, star.nvars
    T = LAM ** np.arange(C_)
    rng = np.random.default_rng(2)
    v0 = np.tile(star.params[star.index_to_relax], (C_, 1))
    st0 = _state(oracle, star, y, v0, T)
    err = 1e-3 * np.maximum(np.abs(v0[0]), 1.0)
    law = (v0.copy(), np.tile(np.diag(err ** 2), (C_, 1, 1)), 2.38 ** 2 * T ** 0.2 / Nv)
    z = rng.standard_normal((C_, Nv))
    u = np.ones(C_)                      # nothing moves (r < 1 almost surely; u = 1 refuses unless r = 1)
    a, lawa, _ = oracle.sampler_iteration(star, y, T, st0["logL"], st0, law, i=30, z=z, u_mh=u, learn=True, do_swap=True, ind_A=1, u_swap=0.0, c0=4.0)
    b, lawb, _ = oracle.sampler_iteration(star, y, T, st0["logL"], st0, law, i=30, z=z, u_mh=u, learn=True, do_swap=True, ind_A=1, u_swap=0.0, c0=4.0,
                                          use_drift=True, fd_step_rel=1e-6)
    if not a["moved"].any() and not b["moved"].any():
        for k in ("vars", "logL", "logPrior", "logPost"):
            assert np.array_equal(a[k], b[k]), k
        assert a["swapped"] == b["swapped"] == 1
        assert np.array_equal(lawa[0], lawb[0]) and np.array_equal(lawa[1], lawb[1])
        assert not np.array_equal(lawa[2], lawb[2]) or np.array_equal(a["Pmove"], b["Pmove"])   # sigma follows each step's own Pmove
