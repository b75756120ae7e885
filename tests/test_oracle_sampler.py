"""Known-answer tests of the sampler-step oracle (oracle/sampler_oracle.c = MALA.cpp:135-176, :296-319, :339-369, :397-461, :490-551):
every expectation below is worked out by hand (or with plain numpy) from the reference's statements.  No GPU, no product code."""
import ctypes as C

import numpy as np

from oracle_lib import _dp, _ip

LD = C.c_longdouble


def test_clip_functions(oracle):
    L = oracle.lib
    # p1 (MALA.cpp:135-151): identity inside [epsilon1, A1], clipped outside
    assert float(L.orc_p1_fct(LD(0.3), LD(1e-12), LD(1e14))) == 0.3
    assert float(L.orc_p1_fct(LD(-2.0), LD(1e-12), LD(1e14))) == 1e-12
    assert float(L.orc_p1_fct(LD(1e20), LD(1e-12), LD(1e14))) == 1e14
    # p3 (:166-176): a vector of norm 5 against A1 = 2 is rescaled to norm 2, below A1 it is untouched
    v = np.array([3.0, 4.0])
    L.orc_p3_fct(_dp(v), 2, 2.0)
    assert np.allclose(v, [1.2, 1.6], rtol=1e-15)
    v = np.array([3.0, 4.0])
    L.orc_p3_fct(_dp(v), 2, 5.0)
    assert np.array_equal(v, [3.0, 4.0])
    # p2 (:153-164): Frobenius norm of [[1,2],[2,4]] is 5
    M = np.array([[1.0, 2.0], [2.0, 4.0]])
    L.orc_p2_fct(_dp(M), 2, 1.0)
    assert np.allclose(M, np.array([[1.0, 2.0], [2.0, 4.0]]) / 5.0, rtol=1e-15)


def test_update_proposal_by_hand(oracle):
    """MALA.cpp:296-319 with gamma = 0.5, vars = (2, 0), mu = (0, 0), covarmat = I, sigma = 1, acceptance 1, target 0.25:
    mu' = (1, 0); d = vars - mu' = (1, 0); cov' = I + 0.5 ([[1,0],[0,0]] - I) = [[1,0],[0,0.5]]; sigma' = 1 + 0.5 (1 - 0.25) = 1.375."""
    L = oracle.lib
    mu, cov, sig, v = np.zeros(2), np.eye(2), np.array([1.0]), np.array([2.0, 0.0])
    L.orc_update_proposal(_dp(mu), _dp(cov), _dp(sig), _dp(v), 2, LD(1.0), LD(0.5), LD(0.25), LD(1e-12), LD(1e14))
    assert np.array_equal(mu, [1.0, 0.0]) and np.array_equal(cov, [[1.0, 0.0], [0.0, 0.5]]) and sig[0] == 1.375
    # the deviation uses the UPDATED mu (:311): with the old mu the (0,0) entry would be 1 + 0.5 (4 - 1) = 2.5
    # sigma is clipped by p1: a large negative step lands on epsilon1
    sig = np.array([0.1])
    L.orc_update_proposal(_dp(mu), _dp(cov), _dp(sig), _dp(v), 2, LD(0.0), LD(10.0), LD(0.234), LD(1e-12), LD(1e14))
    assert sig[0] == 1e-12
    # gamma = 1 replaces mu by vars and the covariance by the (zero) outer product
    mu, cov, sig = np.array([5.0, 5.0]), np.eye(2) * 3, np.array([1.0])
    L.orc_update_proposal(_dp(mu), _dp(cov), _dp(sig), _dp(v), 2, LD(0.234), LD(1.0), LD(0.234), LD(1e-12), LD(1e14))
    assert np.array_equal(mu, v) and np.array_equal(cov, np.zeros((2, 2))) and sig[0] == 1.0


def test_new_prop_values_is_x_plus_chol_z(oracle):
    """MALA.cpp:339-355: x' = x + L z with L L^T = (covarmat + epsilon2 I) sigma.  2x2 by hand: covarmat = [[4,2],[2,3]], eps2 = 0,
    sigma = 1 -> L = [[2,0],[1,sqrt(2)]]; z = (1, 1) -> x' = x + (2, 1 + sqrt 2)."""
    L = oracle.lib
    cov, x, z, out, Lf = np.array([[4.0, 2.0], [2.0, 3.0]]), np.array([10.0, 20.0]), np.array([1.0, 1.0]), np.zeros(2), np.zeros((2, 2))
    assert L.orc_new_prop_values(_dp(cov), 1.0, 0.0, _dp(x), _dp(z), 2, _dp(out), _dp(Lf)) == 0
    assert np.allclose(Lf, [[2.0, 0.0], [1.0, np.sqrt(2.0)]], rtol=1e-15) and np.allclose(out, [12.0, 21.0 + np.sqrt(2.0)], rtol=1e-15)
    # random symmetric positive definite case against numpy's factor; epsilon2 sits on the diagonal BEFORE the sigma scaling (:348)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((7, 7))
    cov, sig, eps = A @ A.T + np.eye(7), 0.37, 1e-3
    x, z, out, Lf = rng.standard_normal(7), rng.standard_normal(7), np.zeros(7), np.zeros((7, 7))
    assert L.orc_new_prop_values(_dp(cov), sig, eps, _dp(x), _dp(z), 7, _dp(out), _dp(Lf)) == 0
    Lnp = np.linalg.cholesky((cov + eps * np.eye(7)) * sig)
    assert np.allclose(Lf, Lnp, rtol=1e-13) and np.allclose(out, x + Lnp @ z, rtol=1e-13)
    # not positive definite -> flagged (the reference would carry Eigen's partial factor on)
    bad = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert L.orc_new_prop_values(_dp(bad), 1.0, 0.0, _dp(x[:2].copy()), _dp(z[:2].copy()), 2, _dp(np.zeros(2)), None) == 1


def test_accept_rule_cases(oracle):
    """MALA.cpp:490-551: r = min(1, exp(dlogPost)); NaN likelihood -> r = 0 (:522-524); -inf posterior -> r = 0 (:491-493); a NaN
    ratio stops the reference (:519-521); the move is made when u <= r (:536)."""
    L = oracle.lib
    r = np.zeros(1)
    assert L.orc_mh_accept(-10.0, -10.0, -12.0, 0.999, _dp(r)) == 1 and r[0] == 1.0            # uphill: always
    assert L.orc_mh_accept(-10.0, -13.0, -12.0, 0.3, _dp(r)) == 1 and np.isclose(r[0], np.exp(-1.0), rtol=1e-15)
    assert L.orc_mh_accept(-10.0, -13.0, -12.0, 0.4, _dp(r)) == 0 and np.isclose(r[0], np.exp(-1.0), rtol=1e-15)
    assert L.orc_mh_accept(-10.0, -13.0, -12.0, np.exp(-1.0), _dp(r)) == 1                        # u == r: "<=" moves
    assert L.orc_mh_accept(float("nan"), -1.0, -12.0, 1e-300, _dp(r)) == 0 and r[0] == 0.0       # NaN model: never
    assert L.orc_mh_accept(-10.0, -np.inf, -12.0, 1e-300, _dp(r)) == 0 and r[0] == 0.0          # outside a prior's support: never
    assert L.orc_mh_accept(-10.0, np.nan, -12.0, 0.5, _dp(r)) == -1                               # NaN ratio: the reference stops
    assert L.orc_mh_accept(-10.0, -5.0, -np.inf, 0.5, _dp(r)) == 1 and r[0] == 1.0              # from a -inf start everything is uphill


def _pt_case(oracle, literal, u):
    L = oracle.lib
    T = np.array([1.0, 2.0, 4.0])
    logL = np.array([-100.0, -60.0, -20.0])            # tempered: chain m holds L_m / T_m
    logPr = np.array([-1.0, -2.0, -3.0])
    logPo = logL + logPr
    vars_ = np.arange(6.0).reshape(3, 2)
    params = np.arange(9.0).reshape(3, 3) * 10
    moved, Pmove, Pswap = np.array([1, 0, 1], dtype=np.int32), np.array([0.9, 0.1, 0.5]), np.zeros(1)
    sw = L.orc_parallel_tempering(_dp(logL), _dp(logPr), _dp(logPo), _dp(vars_), _dp(params), _ip(moved), _dp(Pmove), _dp(T), 2, 3, 0, u,
                                  int(literal), _dp(Pswap))
    return sw, logL, logPr, logPo, vars_, params, moved, Pmove, Pswap[0]


def test_parallel_tempering_by_hand(oracle):
    """Pair (0, 1), T = (1, 2): logL_A_TB = -100 * 1/2 = -50, logL_B_TA = -60 * 2/1 = -120, r_T = min(1, exp(-50 - 120 + 100 + 60)) =
    exp(-10).  Swapped: A gets B's rows, logL = -120, prior -2, posterior -122; B gets A's rows, logL = -50, prior -1 and posterior
    -50 + (-1) = -51 (consistent rule) or -50 + (-2) = -52 (MALA.cpp:444 as executed: logPrior[A] was already overwritten at :433)."""
    for literal, post_B in ((0, -51.0), (1, -52.0)):
        sw, logL, logPr, logPo, v, p, moved, Pmove, Ps = _pt_case(oracle, literal, 1e-6)
        assert sw == 1 and np.isclose(Ps, np.exp(-10.0), rtol=1e-15)
        assert np.array_equal(logL, [-120.0, -50.0, -20.0]) and np.array_equal(logPr, [-2.0, -1.0, -3.0])
        assert logPo[0] == -122.0 and logPo[1] == post_B and logPo[2] == -23.0
        assert np.array_equal(v, [[2.0, 3.0], [0.0, 1.0], [4.0, 5.0]]) and np.array_equal(p[0], [30.0, 40.0, 50.0]) and np.array_equal(p[1], [0.0, 10.0, 20.0])
        assert np.array_equal(moved, [0, 1, 1]) and np.array_equal(Pmove, [0.1, 0.9, 0.5])   # moved / Pmove travel with the rows (:436-437, :446-447)
    sw, logL, logPr, logPo, v, p, moved, Pmove, Ps = _pt_case(oracle, 0, 0.5)                # u > r_T: nothing changes
    assert sw == 0 and Ps == 0.0 and np.array_equal(logL, [-100.0, -60.0, -20.0]) and np.array_equal(v, np.arange(6.0).reshape(3, 2))
    assert np.array_equal(moved, [1, 0, 1])


def test_learning_schedule(oracle):
    """MALA.cpp:656-667 with the default Nt_learn = 1000, 1500, 100000 and periods 1, 10: no learning below 1000 or from 100000 on,
    every iteration in [1000, 1500), every tenth in [1500, 100000)."""
    L = oracle.lib
    Nt = (C.c_long * 3)(1000, 1500, 100000)
    per = (C.c_long * 2)(1, 10)
    f = lambda i: L.orc_learn_at(i, Nt, per, 2)
    assert [f(i) for i in (0, 999, 1000, 1001, 1499, 1500, 1501, 1510, 99990, 99999, 100000, 200000)] == [0, 0, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0]


def test_whole_iteration_on_a_small_star(oracle, synth):
    """orc_sampler_iteration assembles the pieces in the reference's order: checked against the same pieces called one by one from
    python (propose -> generate_model -> accept -> learn, then swap), on a C2-size star with 4 chains."""
    star = synth.make_c2_star(nx=2048)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    y = star.set_spectrum_from_model(m0, seed=3)
    C_, Nv, lam = 4, star.nvars, 1.7
    T = lam ** np.arange(C_)
    rng = np.random.default_rng(8)
    params = np.tile(star.params, (C_, 1))
    vars_ = params[:, star.index_to_relax].copy()
    logL = oracle.loglike_batch(star.model_id, params, star.plength, star.x, y, 1.0, T)[0]
    logPr = np.array([oracle.call_prior(star, p) for p in params])
    state = dict(params=params, vars=vars_, logL=logL, logPrior=logPr, logPost=logL + logPr)
    err = 1e-4 * np.maximum(np.abs(vars_[0]), 1.0)
    mu, cov, sigma = vars_.copy(), np.tile(np.diag(err ** 2), (C_, 1, 1)), 2.38 ** 2 * T ** 0.2 / Nv
    z, u = rng.standard_normal((C_, Nv)), np.array([1e-12, 1.0 - 1e-12, 1e-12, 0.5])   # comparators that force both outcomes
    for literal in (False, True):
        st, law, rc = oracle.sampler_iteration(star, y, T, logL, state, (mu, cov, sigma), i=25, z=z, u_mh=u, learn=True, do_swap=True, ind_A=1,
                                               u_swap=0.0, literal_444=literal, c0=5.0)
        assert rc == 0 and st["swapped"] == 1
        # the same iteration, piece by piece
        Lb = oracle.lib
        exp = {k: np.array(v, copy=True) for k, v in state.items()}
        mu2, cov2, sig2 = mu.copy(), cov.copy(), sigma.copy()
        moved, Pmove = np.zeros(C_, dtype=np.int32), np.zeros(C_)
        for m in range(C_):
            vnew = np.zeros(Nv)
            Lb.orc_new_prop_values(_dp(cov2[m]), sig2[m], 1e-12, _dp(exp["vars"][m]), _dp(np.ascontiguousarray(z[m])), Nv, _dp(vnew), None)
            assert np.array_equal(vnew, st["prop_vars"][m])
            pnew = exp["params"][m].copy()
            pnew[star.index_to_relax] = vnew
            pr = oracle.call_prior(star, pnew)
            l = oracle.loglike_batch(star.model_id, pnew, star.plength, star.x, y, 1.0, T[m:m + 1])[0][0] if pr != -np.inf else logL[m]
            po = l + pr if pr != -np.inf else -np.inf
            assert np.allclose(st["prop_stats"][m], [l, pr, po], rtol=1e-15, equal_nan=True)
            r = np.zeros(1)
            acc = Lb.orc_mh_accept(l, po, exp["logPost"][m], u[m], _dp(r))
            if acc == 1:
                exp["params"][m], exp["vars"][m], exp["logL"][m], exp["logPrior"][m], exp["logPost"][m] = pnew, vnew, l, pr, po
            moved[m], Pmove[m] = acc, r[0]
            Lb.orc_update_proposal(_dp(mu2[m]), _dp(cov2[m]), _dp(sig2[m:m + 1]), _dp(exp["vars"][m]), Nv, LD(Pmove[m]), LD(5.0 / 26.0), LD(0.234),
                                   LD(1e-12), LD(1e14))
        Ps = np.zeros(1)
        Lb.orc_parallel_tempering(_dp(exp["logL"]), _dp(exp["logPrior"]), _dp(exp["logPost"]), _dp(exp["vars"]), _dp(exp["params"]), _ip(moved),
                                  _dp(Pmove), _dp(T), Nv, params.shape[1], 1, 0.0, int(literal), _dp(Ps))
        for k in ("params", "vars", "logL", "logPrior", "logPost"):
            assert np.array_equal(st[k], exp[k]), k
        assert np.array_equal(st["moved"], moved) and np.array_equal(st["Pmove"], Pmove)
        assert np.array_equal(law[0], mu2) and np.array_equal(law[1], cov2) and np.array_equal(law[2], sig2)
        assert 1 <= moved.sum() < C_                              # the case exercises both branches of the accept rule
