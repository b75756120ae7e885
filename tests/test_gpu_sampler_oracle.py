"""Rows a10 / N4 against the ORACLE (-m gpu): one iteration of each engine (host-driven mirror of MALA::execute, device-resident
k_iterate / fused step) equals oracle/sampler_oracle.c's restatement of MALA.cpp:645-703 fed with the SAME random numbers
(tamcmc_sampler_draws), on the headline shape: C3 star, 1e5 bins x 111 parameters (93 free) x 20 tempered chains.
Cases: with and without adaptation, swap pair inside a chain group and across the two groups, swap accepted and refused, both
rules for chain B's stored posterior (MALA.cpp:444 as executed / consistent).  Stated tolerances: positions 1e-11 relative (the
Cholesky factor is unique only to rounding), tempered logL 2e-11 relative (FAST arithmetic, include/tamcmc_hip.h), move
probabilities 1e-4 relative (exp of a difference of two ~1e5-sized log-posteriors)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCH, LAM = 20, 1.3


@pytest.fixture(scope="module")
def c3(pkg, synth):
    star = synth.make_c3_star(seed=20240229, nx=100000, step=0.02)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    star.set_spectrum_from_model(m0[0], seed=20240301)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    yield star, ctx
    ctx.close()


def _full_state(star, s):
    st = s.state()
    params = np.tile(star.params, (NCH, 1))
    params[:, star.index_to_relax] = st["vars"]
    return dict(params=params, vars=st["vars"], logL=st["logL"], logPrior=st["logPrior"], logPost=st["logPost"]), st


def _one_iteration_against_oracle(oracle, star, s, init_logL, learn, literal, c0):
    before, raw = _full_state(star, s)
    law = s.proposal_law()
    it = raw["iteration"]
    z, u, u_swap, ind_A = s.draws(it)
    do_swap = it != 0                                   # dN_mixing = 1 (MALA.cpp:688)
    T = LAM ** np.arange(NCH)
    exp, law2, rc = oracle.sampler_iteration(star, star.y, T, init_logL, before, law, i=it, z=z, u_mh=u, learn=learn, do_swap=do_swap, ind_A=ind_A,
                                             u_swap=u_swap, literal_444=literal, c0=c0)
    assert rc == 0
    smp, stt = s.run(1, stats=True)
    after, raw2 = _full_state(star, s)
    assert raw2["iteration"] == it + 1
    assert np.allclose(after["vars"], exp["vars"], rtol=1e-11, atol=1e-13), np.max(np.abs(after["vars"] - exp["vars"]))
    assert np.array_equal(smp[0], after["vars"])                                         # the recorded sample IS the state after the swap step
    assert np.allclose(after["logL"], exp["logL"], rtol=2e-11, atol=0)
    assert np.allclose(after["logPrior"], exp["logPrior"], rtol=1e-12, atol=1e-12)
    assert np.allclose(after["logPost"], exp["logPost"], rtol=2e-11, atol=0)
    assert np.allclose(stt[0][:, 0], exp["logL"], rtol=2e-11) and np.allclose(stt[0][:, 2], exp["logPost"], rtol=2e-11)
    assert np.allclose(raw2["Pmove"], exp["Pmove"], rtol=1e-4, atol=1e-300)
    if learn:
        mu, cov, sig = s.proposal_law()
        assert np.allclose(mu, law2[0], rtol=1e-11, atol=1e-13) and np.allclose(sig, law2[2], rtol=1e-9)
        assert np.allclose(cov, law2[1], rtol=1e-9, atol=1e-9 * np.abs(law2[1]).max())
    return exp, ind_A, u_swap


def _find_iteration(s, start, want_A, u_lo, u_hi):
    for k in range(start, start + 20000):
        _, _, us, ia = s.draws(k)
        if ia == want_A and u_lo <= us <= u_hi:
            return k
    raise AssertionError("no such iteration")


@pytest.mark.parametrize("engine", ["host", "device"])
@pytest.mark.parametrize("swap_rule", [0, 1])
def test_one_iteration_equals_the_oracle_at_the_headline_shape(pkg, oracle, c3, engine, swap_rule):
    star, ctx = c3
    c0 = 2.0
    s = pkg.Sampler(ctx, star, nchains=NCH, lambda_temp=LAM, engine=engine, seed=77, Nt_learn=(10, 40), periods_learn=(1,), dN_mixing=1, c0=c0,
                    swap_rule=swap_rule)
    assert s.nvars == 93 and star.params.size == 111 and star.x.size == 100000
    init_logL = s.state()["logL"].copy()                # Model_def's init_logLikelihood (model_def.cpp:142-150): the start point's
    swaps_seen = {"in": 0, "across": 0, "refused": 0}
    quirk_visible = 0

    def note(exp, ind_A):
        nonlocal quirk_visible
        if exp["swapped"]:
            swaps_seen["across" if ind_A == NCH // 2 - 1 else "in"] += 1
        else:
            swaps_seen["refused"] += 1

    # (1) from the start point, no adaptation yet (iterations 0..2: iteration 0 has no swap step)
    for _ in range(3):
        exp, ia, _ = _one_iteration_against_oracle(oracle, star, s, init_logL, learn=False, literal=bool(swap_rule), c0=c0)
        note(exp, ia)
    s.run(7, record=False)                              # iterations 3..9
    # (2) with adaptation (iterations 10..39 learn): two checked steps, then let it adapt
    for _ in range(2):
        exp, ia, _ = _one_iteration_against_oracle(oracle, star, s, init_logL, learn=True, literal=bool(swap_rule), c0=c0)
        note(exp, ia)
    s.run(28, record=False)                             # iterations 12..39: the proposal law is adapted now
    assert s.state()["iteration"] == 40
    # (3) settled chains, no adaptation: a swap pair inside a chain group, the pair that straddles the two groups (chains 9 | 10), each
    #     with a comparator that accepts and one that refuses
    vars_now = s.state()["vars"]
    for want_A, (u_lo, u_hi) in ((3, (0.0, 0.02)), (NCH // 2 - 1, (0.0, 0.02)), (NCH // 2 - 1, (0.97, 1.0)), (14, (0.97, 1.0))):
        k = _find_iteration(s, 1000, want_A, u_lo, u_hi)
        s.set_state(vars_now, iteration=k)
        exp, ia, us = _one_iteration_against_oracle(oracle, star, s, init_logL, learn=False, literal=bool(swap_rule), c0=c0)
        assert ia == want_A
        note(exp, ia)
        if exp["swapped"]:
            # the two rules differ by (prior_A - prior_B) in chain B's stored posterior
            B = ia + 1
            quirk_visible += int(abs(exp["logPost"][B] - (exp["logL"][B] + exp["logPrior"][B])) > 1e-9)
        vars_now = s.state()["vars"]
    assert swaps_seen["in"] >= 1 and swaps_seen["across"] >= 1 and swaps_seen["refused"] >= 1, swaps_seen
    assert (quirk_visible >= 1) == (swap_rule == 1)     # rule 0 stores logL + logPrior; rule 1 keeps B's old prior in the sum
    s.close()


def _one_langevin_iteration_against_oracle(oracle, star, y, T, s, init_logL, learn, c0, fd_step_rel, delta, report, check=None):
    """One settled Langevin iteration of the product against oracle/sampler_oracle.c::orc_langevin_iteration fed with the same draws.  The
    oracle recomputes every gradient from the position (term-by-term long double differences) and evaluates both proposal densities as
    full multivariate-normal densities by pivoted elimination -- the product carries gradients along, re-tempers them on swaps and uses
    two triangular solves with its Cholesky factor.  Stated tolerances:
      * the proposal x' = x + drift + L z agrees with the oracle's own to 2e-6 of the step's length (measured on the MI355X: ~1e-9 outside the
        adaptation window, up to 9e-7 inside a violent one -- gain 0.3, drift as long as the random part; the product's finite-difference
        gradient is within 4e-10 of the oracle's on the gradient's scale at the star's parameters, single components of hot chains 1e-5);
      * tested AT the product's proposal (see the comment at prop_given below): the positions after the test are identical, log-posteriors
        agree to 1e-9, move probabilities to 1e-2 relative / 1e-5 absolute (hot chains multiply the gradient's ~1e-6 accuracy by whitened
        drifts of ~10 and steps of ~10; cold chains agree to ~1e-6), the adapted law to rounding of the shared inputs.
    check: the chains the oracle advances and the comparison covers (None = all; the swap pair is always among them)."""
    nch = len(T)
    st = s.state()
    params = np.tile(star.params, (nch, 1))
    params[:, star.index_to_relax] = st["vars"]
    before = dict(params=params, vars=st["vars"], logL=st["logL"], logPrior=st["logPrior"], logPost=st["logPost"])
    law = s.proposal_law()
    it = st["iteration"]
    z, u, u_swap, ind_A = s.draws(it)
    mask = np.ones(nch, dtype=np.int32)
    if check is not None:
        mask[:] = 0
        mask[list(check)] = 1
        if it != 0:
            mask[[ind_A, ind_A + 1]] = 1
    c = np.flatnonzero(mask)
    smp, stt = s.run(1, stats=True)
    aft = s.state()
    assert aft["iteration"] == it + 1
    # The oracle makes its test AT the product's proposals (after comparing them with its own, below).  The model is truncated to windows
    # of whole bins, so the forward-difference gradient jumps where a step moves a window edge across a bin; two proposals that agree to
    # 1e-8 of a step sit on different sides of such a jump in ~1 % of the gradient evaluations at this shape (measured: tools/langevin_diag.py)
    # and then have drifts -- hence reverse densities -- that differ by O(1).  That is a property of finite differences of the reference's
    # truncated likelihood, not a disagreement about the step.
    prop_prod = s.last_test()[0]
    exp, law2, rc = oracle.sampler_iteration(star, y, T, init_logL, before, law, i=it, z=z, u_mh=u, learn=learn, do_swap=it != 0, ind_A=ind_A,
                                             u_swap=u_swap, c0=c0, use_drift=True, fd_step_rel=fd_step_rel, delta=delta, chain_mask=mask,
                                             prop_given=prop_prod)
    assert rc == 0
    step0 = np.linalg.norm(exp["prop_vars"][c] - before["vars"][c], axis=1)
    dprop = np.max(np.linalg.norm(prop_prod[c] - exp["prop_vars"][c], axis=1) / step0)      # x + drift + L z: product against oracle
    assert dprop < 2e-6, (it, dprop)
    exp["prop_vars"][c] = prop_prod[c]
    # the comparators were drawn by the product's generator, not chosen: a chain whose comparator lies within the tolerance of its move
    # probability may fall either way and is left out of this iteration's comparison (with its swap partner); every iteration starts from
    # the product's own state, so nothing carries over
    r_test, u_test = exp["Pmove"].copy(), u
    if exp["swapped"]:
        r_test[[ind_A, ind_A + 1]] = r_test[[ind_A + 1, ind_A]]             # Pmove travels with the rows (MALA.cpp:437, :447)
    knife = np.abs(r_test - u_test) < 5e-3 * r_test
    if it != 0 and (knife[ind_A] or knife[ind_A + 1]):
        knife[[ind_A, ind_A + 1]] = True
    c = np.array([m for m in c if not knife[m]], dtype=int)
    assert c.size >= max(1, mask.sum() - 3)
    # the step each chain took (zero for a refused move on both sides), in units of the oracle's proposed step
    step = np.linalg.norm(exp["prop_vars"][c] - before["vars"][c], axis=1)
    dv = np.max(np.linalg.norm(aft["vars"][c] - exp["vars"][c], axis=1) / step)
    dP = np.max(np.abs(aft["Pmove"][c] - exp["Pmove"][c]) / np.maximum(exp["Pmove"][c], 1e-3))
    dL = np.max(np.abs(aft["logPost"][c] - exp["logPost"][c]) / np.abs(exp["logPost"][c]))
    report.append((it, learn, int(exp["moved"][c].sum()), exp["swapped"], dprop, dL, dP))
    assert dv < 1e-12, (it, dv)     # (tested at the same proposal: the positions after the test are the same numbers)
    assert np.array_equal(smp[0], aft["vars"])
    assert np.allclose(aft["logL"][c], exp["logL"][c], rtol=1e-9, atol=0) and np.allclose(aft["logPost"][c], exp["logPost"][c], rtol=1e-9, atol=0)
    assert np.allclose(aft["logPrior"][c], exp["logPrior"][c], rtol=1e-9, atol=1e-9)
    assert np.allclose(stt[0][c, 0], exp["logL"][c], rtol=1e-9) and np.allclose(stt[0][c, 2], exp["logPost"][c], rtol=1e-9)
    assert np.allclose(aft["Pmove"][c], exp["Pmove"][c], rtol=1e-2, atol=1e-5), (it, dP)
    if learn:
        mu, cov, sig = s.proposal_law()
        assert np.allclose(mu[c], law2[0][c], rtol=1e-10, atol=1e-12)
        assert np.allclose(sig[c], law2[2][c], rtol=0, atol=1e-2 * c0 / (1 + it) + 1e-12)          # sigma moves by gamma (Pmove - target)
        assert np.allclose(cov[c], law2[1][c], rtol=1e-8, atol=1e-8 * np.abs(law2[1][c]).max())
    return exp, ind_A


def _langevin_walk(pkg, oracle, star, y, ctx, nch, lam, engine, seed, fd_step_rel, delta, adapt_to, n_settle, want_swaps, check=None):
    """The checked walk: from the start point (no adaptation, iteration 0 without a swap step), two adapting iterations, the rest of the
    adaptation window unchecked, then settled iterations whose swap pair / comparator are chosen so that accepted and refused swaps (for
    the two-group layouts: inside a group and across the groups) are among them -- each FOLLOWED by a second checked iteration, whose
    drifts come from the gradients the product carried through the swap (likelihood share re-tempered) where the oracle recomputes them
    at the new temperatures."""
    c0 = 2.0
    T = lam ** np.arange(nch)
    s = pkg.Sampler(ctx, star, nchains=nch, lambda_temp=lam, engine=engine, use_drift=1, seed=seed, Nt_learn=(20, adapt_to), periods_learn=(1,),
                    dN_mixing=1, c0=c0, fd_step_rel=fd_step_rel, delta=delta)
    init_logL = s.state()["logL"].copy()
    rep, moved, refused, swapped, kept, visible = [], 0, 0, 0, 0, 0
    one = lambda learn: _one_langevin_iteration_against_oracle(oracle, star, y, T, s, init_logL, learn, c0, fd_step_rel, delta, rep, check)
    for _ in range(2):
        one(False)
    s.run(18, record=False)
    for _ in range(3):                      # iterations 20, 21, 22: gain c0/(1+i) ~ 0.1; the second and third use an adapted law
        one(True)
    s.run(adapt_to - 23, record=False)
    assert s.state()["iteration"] == adapt_to
    s.run(n_settle, record=False)
    vars_now = s.state()["vars"]
    for want_A, (u_lo, u_hi) in want_swaps:
        k = _find_iteration(s, 5000, want_A, u_lo, u_hi)
        s.set_state(vars_now, iteration=k)
        for j in range(2):
            exp, ia = one(False)
            if j == 0:
                assert ia == want_A
                swapped += int(exp["swapped"])
                kept += int(not exp["swapped"])
            moved += int(exp["moved"].sum())
            refused += int((exp["moved"] == 0).sum())
            visible += int(np.any((exp["Pmove"] > 1e-6) & (exp["Pmove"] < 1)))     # the correction term is visible in a move probability
        vars_now = s.state()["vars"]
    s.close()
    print("\nlangevin walk", engine, "nch", nch, "delta", delta, "\n     it learn moved swapped dvars dlogPost dPmove")
    for r in rep:
        print("  %6d %d %3d %d  %.2e %.2e %.2e" % r)
    # (the comparators are the product's own draws: a swap asked to fail with u in (0.97, 1) still happens when r_T = 1; the ten-chain walks
    #  hold refused swaps, the twenty-chain one is asked for three checked swap steps of either kind)
    assert moved >= 1 and refused >= 1 and swapped >= 1 and (kept >= 1 or nch > 10) and visible >= 2, (moved, refused, swapped, kept, visible)


@pytest.mark.parametrize("engine", ["host", "device"])
@pytest.mark.parametrize("delta", [0.0, 30.0])
def test_langevin_iteration_equals_the_oracle_on_a_local_slice(pkg, oracle, synth, engine, delta):
    """C2 family: local slice, 1e4 bins, 21 free variables, 10 chains; with and without truncation of the drift."""
    star = synth.make_c2_star(nx=10000)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    y = star.set_spectrum_from_model(m0, seed=11)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, y)
    _langevin_walk(pkg, oracle, star, y, ctx, 10, 1.7, engine, 5, 1e-7, delta, adapt_to=150, n_settle=20,
                   want_swaps=((2, (0.0, 0.05)), (4, (0.0, 0.05)), (4, (0.97, 1.0)), (7, (0.97, 1.0)), (1, (0.97, 1.0)), (8, (0.97, 1.0)),
                               (5, (0.985, 1.0))))   # (several pairs asked to refuse: a pair whose swap probability is 1 swaps whatever u)
    ctx.close()


@pytest.mark.parametrize("engine", ["host", "device"])
def test_langevin_iteration_equals_the_oracle_at_the_headline_shape(pkg, oracle, c3, engine):
    """The north star's named step at C3 size: 1e5 bins x 111 parameters (93 free) x 20 tempered chains."""
    star, ctx = c3
    _langevin_walk(pkg, oracle, star, star.y, ctx, NCH, LAM, engine, 31, 1e-7, 0.0, adapt_to=120, n_settle=10,
                   want_swaps=((3, (0.0, 0.01)), (NCH // 2 - 1, (0.0, 0.01)), (NCH // 2 - 1, (0.97, 1.0))), check=(0, 10, NCH - 1))


def test_langevin_at_the_headline_shape(pkg, oracle, c3):
    """North star's named path at C3 size on the DEVICE engine: the forward-difference gradient that drives the Langevin proposal equals the
    oracle's finite differences of the reference log-likelihood (same steps), and the sampler it drives accepts at a healthy rate after
    its adaptation window and moves every chain."""
    star, ctx = c3
    # (1) gradient of the tempered log-likelihood at the star's parameters, three temperatures: device FD batch vs orc_fd_gradient
    # steps of 1e-5 |theta| here (the sampler's default is 1e-7): the ORACLE's difference of two ~1e5-term sums carries ~1e-8 of rounding
    # noise, i.e. ~1e-2 of a typical gradient component at the 1e-7 step; the device forms its differences term by term and does not
    h = 1e-5 * np.maximum(np.abs(star.params[star.index_to_relax]), 1e-3)
    T = np.array([1.0, LAM ** 7, LAM ** 19])
    l0, g = ctx.fd_gradient(star.model_id, np.tile(star.params, (3, 1)), star.plength, star.index_to_relax, h, T)
    for j in (0, 2):
        st, l0o, go = oracle.fd_gradient(star.model_id, star.params, star.plength, star.index_to_relax, h, star.x, star.y, 1.0, T[j])
        assert st == 0 and abs(l0[j] - l0o) <= 2e-11 * abs(l0o)
        scale = np.maximum(np.abs(go), 1e-3 * np.abs(go).max())     # compare on the gradient's own scale
        assert np.max(np.abs(g[j] - go) / scale) < 1e-3, np.max(np.abs(g[j] - go) / scale)
    # (2) the sampler
    s = pkg.Sampler(ctx, star, nchains=NCH, lambda_temp=LAM, engine="device", use_drift=1, seed=31, Nt_learn=(30, 230), periods_learn=(1,), dN_mixing=1,
                    c0=2.0)
    s.run(230, record=False)
    smp, stt = s.run(150, stats=True)
    acc = np.mean(np.any(smp[1:] != smp[:-1], axis=2), axis=0)
    assert np.all(np.isfinite(stt)) and 0.1 < acc[0] < 0.9 and 0.1 < acc.mean() < 0.9, acc
    assert all((smp[1:, m] != smp[:-1, m]).any() for m in range(NCH))
    s.close()
