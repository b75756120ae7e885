"""GPU tests of the sampler that drives the hot path (-m gpu): host-driven engine (mirror of MALA.cpp) and the
device-resident engine consume the same counter-based random numbers, so their trajectories must coincide
until a knife-edge accept decision; statistical checks pin the sampler itself (the reference is unseedable,
MALA.cpp:62-63, so only statistical agreement is defined: SURVEY section 0, finding 7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _star_with_data(pkg, oracle, synth, nx=4000, seed=5):
    star = synth.make_c2_star(nx=nx)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    star.set_spectrum_from_model(m0, seed)
    return star


def _constrained_star(pkg, oracle, synth, seed=5, nx=11000):
    """The C2 local slice with its grid moved so that all six multiplets lie INSIDE the spectrum (2875 .. 3224 muHz; the stock grid starts
    at 2900 muHz, above two of them, and a 4000-bin cut keeps two in range: parameters of modes the data does not hold are as wide as
    their priors and mix over thousands of iterations).  Every parameter is constrained by the data: autocorrelation times of ~1e2
    iterations, Monte-Carlo errors of a few percent of a posterior sigma from ~1e5 samples."""
    star = synth.make_c2_star(nx=nx)
    star.x = 2875.0 + (star.x[1] - star.x[0]) * np.arange(nx)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    star.set_spectrum_from_model(m0, seed)
    return star


@pytest.fixture()
def ctx(pkg):
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    yield c
    c.close()


def test_initial_state_matches_oracle(pkg, oracle, synth, ctx):
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    s = pkg.Sampler(ctx, star, nchains=4, lambda_temp=1.7, Nt_learn=(10**9, 10**9 + 1), periods_learn=(1,))
    st = s.state()
    T = 1.7 ** np.arange(4)
    ref, _, _ = oracle.loglike_batch(star.model_id, np.tile(star.params, (4, 1)), star.plength, star.x, star.y, 1.0, T)
    assert np.allclose(st["logL"], ref, rtol=1e-11)
    assert np.all(np.isfinite(st["logPrior"])) and np.allclose(st["logPost"], st["logL"] + st["logPrior"])
    s.close()


def test_device_engine_follows_host_engine(pkg, oracle, synth, ctx):
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(nchains=6, lambda_temp=1.5, seed=11, Nt_learn=(10**9, 10**9 + 1), periods_learn=(1,), dN_mixing=1)
    h = pkg.Sampler(ctx, star, engine="host", **kw)
    d = pkg.Sampler(ctx, star, engine="device", **kw)
    n = 120
    sh, th = h.run(n, stats=True)
    sd, td = d.run(n, stats=True)
    # identical random numbers and algorithm: trajectories agree to rounding until a knife-edge decision
    same = np.all(np.isclose(sh, sd, rtol=1e-9, atol=1e-12), axis=(1, 2))
    first_div = n if same.all() else int(np.argmin(same))
    assert first_div >= 60, f"engines diverge at iteration {first_div}"
    assert np.allclose(th[:first_div], td[:first_div], rtol=1e-9, atol=1e-7)
    a, b = h.state(), d.state()
    assert a["iteration"] == b["iteration"] == n
    assert abs(a["swaps"] - b["swaps"]) <= 6 and a["swap_attempts"] == b["swap_attempts"] == n - 1
    assert (sh[:, 0] != sh[0, 0]).any()  # chain 0 moved
    h.close(); d.close()


@pytest.mark.parametrize("nchains,dN_mixing,learn", [(7, 1, None), (7, 3, None), (7, 7, (40, 90)), (7, 0, None), (7, 1, (5, 30)),
                                                     (12, 1, None), (20, 1, (40, 90)), (9, 2, None), (8, 1, None)])
def test_fused_steps_are_bitwise_the_lockstep_chain(pkg, oracle, synth, ctx, nchains, dN_mixing, learn):
    """Stretches without adaptation run one fused launch per iteration (k_step: likelihood tiles with the settle step in their tail +
    the branch-ahead candidates of the next iteration); the lockstep kernels (k_iterate, k_loglike) do the same work in sequence.
    Same random numbers, same arithmetic, same summation orders: samples, statistics, swap counts and the final state must be
    IDENTICAL, whatever the mixing period and wherever the adaptation window falls.  From 8 chains on the fused iteration is TWO
    launches, one per chain group on its own stream (the groups drift apart and meet again at every swap whose pair straddles them):
    still the same chains."""
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    Nt = learn if learn else (10**9, 10**9 + 1)
    kw = dict(nchains=nchains, lambda_temp=1.4 if nchains < 10 else 1.2, seed=23, Nt_learn=Nt, periods_learn=(2,), dN_mixing=dN_mixing)
    out = []
    for scheme in (1, 0 if nchains < 8 else 3):   # 3: two chain groups even on this small star (automatic: only when a launch outgrows the GPU)
        ctx.set_option(pkg.OPT_STEP_SCHEME, scheme)
        d = pkg.Sampler(ctx, star, engine="device", **kw)
        s1, t1 = d.run(150 if nchains < 8 else 700, stats=True)
        s2, t2 = d.run(61, stats=True)   # a second call continues the same chains
        s3, t3 = d.run(2, stats=True)    # shorter than a fused stretch: lockstep either way
        out.append((np.concatenate([s1, s2, s3]), np.concatenate([t1, t2, t3]), d.state()))
        d.close()
    ctx.set_option(pkg.OPT_STEP_SCHEME, 0)
    smp, st, state = out[1]
    assert np.array_equal(smp, out[0][0])
    assert np.array_equal(st, out[0][1])
    for k in ("iteration", "swaps", "swap_attempts", "accepted0"):
        assert state[k] == out[0][2][k], k
    for k in ("vars", "logL", "logPrior", "logPost", "Pmove", "sigma"):
        assert np.array_equal(state[k], out[0][2][k]), k
    assert (out[0][0][:, 0] != out[0][0][0, 0]).any()
    if dN_mixing:
        assert 0 < state["swaps"] <= state["swap_attempts"]


def test_records_written_into_pinned_buffers_equal_the_copied_ones(pkg, oracle, synth, ctx):
    """run(out=...) with page-locked arrays (pinned_empty / tamcmc_hip_host_alloc): the settle step writes every iteration's sample and
    statistics straight into the caller's memory; with ordinary arrays the records are kept on the device and copied at the end of the
    call.  Same records either way, through learning (lockstep kernels) and fused stretches, samples only / statistics only included.
    The two samplers share one context and take turns on it, with a batched evaluation in between: what a sampler carries from one call
    to the next (the prepared candidates of the next iteration) must not live in the context's scratch memory."""
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(nchains=9, lambda_temp=1.4, seed=4, Nt_learn=(20, 60), periods_learn=(1,), dN_mixing=1, engine="device")
    a, b = pkg.Sampler(ctx, star, **kw), pkg.Sampler(ctx, star, **kw)
    for n in (100, 37, 5):
        ps, pt = pkg.pinned_empty((n, 9, a.nvars)), pkg.pinned_empty((n, 9, 3))
        ps[:] = np.nan; pt[:] = np.nan
        a.run(n, out=(ps, pt))
        ctx.loglike_params_batch(star.model_id, star.params, star.plength)     # (anything else on the context between two calls:
        s, t = b.run(n, stats=True)                                            # the candidates a sampler carries over are its own)
        assert np.array_equal(ps, s) and np.array_equal(pt, t), n
    ps = pkg.pinned_empty((50, 9, a.nvars))
    a.run(50, out=(ps, None))
    s, _ = b.run(50)
    assert np.array_equal(ps, s)
    pt = pkg.pinned_empty((50, 9, 3))
    a.run(50, out=(None, pt))
    _, t = b.run(50, record=False, stats=True)
    assert np.array_equal(pt, t)
    assert np.array_equal(a.state()["vars"], b.state()["vars"])
    a.close(); b.close()


def test_device_engine_learning_adapts(pkg, oracle, synth, ctx):
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(nchains=4, lambda_temp=1.7, seed=3, Nt_learn=(50, 450), periods_learn=(1,), c0=5.0)
    out = {}
    for eng in ("host", "device"):
        s = pkg.Sampler(ctx, star, engine=eng, **kw)
        s.run(50, record=False)
        mu0, cov0 = s.get_proposal(0)
        s.run(400, record=False)
        mu1, cov1 = s.get_proposal(0)
        st = s.state()
        assert not np.allclose(cov0, cov1) and np.all(np.isfinite(cov1))
        assert np.all(np.linalg.eigvalsh((cov1 + cov1.T) / 2) > -1e-12)
        assert 0 < st["sigma"][0] < 10
        smp, _ = s.run(300)
        acc = np.mean(np.any(smp[1:, 0] != smp[:-1, 0], axis=1))
        out[eng] = (st["sigma"].copy(), acc)
        assert 0.05 < acc < 0.8, (eng, acc)   # adapted towards the 0.234 target
        s.close()
    # same algorithm on both engines: adapted scales agree within a factor of a few
    assert np.all(out["host"][0] / out["device"][0] < 4) and np.all(out["device"][0] / out["host"][0] < 4)


def test_posterior_recovers_truth(pkg, oracle, synth, ctx):
    """Posterior summary statistics (mean / sigma per variable, tools/bin2txt_params.cpp:165-168) of the coldest chain
    bracket the true parameters of the synthetic star; the two engines (run with different seeds, so that their chains are independent)
    agree within MONTE-CARLO error: difference of the means / variances in units of sigma / sqrt(ESS) (tests/mc_stats.py)."""
    import mc_stats
    star = _constrained_star(pkg, oracle, synth, seed=9)
    ctx.set_spectrum(star.x, star.y)
    truth = star.params[star.index_to_relax]
    res = {}
    for eng, seed, n in (("device", 21, 60000), ("host", 22, 15000)):
        s = pkg.Sampler(ctx, star, engine=eng, nchains=5, lambda_temp=1.6, seed=seed, Nt_learn=(200, 3200), periods_learn=(1,), c0=5.0)
        s.run(3200, record=False)
        smp, _ = s.run(n)
        res[eng] = smp[:, 0, :].copy()
        s.close()
    mean, std = res["device"].mean(0), res["device"].std(0)
    fidx = [i for i, k in enumerate(star.index_to_relax) if star.names[k] == "Frequency_l"]
    # frequencies are the best-constrained parameters: truth within 5 posterior sigma, sigma of a sane size
    z = np.abs(mean[fidx] - truth[fidx]) / std[fidx]
    assert np.all(z < 5), z
    assert np.all(std[fidx] < 3.0) and np.all(std[fidx] > 1e-4)
    zm, zv, ea, eb = mc_stats.compare_chains(res["device"], res["host"])
    print("\ndevice vs host engine: ESS min %.0f / %.0f, max |z_mean| %.2f, max |z_var| %.2f" % (ea.min(), eb.min(), np.abs(zm).max(), np.abs(zv).max()))
    assert ea.min() > 60 and eb.min() > 20, (ea.min(), eb.min())
    assert np.all(np.abs(zm) < 4) and np.all(np.abs(zv) < 4.5), (np.abs(zm).max(), np.abs(zv).max())


def test_langevin_drift_sampler(pkg, oracle, synth, ctx):
    """use_drift=1 (the path the reference leaves as stubs, MALA.cpp:321-337,496-500): forward-difference gradient from the
    device FD batches, preconditioned drift, asymmetric-proposal correction, on the HOST engine.  It must sample the same posterior as
    the random-walk engine: means and variances of every variable within Monte-Carlo error (sigma / sqrt(ESS)), and accept at a healthy
    rate."""
    import mc_stats
    star = _constrained_star(pkg, oracle, synth, seed=5)
    ctx.set_spectrum(star.x, star.y)
    res = {}
    for drift, n in ((0, 15000), (1, 6000)):
        s = pkg.Sampler(ctx, star, engine="host", use_drift=drift, nchains=4, lambda_temp=1.6, seed=13 + drift, Nt_learn=(100, 2100),
                        periods_learn=(1,), c0=5.0)
        s.run(2100, record=False)
        smp, stt = s.run(n, stats=True)
        cold = smp[:, 0, :]
        acc = np.mean(np.any(cold[1:] != cold[:-1], axis=1))
        assert np.all(np.isfinite(stt))
        assert 0.05 < acc < 0.95, (drift, acc)
        res[drift] = cold.copy()
        s.close()
    zm, zv, ea, eb = mc_stats.compare_chains(res[0], res[1])
    print("\nhost engine, MH vs Langevin: ESS min %.0f / %.0f, max |z_mean| %.2f, max |z_var| %.2f" % (ea.min(), eb.min(), np.abs(zm).max(), np.abs(zv).max()))
    assert ea.min() > 15 and eb.min() > 15, (ea.min(), eb.min())
    assert np.all(np.abs(zm) < 4) and np.all(np.abs(zv) < 4.5), (np.abs(zm).max(), np.abs(zv).max())


def test_langevin_and_random_walk_sample_the_same_posterior_at_monte_carlo_resolution(pkg, oracle, synth, ctx):
    """Detailed balance, checked where it shows: long runs of the DEVICE engine with and without the Langevin drift on the same star.
    The random-walk chain is the reference's algorithm (a10); a Langevin step with a wrong correction term q(x|x') / q(x'|x) still
    moves and still accepts, but samples a shifted law.  Every variable's mean and variance of the coldest chain must agree within
    4 / 4.5 combined Monte-Carlo errors sigma / sqrt(ESS), i.e. within a fraction of a posterior sigma (printed; below 0.12 sigma per unit,
    asserted) -- against the 4 posterior sigma of a window that any sampler that moves would pass.  The same statistic separates the coldest chain from its warmer neighbour (T = 1.6) by tens of units: the check
    has the resolution it claims."""
    import mc_stats
    star = _constrained_star(pkg, oracle, synth, seed=5)
    ctx.set_spectrum(star.x, star.y)
    res = {}
    for drift, n in ((0, 200000), (1, 50000)):
        s = pkg.Sampler(ctx, star, engine="device", use_drift=drift, nchains=4, lambda_temp=1.6, seed=91 + drift, Nt_learn=(100, 4100),
                        periods_learn=(1,), c0=5.0)
        s.run(4100, record=False)
        smp, _ = s.run(n)
        res[drift] = smp
        cold = smp[:, 0, :]
        acc = np.mean(np.any(cold[1:] != cold[:-1], axis=1))
        assert 0.1 < acc < 0.8, (drift, acc)
        s.close()
    zm, zv, ea, eb = mc_stats.compare_chains(res[0][:, 0, :], res[1][:, 0, :])
    print("\nMH vs Langevin: ESS min %.0f / %.0f, max |z_mean| %.2f, max |z_var| %.2f" % (ea.min(), eb.min(), np.abs(zm).max(), np.abs(zv).max()))
    assert ea.min() > 200 and eb.min() > 80, (ea.min(), eb.min())
    sd = res[0][:, 0, :].std(0)
    mc = np.sqrt(sd ** 2 / ea + res[1][:, 0, :].var(0) / eb)
    print("Monte-Carlo error of the difference of means: %.3f .. %.3f posterior sigma" % ((mc / sd).min(), (mc / sd).max()))
    assert np.all(mc < 0.12 * sd)                                   # the resolution: a few percent of a posterior sigma (a 4-sigma window: 400 %)
    assert np.all(np.abs(zm) < 4) and np.all(np.abs(zv) < 4.5), (zm, zv)
    # power: the chain one rung up the ladder samples L^(1/1.6) x prior -- the same statistic tells it from the coldest chain
    zm1, zv1, _, _ = mc_stats.compare_chains(res[0][:, 0, :], res[0][:, 1, :])
    assert np.abs(zv1).max() > 8, np.abs(zv1).max()


def test_device_langevin_engine_follows_the_host_engine(pkg, oracle, synth, ctx):
    """use_drift = 1 on the device-resident engine (k_mala_settle / finite-difference batch / k_mala_test, dev_mala_impl.h): same Philox
    streams and the same algorithm as the host engine (host_mala.cpp) -> the chains coincide to rounding, through the adaptation window
    and the swaps, until a knife-edge decision; afterwards both still sample (acceptance in a sane range)."""
    star = _star_with_data(pkg, oracle, synth, nx=4000, seed=5)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(use_drift=1, nchains=5, lambda_temp=1.5, seed=21, Nt_learn=(20, 60), periods_learn=(1,), c0=3.0, dN_mixing=1)
    h = pkg.Sampler(ctx, star, engine="host", **kw)
    d = pkg.Sampler(ctx, star, engine="device", **kw)
    n = 90
    sh, th = h.run(n, stats=True)
    sd1, td1 = d.run(50, stats=True)
    sd2, td2 = d.run(n - 50, stats=True)             # a second call continues the same chains (gradients stay valid)
    sd, td = np.concatenate([sd1, sd2]), np.concatenate([td1, td2])
    # (the forward differences divide ~1e-12 rounding differences of the two engines' arithmetic by steps of ~1e-7 |theta|: the gradients,
    # hence the proposals, agree to ~1e-6 relative, and the adaptation feeds that back)
    dev = np.max(np.abs(sh - sd) / (np.abs(sh) + 1e-3), axis=(1, 2))
    same = dev < 1e-4
    first_div = n if same.all() else int(np.argmin(same))
    assert first_div >= 40, f"engines diverge at iteration {first_div}: {dev[max(first_div - 3, 0):first_div + 2]}"
    assert np.allclose(th[:first_div], td[:first_div], rtol=1e-5, atol=1e-3)
    a, b = h.state(), d.state()
    assert a["iteration"] == b["iteration"] == n and a["swap_attempts"] == b["swap_attempts"] == n - 1
    if first_div == n:
        mh, ch = h.get_proposal(1)
        md, cd = d.get_proposal(1)
        assert np.allclose(mh, md, rtol=1e-4) and np.allclose(ch, cd, rtol=1e-3, atol=1e-6 * np.abs(ch).max())
    smp, _ = d.run(400)
    acc = np.mean(np.any(smp[1:, 0] != smp[:-1, 0], axis=1))
    assert 0.05 < acc < 0.98, acc
    h.close(); d.close()


def test_forty_chains_device_engine(pkg, oracle, synth, ctx):
    """BASELINE config 5 asks for 40 tempered chains; the reference caps Nchains at 24 (MALA.cpp:580-587), this build at 64."""
    star = _star_with_data(pkg, oracle, synth, nx=2048, seed=2)
    ctx.set_spectrum(star.x, star.y)
    s = pkg.Sampler(ctx, star, engine="device", nchains=40, lambda_temp=1.15, seed=4, Nt_learn=(10**9, 10**9 + 1), periods_learn=(1,))
    smp, stt = s.run(150, stats=True)
    st = s.state()
    assert smp.shape == (150, 40, star.nvars) and np.all(np.isfinite(stt))
    assert st["swap_attempts"] == 149 and st["swaps"] > 10
    moved = [(smp[1:, m] != smp[:-1, m]).any() for m in range(40)]
    assert all(moved)
    s.close()


@pytest.mark.parametrize("engine", ["host", "device"])
def test_checkpoint_and_resume(pkg, oracle, synth, ctx, tmp_path, engine):
    """write_restore after 120 iterations (adaptation running), a NEW sampler reads the files and continues: it must follow the
    uninterrupted run (random numbers are addressed by (seed, chain, iteration); a swapped chain's re-tempered logL is
    recomputed from scratch on restart, so agreement is to rounding until a knife-edge decision)."""
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(nchains=5, lambda_temp=1.6, seed=17, Nt_learn=(30, 400), periods_learn=(1,), dN_mixing=2, engine=engine)
    a = pkg.Sampler(ctx, star, **kw)
    a.run(120, record=False)
    root = str(tmp_path / "restore_")
    a.write_restore(root, [star.names[i] for i in star.index_to_relax])
    sa, ta = a.run(80, stats=True)
    b = pkg.Sampler(ctx, star, **kw)
    b.read_restore(root)
    st = b.state()
    assert st["iteration"] == 120
    mu_a, cov_a = a.get_proposal(1)
    sb, tb = b.run(80, stats=True)
    same = np.all(np.isclose(sa, sb, rtol=1e-9, atol=1e-12), axis=(1, 2))
    first_div = 80 if same.all() else int(np.argmin(same))
    assert first_div >= 40, f"resumed run diverges at iteration {first_div}"
    assert np.allclose(ta[:first_div], tb[:first_div], rtol=1e-9, atol=1e-7)
    assert b.state()["iteration"] == 200
    # restoring only the proposal law keeps the fresh start point and iteration 0
    c = pkg.Sampler(ctx, star, **kw)
    v0 = c.state()["vars"].copy()
    c.read_restore(root, variables=False, proposal=True, last_index=False)
    assert c.state()["iteration"] == 0 and np.array_equal(c.state()["vars"], v0)
    mu_r, cov_r = c.get_proposal(1)
    mu_b, cov_b = pkg.Sampler(ctx, star, **kw).get_proposal(1)
    assert not np.allclose(cov_r, cov_b)     # the adapted covariance replaced the initial diagonal one
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("engine,groups", [("device", 1), ("device", 2), ("host", 0)])
def test_packed_stars_reproduce_their_solo_runs(pkg, oracle, synth, engine, groups):
    """Several stars co-resident on one GPU (tamcmc_sampler_run_packed: one context and one host thread per star).  The stars share
    nothing, so each one's samples and statistics must be bit-identical to a run on its own -- with adaptation running, for both
    engines, and with the chains of a star in one or two stream groups."""
    from tamcmc_c_amd import sampler as S
    stars = []
    for k, (nx, seed) in enumerate(((4000, 5), (3000, 6), (5000, 7))):
        st = _star_with_data(pkg, oracle, synth, nx=nx, seed=seed)
        stars.append(st)

    def build():
        cs, ss = [], []
        for k, st in enumerate(stars):
            c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
            c.set_spectrum(st.x, st.y)
            cs.append(c)
            ss.append(pkg.Sampler(c, st, engine=engine, nchains=4 + 2 * k, lambda_temp=1.5, seed=40 + k, Nt_learn=(20, 160), periods_learn=(1,),
                                  chain_groups=groups))
        return cs, ss

    n = 240
    cs, ss = build()
    solo = [s.run(n, stats=True) for s in ss]
    for s in ss:
        s.close()
    for c in cs:
        c.close()
    cs, ss = build()
    smp, stt = S.run_packed(ss, n, stats=True)
    for k in range(len(stars)):
        assert np.array_equal(smp[k], solo[k][0]) and np.array_equal(stt[k], solo[k][1]), k
        assert ss[k].state()["iteration"] == n and (smp[k][1:, 0] != smp[k][:-1, 0]).any()
    with pytest.raises(pkg.TamcmcError):
        S.run_packed([ss[0], ss[0]], 1)                                          # one sampler twice
    twin = pkg.Sampler(cs[0], stars[0], engine=engine, nchains=4, lambda_temp=1.5, seed=1, Nt_learn=(20, 160), periods_learn=(1,))
    with pytest.raises(pkg.TamcmcError):
        S.run_packed([ss[0], twin], 1)                                           # two samplers on one context
    twin.close()
    for s in ss:
        s.close()
    for c in cs:
        c.close()


@pytest.mark.parametrize("nchains,nx,dn,wg,c0", [(1, 700, 1, 64, 2.0), (2, 513, 1, 256, 2.0), (3, 100, 2, 64, 2.0), (7, 2049, 1, 64, 2.0),
                                                 (64, 1500, 1, 64, 2.0), (4, 900, 1, 64, 10.0)])
def test_engine_matrix_of_awkward_shapes(pkg, oracle, synth, nchains, nx, dn, wg, c0):
    """Shapes off the beaten path: a single chain (no swap partner), two chains (the pair IS the ladder), odd counts split over two
    stream groups, the chain cap; spectra shorter than one tile, one bin past a tile boundary, both likelihood-kernel geometries;
    adaptation from iteration 5 on -- with c0 = 10 the first updates have gamma = c0/(1+i) > 1 and the covariance stops being
    positive definite (both engines then keep the previous Cholesky factor).  The device-resident engine must follow the
    host-driven one (same Philox streams, same adaptation) and leave the same counters and move probabilities."""
    star = _star_with_data(pkg, oracle, synth, nx=nx, seed=nx)
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST, workgroup=wg)
    c.set_spectrum(star.x, star.y)
    kw = dict(nchains=nchains, lambda_temp=1.2 if nchains > 10 else 1.6, seed=3 + nchains, Nt_learn=(5, 40), periods_learn=(1,), dN_mixing=dn, c0=c0)
    h = pkg.Sampler(c, star, engine="host", **kw)
    d = pkg.Sampler(c, star, engine="device", **kw)
    n = 90
    sh, th = h.run(n, stats=True)
    sd, td = d.run(n, stats=True)
    assert np.isfinite(td).all() and np.isfinite(sd).all()
    same = np.all(np.isclose(sh, sd, rtol=1e-8, atol=1e-11), axis=(1, 2))
    first_div = n if same.all() else int(np.argmin(same))
    # (nearly singular covariance matrices decide "positive definite or not" on the last bits: no trajectory identity asked of c0 = 10)
    assert first_div >= (30 if c0 < 5 else 6), f"engines diverge at iteration {first_div}"
    a, b = h.state(), d.state()
    want_attempts = 0 if nchains == 1 else len([i for i in range(1, n) if i % dn == 0])
    assert a["iteration"] == b["iteration"] == n and a["swap_attempts"] == b["swap_attempts"] == want_attempts
    assert b["swaps"] <= want_attempts and (sd[:, 0] != sd[0, 0]).any()
    if first_div == n:
        assert a["swaps"] == b["swaps"] and a["accepted0"] == b["accepted0"]
        assert np.allclose(a["Pmove"], b["Pmove"], rtol=1e-6, atol=1e-9) and np.allclose(a["sigma"], b["sigma"], rtol=1e-7)
    h.close(); d.close(); c.close()


@pytest.mark.parametrize("n_free", [3, 7, 8, 9, 15, 16, 17, 21, 24, 25, 31, 32, 33, 47])
def test_learning_factor_at_every_panel_remainder(pkg, oracle, synth, n_free):
    """The device engine factors (Sigma + eps)sigma in panels of 8 columns with the next panel's diagonal block taken ahead; the host engine
    factors column by column.  Same operations per element, so with adaptation in every iteration the two engines must stay on one
    trajectory whatever the number of free variables leaves after the last full panel (0, 1, 7 columns; fewer than one panel; one
    panel exactly)."""
    if n_free <= 21:
        star = _star_with_data(pkg, oracle, synth, nx=1500, seed=n_free)
    else:
        star = synth.make_c3_star(nx=3000, step=0.7)
        _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
        star.set_spectrum_from_model(m0, n_free)
    for i in star.index_to_relax[n_free:]:  # freeze the rest (prior switch 0 = Fix)
        star.relax[i] = 0
        star.priors_switch[i] = 0
        star.priors[:, i] = -9999.0
    assert star.nvars == n_free
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    c.set_spectrum(star.x, star.y)
    kw = dict(nchains=3, lambda_temp=1.5, seed=40 + n_free, Nt_learn=(4, 10**6), periods_learn=(1,), c0=2.0)
    h = pkg.Sampler(c, star, engine="host", **kw)
    d = pkg.Sampler(c, star, engine="device", **kw)
    n = 60
    sh, _ = h.run(n, stats=True)
    sd, _ = d.run(n, stats=True)
    same = np.all(np.isclose(sh, sd, rtol=1e-8, atol=1e-11), axis=(1, 2))
    first_div = n if same.all() else int(np.argmin(same))
    assert first_div >= 40, f"engines diverge at iteration {first_div}"
    if first_div == n:
        for m in range(3):
            (mh, ch), (md, cd) = h.get_proposal(m), d.get_proposal(m)
            assert np.allclose(mh, md, rtol=1e-9, atol=1e-12) and np.allclose(ch, cd, rtol=1e-7, atol=1e-14)
    assert (sd[1:, 0] != sd[:-1, 0]).any()  # the adapted proposal moves the chain
    h.close(); d.close(); c.close()


def test_context_and_samplers_can_go_in_any_order(pkg, oracle, synth):
    """A sampler borrows its context (stream, device buffers).  tamcmc_hip_destroy with samplers still alive only marks the context;
    the last tamcmc_sampler_destroy releases it -- so garbage collection in arbitrary order cannot touch freed memory."""
    star = _star_with_data(pkg, oracle, synth, nx=600, seed=1)
    for engine in ("device", "host"):
        c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
        c.set_spectrum(star.x, star.y)
        s1 = pkg.Sampler(c, star, engine=engine, nchains=3, lambda_temp=1.5, Nt_learn=(10**9, 10**9 + 1), periods_learn=(1,))
        s2 = pkg.Sampler(c, star, engine=engine, nchains=2, lambda_temp=1.5, Nt_learn=(10**9, 10**9 + 1), periods_learn=(1,))
        s1.run(5, record=False)
        L = c._L
        hc, h1, h2 = c._h, s1._h, s2._h
        c._h = None; s1._h = None; s2._h = None      # take the handles: the Python layer's own ordering is out of the way
        L.tamcmc_hip_destroy(hc)                      # context first ...
        L.tamcmc_sampler_destroy(h1)                  # ... its samplers afterwards
        L.tamcmc_sampler_destroy(h2)


@pytest.mark.parametrize("engine", ["host", "device"])
def test_move_counts_are_the_moved_flags_the_reference_buffers(pkg, oracle, synth, ctx, tmp_path, engine):
    """tamcmc_sampler_get_move_counts: per chain, the iterations whose record carries moved = 1 (MALA.cpp:543-545; what the reference's
    acceptance diagnostic counts per buffer, outputs.cpp:1824-1858).  Without swaps a chain's flag says its position changed in that
    iteration, so the counters must equal the changes in the recorded samples -- through a learning window (lockstep kernels), quiet
    stretches (fused launches) and a second call; with swaps the flags travel with the rows (MALA.cpp:436, :446) and both engines,
    fed by the same random streams, must count the same.  The buffer's rates go to the reference's acceptance file."""
    from tamcmc_c_amd import sampler as S
    star = _star_with_data(pkg, oracle, synth)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(nchains=9, lambda_temp=1.4, seed=8, Nt_learn=(30, 80), periods_learn=(1,), c0=3.0)
    s = pkg.Sampler(ctx, star, engine=engine, dN_mixing=0, **kw)
    assert np.array_equal(s.move_counts(), np.zeros(9))
    smp1, _ = s.run(150)
    c1 = s.move_counts()
    smp2, _ = s.run(70)
    c2 = s.move_counts()
    start = star.params[star.index_to_relax]
    chg1 = np.any(np.concatenate([np.tile(start, (1, 9, 1)), smp1])[1:] != np.concatenate([np.tile(start, (1, 9, 1)), smp1])[:-1], axis=2).sum(0)
    chg2 = np.any(np.concatenate([smp1[-1:], smp2])[1:] != np.concatenate([smp1[-1:], smp2])[:-1], axis=2).sum(0)
    assert np.array_equal(c1, chg1) and np.array_equal(c2 - c1, chg2), (c1, chg1, c2 - c1, chg2)
    assert c2[0] == s.state()["accepted0"] and 0 < c2[0] < 220
    out = str(tmp_path / "acceptance.txt")
    S.write_acceptance(out, 0.5 * 150, c1 / 150.0, first=True)                    # buffer 0: x = (Ncopy + 0.5) Nbuffer
    S.write_acceptance(out, 150 + 0.5 * 70, (c2 - c1) / 70.0, first=False)
    x, r = S.read_acceptance(out)
    assert np.array_equal(x, [75.0, 185.0]) and np.allclose(r[0], c1 / 150.0, rtol=1e-5) and np.allclose(r[1], (c2 - c1) / 70.0, rtol=1e-5)
    s.close()
    # with swaps: the two engines agree on the flags (identical trajectories over this stretch)
    h = pkg.Sampler(ctx, star, engine="host", dN_mixing=1, **kw)
    d = pkg.Sampler(ctx, star, engine=engine, dN_mixing=1, **kw)
    sh, _ = h.run(60)
    sd, _ = d.run(60)
    if np.allclose(sh, sd, rtol=1e-9, atol=1e-12):
        assert np.array_equal(h.move_counts(), d.move_counts())
    assert d.state()["swaps"] > 0 and d.move_counts().sum() > 0
    h.close(); d.close()
