"""The sampler beyond its on-chip size limits (-m gpu).  The reference factors a proposal covariance of any size with Eigen's LLT
(MALA.cpp:339-369) and lists ~250 free variables for its largest fits; the device-resident engine keeps the adaptation's work
matrix in LDS up to ~135 free variables and the fused iteration's candidate roles in the tile's LDS up to Nparams + 2 Nvars = 971
(include/tamcmc_sampler.h, size-limit table).  Beyond either limit another branch runs -- global-memory scratch for the covariance
update and the Cholesky factor (dev_sampler.hip adapt_chain / k_mala_test), lockstep kernels instead of the fused step -- and it must
compute the same thing: against the sampler-step oracle, against the host-driven engine and against the forced lockstep scheme."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _big_star(pkg, oracle, synth, nmax, nx, step, seed):
    """C3 family (model_MS_Global_aj_HarveyLike), nmax radial orders x l <= 3: 6 nmax + 27 parameters, 6 nmax + 9 free."""
    star = synth.make_c3_star(seed=seed, nx=nx, nmax=nmax, lmax=3, step=step, fmin=15 * 135.1 - 80.0)
    _, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    star.set_spectrum_from_model(m0, seed + 1)
    return star


@pytest.fixture(scope="module")
def star165(pkg, oracle, synth):      # 24 orders: 171 parameters, 153 free
    return _big_star(pkg, oracle, synth, nmax=24, nx=30000, step=0.12, seed=7)


@pytest.fixture(scope="module")
def star_long(pkg, oracle, synth):    # 54 orders: 351 parameters, 333 free -- past the fused step's limit as well (351 + 666 > 971)
    return _big_star(pkg, oracle, synth, nmax=54, nx=40000, step=0.19, seed=9)


def test_the_sizes_are_beyond_the_limits(pkg, star165, star_long):
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    for star, fused in ((star165, 1), (star_long, 0)):
        c.set_spectrum(star.x, star.y)
        s = pkg.Sampler(c, star, engine="device", nchains=3, lambda_temp=1.3)
        info = s.info()
        assert info["nvars"] == star.nvars >= 140 and info["adapt_in_lds"] == 0 and info["fused_available"] == fused, info
        assert (info["nparams"] + 2 * info["nvars"] <= 971) == bool(fused)
        s.close()
    c.close()


def test_scratch_cholesky_follows_the_host_engine(pkg, star165, star_long):
    """Adaptation after every test with the work matrix in device memory (Nvars = 153 and 333): the blocked, look-ahead factorisation of
    the device engine against the host engine's column Cholesky -- same operations per element, so the engines stay on one trajectory
    and end with the same proposal law (as tests/test_gpu_sampler.py::test_learning_factor_at_every_panel_remainder, which covers the
    LDS path up to 93)."""
    for star, n in ((star165, 50), (star_long, 30)):
        c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
        c.set_spectrum(star.x, star.y)
        kw = dict(nchains=3, lambda_temp=1.4, seed=40 + star.nvars, Nt_learn=(4, 10**6), periods_learn=(1,), c0=2.0)
        h = pkg.Sampler(c, star, engine="host", **kw)
        d = pkg.Sampler(c, star, engine="device", **kw)
        assert d.info()["adapt_in_lds"] == 0
        sh, _ = h.run(n, stats=True)
        sd, _ = d.run(n, stats=True)
        same = np.all(np.isclose(sh, sd, rtol=1e-8, atol=1e-11), axis=(1, 2))
        first_div = n if same.all() else int(np.argmin(same))
        assert first_div >= (2 * n) // 3, f"engines diverge at iteration {first_div} (Nvars {star.nvars})"
        if first_div == n:
            for m in range(3):
                (mh, ch), (md, cd) = h.get_proposal(m), d.get_proposal(m)
                assert np.allclose(mh, md, rtol=1e-9, atol=1e-12) and np.allclose(ch, cd, rtol=1e-7, atol=1e-14 + 1e-9 * np.abs(ch).max())
        assert (sd[1:, 0] != sd[:-1, 0]).any()
        assert d.info()["iter_lockstep"] >= n - 4
        h.close(); d.close(); c.close()


@pytest.mark.parametrize("engine", ["host", "device"])
def test_one_iteration_equals_the_oracle_beyond_the_lds_limit(pkg, oracle, star165, engine):
    """One iteration against oracle/sampler_oracle.c (MALA.cpp:645-703) with 153 free variables and 8 chains (two groups): before, during
    and after the adaptation window -- the proposal x + L z with the factor the scratch branch produced, the Robbins-Monro update itself."""
    star = star165
    nch, lam, c0 = 8, 1.25, 2.0
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    c.set_spectrum(star.x, star.y)
    s = pkg.Sampler(c, star, nchains=nch, lambda_temp=lam, engine=engine, seed=5, Nt_learn=(3, 30), periods_learn=(1,), dN_mixing=1, c0=c0)
    T = lam ** np.arange(nch)
    init_logL = s.state()["logL"].copy()

    def one(learn):
        st = s.state()
        params = np.tile(star.params, (nch, 1))
        params[:, star.index_to_relax] = st["vars"]
        before = dict(params=params, vars=st["vars"], logL=st["logL"], logPrior=st["logPrior"], logPost=st["logPost"])
        law = s.proposal_law()
        it = st["iteration"]
        z, u, us, ia = s.draws(it)
        exp, law2, rc = oracle.sampler_iteration(star, star.y, T, init_logL, before, law, i=it, z=z, u_mh=u, learn=learn, do_swap=it != 0, ind_A=ia,
                                                 u_swap=us, c0=c0)
        assert rc == 0
        s.run(1)
        aft = s.state()
        assert np.allclose(aft["vars"], exp["vars"], rtol=1e-10, atol=1e-12), np.max(np.abs(aft["vars"] - exp["vars"]))
        assert np.allclose(aft["logL"], exp["logL"], rtol=1e-10) and np.allclose(aft["logPost"], exp["logPost"], rtol=1e-10)
        assert np.allclose(aft["Pmove"], exp["Pmove"], rtol=1e-4, atol=1e-300)
        if learn:
            mu, cov, sig = s.proposal_law()
            assert np.allclose(mu, law2[0], rtol=1e-11, atol=1e-13) and np.allclose(sig, law2[2], rtol=1e-9)
            assert np.allclose(cov, law2[1], rtol=1e-9, atol=1e-9 * np.abs(law2[1]).max())
        return exp

    moved = 0
    for _ in range(2):
        one(False)
    s.run(1, record=False)                       # iteration 2
    for _ in range(3):                           # iterations 3, 4, 5 adapt: the second and third propose with a scratch-built factor
        moved += int(one(True)["moved"].sum())
    s.run(24, record=False)
    assert s.state()["iteration"] == 30
    for _ in range(2):                           # settled: the fused step (device engine) with the adapted law
        moved += int(one(False)["moved"].sum())
    assert moved >= 1
    if engine == "device":
        info = s.info()
        assert info["adapt_in_lds"] == 0 and info["iter_lockstep"] >= 27
    s.close(); c.close()


def test_automatic_fallback_from_the_fused_step_equals_forced_lockstep(pkg, star_long, star165):
    """Nparams + 2 Nvars = 1017 > 971: the candidate roles of the fused step do not fit the tile workgroup's LDS and the engine runs every
    iteration on the lockstep kernels by itself (TAMCMC_OPT_STEP_SCHEME left at 0).  The chains must be those of the forced lockstep
    scheme bit for bit -- and, on a star inside the limit, the automatic scheme must really be the fused one."""
    def chains(star, scheme, n=40):
        c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
        c.set_spectrum(star.x, star.y)
        if scheme is not None:
            c.set_option(pkg.OPT_STEP_SCHEME, scheme)
        s = pkg.Sampler(c, star, engine="device", nchains=8, lambda_temp=1.2, seed=3, Nt_learn=(5, 15), periods_learn=(1,), dN_mixing=1, c0=2.0)
        smp, stt = s.run(n, stats=True)
        smp2, stt2 = s.run(12, stats=True)           # a second call continues the chains
        info = s.info()
        s.close(); c.close()
        return np.concatenate([smp, smp2]), np.concatenate([stt, stt2]), info

    a, sa, ia = chains(star_long, None)
    b, sb, ib = chains(star_long, 1)
    assert ia["fused_available"] == 0 and ia["iter_fused"] == 0 and ia["iter_lockstep"] == 52, ia
    assert ib["iter_fused"] == 0 and ib["iter_lockstep"] == 52
    assert np.array_equal(a, b) and np.array_equal(sa, sb)
    assert all((a[1:, m] != a[:-1, m]).any() for m in range(8))
    # inside the limit the automatic scheme fuses the quiet stretches, and forcing the lockstep kernels gives the same chains
    a, sa, ia = chains(star165, None)
    b, sb, ib = chains(star165, 1)
    assert ia["fused_available"] == 1 and ia["iter_fused"] >= 30 and ib["iter_fused"] == 0, (ia, ib)
    assert np.array_equal(a, b) and np.array_equal(sa, sb)


def test_langevin_engine_beyond_the_lds_limit_follows_the_host_engine(pkg, star165):
    """use_drift = 1 with 153 free variables: k_mala_test keeps its adaptation workspace in device memory (dev_sampler.hip run_mala).  Same
    Philox streams and algorithm as the host engine: the chains coincide to the finite-difference noise through the adaptation window."""
    star = star165
    c = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    c.set_spectrum(star.x, star.y)
    kw = dict(use_drift=1, nchains=3, lambda_temp=1.4, seed=21, Nt_learn=(5, 40), periods_learn=(1,), c0=2.0, dN_mixing=1)
    h = pkg.Sampler(c, star, engine="host", **kw)
    d = pkg.Sampler(c, star, engine="device", **kw)
    n = 50
    sh, th = h.run(n, stats=True)
    sd, td = d.run(n, stats=True)
    assert d.info()["adapt_in_lds"] == 0
    dev = np.max(np.abs(sh - sd) / (np.abs(sh) + 1e-3), axis=(1, 2))
    same = dev < 1e-4
    first_div = n if same.all() else int(np.argmin(same))
    assert first_div >= 25, f"engines diverge at iteration {first_div}: {dev[max(first_div - 3, 0):first_div + 2]}"
    assert np.allclose(th[:first_div], td[:first_div], rtol=1e-5, atol=1e-3)
    assert (sd[1:, 0] != sd[:-1, 0]).any()
    h.close(); d.close(); c.close()
