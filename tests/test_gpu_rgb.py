"""Red-giant model (id 25, BASELINE config C5 family) on the device: the ARMM mixed-mode solver and the zeta function run as HIP
kernels (csrc/rgb_prestep.hip), the variable-length table goes through the same k_loglike as the main-sequence models.
Checked against the oracle's restatement (oracle/armm_oracle.c, pinned on the reference's own solver outputs: tests/test_armm_scanner_fixtures.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bias_type,model_type,cte", [(0, 0, False), (1, 0, False), (2, 1, False), (0, 1, False), (1, 0, True), (0, 1, True)])
def test_rgb_model_rows_and_logl_match_the_oracle(pkg, oracle, synth, bias_type, model_type, cte):
    rng = np.random.default_rng(5)
    params, pl = synth.make_params_rgb_model(rng, bias_type=bias_type, model_type=model_type, cte_width=cte)
    model_id = synth.MODEL_RGB_CTE_V4 if cte else synth.MODEL_RGB_V4     # 27: constant-width variant (models.cpp:4334)
    step = 0.05
    x = 110.0 + step * np.arange(3400)
    st, m0 = oracle.call_model(model_id, params, pl, x)
    assert st == 0
    y = m0 * np.random.default_rng(2).exponential(1.0, m0.size)
    B = 5
    P = np.tile(params, (B, 1))
    o = np.cumsum([0] + list(pl))
    P[1:, :pl[0]] *= 1 + 0.05 * rng.standard_normal((B - 1, pl[0]))                 # heights
    P[1:, o[3] + 1] *= 1 + 0.002 * rng.standard_normal(B - 1)                        # period spacing: moves every mixed mode
    P[1:, o[3] + 3] *= 1 + 0.05 * rng.standard_normal(B - 1)                         # coupling
    T = 1.4 ** np.arange(B)
    if cte:
        P[1:, o[7]] *= 1 + 0.1 * rng.standard_normal(B - 1)                          # the one width
    ref, m_o, st_o = oracle.loglike_batch(model_id, P, pl, x, y, 1.0, T, want_model=True)
    assert (st_o == 0).all()
    for prec, tol_m, tol_l in ((pkg.PRECISION_STRICT, 1e-10, 1e-12), (pkg.PRECISION_FAST, 1e-10, 1e-12)):
        ctx = pkg.HipContext(0, precision=prec)
        ctx.set_spectrum(x, y)
        got, m_d, st_d = ctx.loglike_params_batch(model_id, P, pl, T, want_model=True)
        assert (st_d == 0).all()
        # same mixed modes (a missing or extra mode would change the row by O(1) around it).  Stated red-giant tolerance
        # (include/tamcmc_hip.h): rows ||dM||_2 / ||M||_2 <= 1e-10, logL 1e-12 relative.  Measured on the MI355X (tools/rgb_parity_probe.py,
        # round 3): mixed-mode frequencies within 3e-13 muHz of the oracle's (1.5e-11 at the 2e5-bin C5 size), zeta within 4e-13, rows
        # 2e-13 (1e-11 at C5), logL 3e-15 -- the device's double tan / atan against the long double of solver_mm.cpp:179-186
        rel = np.linalg.norm(m_d - m_o, axis=1) / np.linalg.norm(m_o, axis=1)
        assert rel.max() < tol_m, rel
        assert np.allclose(got, ref, rtol=tol_l, atol=0), np.abs(got / ref - 1).max()
        ctx.close()


def test_rgb_bad_vectors_are_reported_not_guessed(pkg, oracle, synth):
    rng = np.random.default_rng(6)
    params, pl = synth.make_params_rgb_model(rng)
    x = 110.0 + 0.05 * np.arange(3400)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(x, np.ones_like(x))
    P = np.tile(params, (3, 1))
    o = np.cumsum([0] + list(pl))
    P[1, o[3] + 1] = 1e9                       # period spacing so large that no g mode falls in the range: "impossible star"
    P[2, o[2]:o[3]] = P[2, o[2]:o[3]][::-1]    # radial modes in decreasing order: negative large separation
    got, _, st = ctx.loglike_params_batch(pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4, P, pl, None)
    assert st[0] == 0 and np.isfinite(got[0])
    assert st[2] != 0 and np.isnan(got[2])
    # no g mode in range: the reference's solver returns an empty set and the model carries on without l=1 modes
    ref, _, st_o = oracle.loglike_batch(synth.MODEL_RGB_V4, P[1:2], pl, x, np.ones_like(x), 1.0, None)
    assert st[1] == 0 and st_o[0] == 0 and np.isclose(got[1], ref[0], rtol=1e-10)
    ctx.close()


@pytest.mark.parametrize("cte", [False, True])
def test_rgb_star_samples_on_the_host_engine(pkg, oracle, synth, cte):
    """Red-giant star end to end: priors (io_asymptotic) on the host, ONE batched device call per iteration whose tables come from
    the device pre-step, adaptive MH + parallel tempering (host-driven engine)."""
    star = synth.make_c5_star(nx=6000, nmax=6, nferr=4, cte_width=cte)
    st, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    assert st == 0
    star.set_spectrum_from_model(m0, 4)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    s = pkg.Sampler(ctx, star, nchains=4, lambda_temp=1.6, seed=3, engine="host", Nt_learn=(20, 200), periods_learn=(1,))
    st0 = s.state()
    T = 1.6 ** np.arange(4)
    ref, _, so = oracle.loglike_batch(star.model_id, np.tile(star.params, (4, 1)), star.plength, star.x, star.y, 1.0, T)
    assert (so == 0).all() and np.allclose(st0["logL"], ref, rtol=1e-11) and np.isfinite(st0["logPrior"]).all()
    smp, stat = s.run(150, stats=True)
    assert np.isfinite(stat).all() and s.state()["iteration"] == 150
    assert (smp[:, 0] != smp[0, 0]).any() and s.state()["swap_attempts"] == 149
    assert stat[-50:, 0, 2].mean() > st0["logPost"][0] - 30.0
    s.close()
    ctx.close()


@pytest.mark.parametrize("cte,learn", [(False, None), (True, None), (False, (15, 70))])
def test_rgb_device_engine_follows_the_host_engine(pkg, oracle, synth, cte, learn):
    """Red-giant star on the device-resident engine: proposal and log-prior in k_iterate, then the pre-step (scalar unpack, mixed-mode
    solver, zeta, rows) on the proposals where they lie in device memory, likelihood, MH test / swap / adaptation in the next k_iterate:
    no host round trip per iteration.  Same random streams as the host-driven engine (which evaluates prior and scalar unpack in long
    double like the reference, the device in double): the trajectories agree to rounding until a knife-edge decision."""
    star = synth.make_c5_star(nx=6000, nmax=6, nferr=4, cte_width=cte)
    st, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    assert st == 0
    star.set_spectrum_from_model(m0, 4)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    kw = dict(nchains=5, lambda_temp=1.5, seed=21, Nt_learn=learn or (10**9, 10**9 + 1), periods_learn=(1,), dN_mixing=1)
    h = pkg.Sampler(ctx, star, engine="host", **kw)
    d = pkg.Sampler(ctx, star, engine="device", **kw)
    n = 100
    sh, th = h.run(n, stats=True)
    sd, td = d.run(60, stats=True)
    sd2, td2 = d.run(n - 60, stats=True)                     # a second call continues the chain
    sd, td = np.concatenate([sd, sd2]), np.concatenate([td, td2])
    same = np.all(np.isclose(sh, sd, rtol=1e-8, atol=1e-11), axis=(1, 2))
    first_div = n if same.all() else int(np.argmin(same))
    assert first_div >= 40, f"engines diverge at iteration {first_div}"
    assert np.allclose(th[:first_div], td[:first_div], rtol=1e-7, atol=1e-6)
    a, b = h.state(), d.state()
    assert a["iteration"] == b["iteration"] == n and a["swap_attempts"] == b["swap_attempts"] == n - 1
    assert np.isfinite(td).all() and (sd[:, 0] != sd[0, 0]).any()
    # what the engine holds for each chain is the likelihood of the position it holds (oracle, red-giant tolerance)
    T = 1.5 ** np.arange(5)
    held = np.tile(star.params, (5, 1))
    held[:, star.index_to_relax] = b["vars"]
    ref, _, so = oracle.loglike_batch(star.model_id, held, star.plength, star.x, star.y, 1.0, T)
    assert (so == 0).all() and np.allclose(b["logL"], ref, rtol=1e-10), np.abs(b["logL"] / ref - 1).max()
    h.close(); d.close(); ctx.close()


def test_rgb_device_engine_rejects_vectors_the_prestep_refuses(pkg, oracle, synth):
    """A chain started where proposals often leave the prior or break the l=0 ladder: such proposals are rejected (never accepted with
    a table that was not built), the chain stays finite."""
    from tamcmc_c_amd.sampler import default_errors
    star = synth.make_c5_star(nx=6000, nmax=6, nferr=4)
    st, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    star.set_spectrum_from_model(m0, 4)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    d = pkg.Sampler(ctx, star, nchains=3, lambda_temp=2.0, seed=2, engine="device", Nt_learn=(10**9, 10**9 + 1), periods_learn=(1,),
                    init_errors=400.0 * default_errors(star))
    smp, stat = d.run(80, stats=True)
    assert np.isfinite(stat).all() and np.isfinite(smp).all()
    moved = np.any(smp[1:] != smp[:-1], axis=2)
    assert moved.mean() < 0.5                                 # most of these wild proposals are refused
    stt = d.state()
    held = np.tile(star.params, (3, 1))
    held[:, star.index_to_relax] = stt["vars"]
    ref, _, so = oracle.loglike_batch(star.model_id, held, star.plength, star.x, star.y, 1.0, 2.0 ** np.arange(3))
    assert (so == 0).all() and np.allclose(stt["logL"], ref, rtol=1e-10), np.abs(stt["logL"] / ref - 1).max()
    d.close(); ctx.close()


def test_structured_scan_finds_the_cells_of_the_dense_walk(pkg, synth):
    """The mixed-mode solver's first pass looks for grid cells where p(nu) - g(nu) changes sign.  The reference walks every grid
    point (solver_mm.cpp:330-376); the device scan uses the function's structure instead (poles of tan in closed form, bisection on
    the grid index between them: csrc/rgb_prestep.hip).  Both must deliver the same cells, hence bit-identical model rows -- over
    period spacings, couplings, both solver drivers and all bias types, including a coarse and a fine frequency grid."""
    rng = np.random.default_rng(17)
    cases = []
    for k in range(10):
        cases.append(dict(DPl=float(rng.uniform(60, 320)), q=float(rng.uniform(0.05, 0.6)), dnu=float(rng.uniform(8, 22)),
                          model_type=k % 2, bias_type=k % 3, alpha_g=float(rng.uniform(0, 0.5)), step=(0.05 if k % 4 else 0.011)))
    for c in cases:
        params, pl = synth.make_params_rgb_model(np.random.default_rng(3), dnu=c["dnu"], DPl=c["DPl"], q=c["q"], alpha_g=c["alpha_g"],
                                                 model_type=c["model_type"], bias_type=c["bias_type"])
        o = np.cumsum([0] + list(pl))
        fl0 = params[o[2]:o[3]]
        x = fl0.min() - 1.2 * c["dnu"] + c["step"] * np.arange(int((fl0.max() - fl0.min() + 2.4 * c["dnu"]) / c["step"]))
        P = np.tile(params, (4, 1))
        P[1:, o[3] + 1] *= 1 + 0.01 * rng.standard_normal(3)
        P[1:, o[3]] += 0.05 * rng.standard_normal(3)
        ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
        ctx.set_spectrum(x, np.ones_like(x))
        out = {}
        for dense in ("1", "0"):
            ctx.set_option(pkg.OPT_ARMM_DENSE_SCAN, int(dense))
            logL, rows, st = ctx.loglike_params_batch(pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4, P, pl, None, want_model=True)
            assert (st == 0).all(), (c, st)
            out[dense] = (logL, rows)
        assert np.array_equal(out["0"][1], out["1"][1]), c
        assert np.array_equal(out["0"][0], out["1"][0]), c
        ctx.close()


def test_real_red_giant_from_the_reference_files(pkg, oracle):
    """KIC 10722175 from the reference's own example files (test/inputs/RGB/v1.86.0, copied as data under tests/golden/): `.model`
    through the red-giant dialect of the input front end, `.data` cut to its range, then the device path against the oracle at the
    file's starting point and a short tempered run of the host-driven engine on the real spectrum."""
    import os
    from tamcmc_c_amd import inputs
    g = os.path.join(os.path.dirname(__file__), "golden")
    star, inp = inputs.load_asymptotic_star(os.path.join(g, "RGB_10722175.model"), os.path.join(g, "RGB_10722175.data"))
    assert inp.model_id == pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4 and star.x.size == 6099
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    T = 1.5 ** np.arange(4)
    P = np.tile(star.params, (4, 1))
    o = np.cumsum([0] + list(star.plength))
    P[1:, o[3] + 1] += [0.05, -0.08, 0.11]           # period spacing
    P[1:, o[3] + 24:o[4]] = 0.01 * np.random.default_rng(1).standard_normal((3, 16))   # bias values at the spline nodes
    ref, m_o, st_o = oracle.loglike_batch(star.model_id, P, star.plength, star.x, star.y, 1.0, T, want_model=True)
    got, m_d, st_d = ctx.loglike_params_batch(star.model_id, P, star.plength, T, want_model=True)
    assert (st_o == 0).all() and (st_d == 0).all()
    rel = np.linalg.norm(m_d - m_o, axis=1) / np.linalg.norm(m_o, axis=1)
    assert rel.max() < 1e-10, rel
    assert np.allclose(got, ref, rtol=1e-11, atol=0), np.abs(got / ref - 1).max()
    s = pkg.Sampler(ctx, star, nchains=4, lambda_temp=1.5, seed=11, engine="host", Nt_learn=(20, 150), periods_learn=(1,))
    st0 = s.state()
    assert np.allclose(st0["logL"], ref[0] * T[0] / T, rtol=1e-11) and np.isfinite(st0["logPrior"]).all()
    smp, stat = s.run(160, stats=True)
    assert np.isfinite(stat).all() and (smp[:, 0] != smp[0, 0]).any() and s.state()["swap_attempts"] == 159
    assert stat[-40:, 0, 2].mean() > st0["logPost"][0] - 40.0
    s.close()
    ctx.close()


@pytest.mark.parametrize("case", [
    dict(nmax=3, dnu=12.0, DPl=90.0, q=0.2, step=0.05, B=1),                    # the smallest star the model accepts, one vector
    dict(nmax=6, dnu=18.0, DPl=310.0, q=0.9, step=0.05, B=33, alpha_g=0.9),     # sparse g modes, strong coupling, odd batch
    dict(nmax=5, dnu=9.0, DPl=70.0, q=0.05, step=0.2, B=3),                     # coarse grid: few points per mixed-mode spacing
    dict(nmax=8, dnu=7.0, DPl=75.0, q=0.15, step=0.02, B=2, model_type=1, bias_type=2, nferr=9),   # dense spectrum, Hermite bias, from-l0 driver
    dict(nmax=6, dnu=15.0, DPl=85.0, q=0.3, step=0.05, B=4, cte=True, model_type=1),
])
def test_rgb_awkward_stars_match_the_oracle(pkg, oracle, synth, case):
    """Red-giant stars away from the bench configuration: the device pre-step (pole-structured scan, lane-parallel refinement,
    sort/unique, zeta normalisation, rows) must deliver the oracle's mixed modes in every regime."""
    rng = np.random.default_rng(23)
    cte = case.get("cte", False)
    params, pl = synth.make_params_rgb_model(rng, nmax=case["nmax"], dnu=case["dnu"], DPl=case["DPl"], q=case["q"], alpha_g=case.get("alpha_g", 0.0),
                                             model_type=case.get("model_type", 0), bias_type=case.get("bias_type", 1), nferr=case.get("nferr", 4),
                                             cte_width=cte)
    model_id = synth.MODEL_RGB_CTE_V4 if cte else synth.MODEL_RGB_V4
    o = np.cumsum([0] + list(pl))
    fl0 = params[o[2]:o[3]]
    lo = fl0.min() - 1.3 * case["dnu"]
    x = lo + case["step"] * np.arange(int((fl0.max() - fl0.min() + 2.6 * case["dnu"]) / case["step"]))
    B = case["B"]
    P = np.tile(params, (B, 1))
    if B > 1:
        P[1:, o[3] + 1] *= 1 + 0.004 * rng.standard_normal(B - 1)
        P[1:, o[3] + 3] *= 1 + 0.05 * rng.standard_normal(B - 1)
        P[1:, :pl[0]] *= 1 + 0.05 * rng.standard_normal((B - 1, pl[0]))
    st, m0 = oracle.call_model(model_id, params, pl, x)
    assert st == 0
    y = m0 * np.random.default_rng(4).exponential(1.0, m0.size)
    T = 1.1 ** np.arange(B)
    ref, m_o, st_o = oracle.loglike_batch(model_id, P, pl, x, y, 1.0, T, want_model=True)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(x, y)
    got, m_d, st_d = ctx.loglike_params_batch(model_id, P, pl, T, want_model=True)
    assert np.array_equal(st_d == 0, st_o == 0), (st_d, st_o)
    ok = st_o == 0
    assert ok.any()
    rel = np.linalg.norm(m_d[ok] - m_o[ok], axis=1) / np.linalg.norm(m_o[ok], axis=1)
    assert rel.max() < 1e-10, rel
    assert np.allclose(got[ok], ref[ok], rtol=1e-12, atol=0), np.abs(got[ok] / ref[ok] - 1).max()
    ctx.close()


def test_rgb_full_size_c5_matches_the_oracle(pkg, oracle, synth):
    """BASELINE configs[4] size: 2e5 bins, ~150 mixed modes per vector (the oracle needs ~2 s per evaluation on the host)."""
    star = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
    st, m0 = oracle.call_model(star.model_id, star.params, star.plength, star.x)
    assert st == 0
    y = star.set_spectrum_from_model(m0, 7)
    rng = np.random.default_rng(9)
    P = np.tile(star.params, (2, 1))
    o = np.cumsum([0] + list(star.plength))
    P[1, o[3] + 1] *= 1.0007          # period spacing: every mixed mode moves
    P[1, :star.plength[0]] *= 1 + 0.03 * rng.standard_normal(star.plength[0])
    T = np.array([1.0, 1.15])
    ref, _, st_o = oracle.loglike_batch(star.model_id, P, star.plength, star.x, y, 1.0, T)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, y)
    got, _, st_d = ctx.loglike_params_batch(star.model_id, P, star.plength, T)
    assert (st_o == 0).all() and (st_d == 0).all()
    assert np.allclose(got, ref, rtol=1e-12, atol=0), np.abs(got / ref - 1).max()
    ctx.close()
