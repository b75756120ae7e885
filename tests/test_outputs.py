"""Output formats of the kept surface (row N3): params .bin/.hdr, stat_criteria, bin2txt-style summary statistics. No GPU."""
import numpy as np


def test_params_bin_hdr_roundtrip_and_summary(pkg, synth, tmp_path):
    from tamcmc_c_amd import sampler
    star = synth.make_c2_star(nx=512)
    rng = np.random.default_rng(2)
    n, nc, nv = 257, 3, star.nvars
    smp = rng.standard_normal((n, nc, nv)) * np.arange(1, nv + 1) + 10.0
    stt = rng.standard_normal((n, nc, 3))
    root = str(tmp_path / "star_A_")
    sampler.write_outputs(root, star, smp[:100], stt[:100], nsamples_total=n)
    sampler.write_outputs(root, star, smp[100:], stt[100:], append=True)
    for m in range(nc):
        back = sampler.read_params(root, m)
        assert back.shape == (n, nv) and np.array_equal(back, smp[:, m, :])
        raw = np.fromfile(root + f"params_chain-{m}.bin", dtype="<f8")      # raw little-endian doubles [sample][var]
        assert np.array_equal(raw.reshape(n, nv), smp[:, m, :])
    hdr = open(root + "params.hdr").read()
    assert f"! Nchains= {nc}" in hdr and f"! Nvars= {nv}" in hdr and f"! Ncons= {star.params.size - nv}" in hdr
    assert "! plength= " + " ".join(str(v) for v in star.plength) in hdr
    assert "! variable_names=" in hdr and "Frequency_l" in hdr and "! constant_values=" in hdr
    st = np.fromfile(root + "stat_criteria.bin", dtype="<f8").reshape(n, 3, nc)   # logL[chains], logPrior[chains], logPost[chains]
    assert np.array_equal(st[:, 0, :], stt[:, :, 0]) and np.array_equal(st[:, 2, :], stt[:, :, 2])
    mean, med, sd = sampler.params_summary(smp[:, 0, :])
    assert np.allclose(mean, smp[:, 0, :].mean(0), rtol=1e-14) and np.allclose(med, np.median(smp[:, 0, :], axis=0), rtol=1e-15)
    assert np.allclose(sd, smp[:, 0, :].std(0), rtol=1e-13)        # population standard deviation, like the reference tool
    mean2, med2, _ = sampler.params_summary(smp[:256, 0, :])       # even count: median = mean of the two central values
    assert np.allclose(med2, np.median(smp[:256, 0, :], axis=0), rtol=1e-15)


def test_restore_files_round_trip(pkg, tmp_path):
    """Checkpoint files in the layout of Outputs::write_buffer_restore (outputs.cpp:863-1025), read back as
    Config::read_restore_files does (config.cpp:1734-1990); 17 significant digits -> exact round trip."""
    import ctypes as C
    from tamcmc_c_amd import sampler as S
    L = S._rebind()
    rng = np.random.default_rng(4)
    nc, nv, it = 3, 5, 1234
    vars_, sig, mus = rng.standard_normal((nc, nv)) * 1e3, rng.uniform(0.1, 2, nc), rng.standard_normal((nc, nv))
    cov = rng.standard_normal((nc, nv, nv)) * 1e-3
    names = (C.c_char_p * nv)(*[("v%d" % i).encode() for i in range(nv)])
    root = str(tmp_path / "restore_")
    assert L.tamcmc_outputs_write_restore(root.encode(), nc, nv, it, names, S._p(vars_), S._p(sig), S._p(mus), S._p(cov)) == 0
    txt = open(root + "1.dat").read()
    assert "! Nchains= 3" in txt and "! Nvars= 5" in txt and "! iteration=1234" in txt and "! variable_names=v0   v1" in txt
    assert "! vars= " in txt and "! vars_mean= " in txt and "*2" in open(root + "3.dat").read() and "! sigmas= " in open(root + "2.dat").read()
    a, b, c = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int64)
    assert L.tamcmc_outputs_read_restore(root.encode(), S._p(a, S._ip), S._p(b, S._ip), S._p(c, S._i64p), None, None, None, None) == 0
    assert (a[0], b[0], c[0]) == (nc, nv, it)
    v2, s2, m2, c2 = np.zeros((nc, nv)), np.zeros(nc), np.zeros((nc, nv)), np.zeros((nc, nv, nv))
    assert L.tamcmc_outputs_read_restore(root.encode(), S._p(a, S._ip), S._p(b, S._ip), S._p(c, S._i64p), S._p(v2), S._p(s2), S._p(m2),
                                         S._p(c2)) == 0
    assert np.array_equal(v2, vars_) and np.array_equal(s2, sig) and np.array_equal(m2, mus) and np.array_equal(c2, cov)
    assert L.tamcmc_outputs_read_restore(str(tmp_path / "absent_").encode(), S._p(a, S._ip), S._p(b, S._ip), S._p(c, S._i64p), None, None,
                                         None, None) != 0


def test_evidence_diagnostic_matches_the_oracle_and_known_answers(pkg, oracle, tmp_path):
    """Diagnostics::evidence_calc (diagnostics.cpp:980-1019) on top of quad_interpol (interpol.cpp:46-101): known answers, then the
    host library against the oracle's restatement on random ladders."""
    from tamcmc_c_amd import sampler as S
    # quad_interpol: resampling to the same length returns the nodes; a straight line stays a straight line; the half-sample
    # parabolas reproduce the node values and are linear in the first and last half interval
    a = np.array([3.0, 1.0, 4.0, 1.0, 5.0, 9.0])
    assert np.array_equal(oracle.quad_interpol(a, a.size), a)
    lin = 2.0 + 0.5 * np.arange(7)
    assert np.allclose(oracle.quad_interpol(lin, 61), np.linspace(lin[0], lin[-1], 61), rtol=1e-15)
    fine = oracle.quad_interpol(a, 51)                       # index step 0.1
    assert np.allclose(fine[::10], a, rtol=1e-15) and np.allclose(fine[:6], a[0] + 0.1 * np.arange(6) * (a[1] - a[0]))
    assert np.isclose(fine[15], (0.5 * (a[1] + a[2]) + 0.5 * (a[1] + a[2])) / 2)   # x = 1.5: the parabola's end value, the mid-point mean
    rng = np.random.default_rng(11)
    for nc, n, k in ((4, 300, 7), (10, 57, 1000), (2, 5, 3)):
        T = 1.7 ** np.arange(nc)
        stats = rng.standard_normal((n, nc, 3)) * 5 - 1000.0 / T[None, :, None]
        ev, beta, Lb, bi, Li = S.evidence(T, stats, k, out_file=tmp_path / "ev.txt", first=True)
        ev_o, beta_o, Lb_o, bi_o, Li_o = oracle.evidence(T, stats[:, :, 0], k)
        assert np.array_equal(beta, 1.0 / T) and np.array_equal(beta, beta_o)
        assert np.allclose(Lb, stats[:, :, 0].mean(0), rtol=1e-14) and np.allclose(Lb, Lb_o, rtol=1e-15)
        assert np.allclose(bi, bi_o, rtol=1e-15) and np.allclose(Li, Li_o, rtol=1e-14) and np.isclose(ev, ev_o, rtol=1e-14)
        assert Li.min() >= Lb.min() - 1e-9 and Li.max() <= Lb.max() + 1e-9      # monotone ladder: the resampling does not overshoot
        S.evidence(T, stats, k, out_file=tmp_path / "ev.txt", first=False)
        lines = open(tmp_path / "ev.txt").read().splitlines()
        body = [ln for ln in lines if not ln.startswith(("#", "!"))]
        assert len(body) == 2 and lines[5].startswith("! beta=") and lines[6] == f"! interpolation_factor={k}"
        cols = np.array(body[1].split(), dtype=float)
        assert cols[0] == n and np.allclose(cols[1:1 + nc], Lb, rtol=1e-9) and np.isclose(cols[-1], ev, rtol=1e-9)
    with np.testing.assert_raises(Exception):
        S.evidence(np.array([1.0]), np.zeros((3, 1, 3)), 1)        # one resampled point: the reference divides by zero
