"""Input front end (include/tamcmc_io.h): `.data` reader and the local-fit `.model` loader, checked on
the reference's own sample input test/inputs/TF_3443483_local-v3.model (committed unchanged under tests/golden/ as a data
fixture) against values derived by hand from that file with the rules of io_local.cpp / io_models.cpp.
The reference cannot run here and holds no dump of the resulting Input_Data: parity unpinned beyond these rules."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL = os.path.join(GOLD, "TF_3443483_local-v3.model")
DATA = os.path.join(GOLD, "TF_3443483_local-v3_slice1.data")


@pytest.fixture(scope="module")
def inputs(pkg):
    from tamcmc_c_amd import inputs as m
    return m


def _harvey(nu):
    return 664.13440 / (1 + (1e-3 * 32.186165 * nu) ** 4.0) + 355.78922 / (1 + (1e-3 * 13.739317 * nu) ** 2.5) + 5.1845856


def test_read_data_and_range_cut(inputs):
    tab = inputs.read_data(DATA)
    assert tab.shape == (973, 2)
    raw = np.array([ln.split() for ln in open(DATA) if ln.strip() and ln.lstrip()[0] not in "#!*"], dtype=np.float64)
    assert np.array_equal(tab, raw)
    a, b = inputs.select_range(tab, 95.0, 96.0)
    assert tab[a, 0] >= 95.0 > tab[a - 1, 0] and tab[b - 1, 0] < 96.0 <= tab[b, 0]
    assert inputs.select_range(tab, 0.0, 1e9) == (0, 973)
    with pytest.raises(Exception):
        inputs.select_range(tab, 1e9, 2e9)       # the reference exits (config.cpp:320-324)
    with pytest.raises(Exception):
        inputs.read_data(os.path.join(GOLD, "no_such_file.data"))


def test_local_model_slice0(inputs, pkg):
    resol = 0.00812042
    inp = inputs.LocalInputs(MODEL, 0, resol)
    assert inp.model_name == "model_MS_local_basic" and inp.model_id == pkg.MODEL_MS_LOCAL_BASIC and inp.prior_class == 3
    assert inp.freq_range == (94.30, 102.20) and inp.dnu == 10.6795 and inp.c_l == 2.59256
    assert list(inp.plength) == [2, 0, 1, 0, 1, 0, 6, 2, 1, 1, 2]          # io_local.cpp:1067-1081
    assert inp.names == ["Height_l", "Height_l", "Frequency_l", "Frequency_l", "Empty", "Asphericity_eta0", "Splitting_a3",
                         "sqrt(splitting_a1).cosi", "sqrt(splitting_a1).sini", "Lorentzian_asymetry", "Width_l", "Width_l",
                         "White_Noise_N0", "Empty", "Truncation parameter", "Switch for fit of Amplitudes or Heights"]
    # rho = (Dnu/135.1)^2 rho_sun; eta0 = 3/(4 pi rho G)  (io_local.cpp:337-343, :894)
    rho_sun = 1.98855e30 * 1e3 / (4 * np.pi * (6.96342e5 * 1e5) ** 3 / 3)
    eta0 = 3.0 / (4.0 * np.pi * (10.6795 / 135.1) ** 2 * rho_sun * 6.667e-8)
    c45 = np.sqrt(0.4) * np.cos(np.radians(45.0))
    n0 = 0.5 * (_harvey(94.30) + _harvey(102.20))                            # set_noise_params_local, io_local.cpp:1221-1226
    want = [1199.06221, 635.50294, 99.10560, 97.77160, 0.0, eta0, 0.0, c45, c45, 0.0, 0.30864, 0.30646, n0, 0.0, 10000.0, 0.0]
    assert np.allclose(inp.params, want, rtol=1e-13, atol=0)
    assert list(inp.relax) == [1, 1, 1, 1, 0, 0, 0, 1, 1, 0, 1, 1, 1, 0, 0, 0]
    assert inp.prior_names == ["Jeffreys", "Jeffreys", "GUG", "GUG", "Fix", "Fix", "Fix", "Uniform", "Uniform", "Fix", "Jeffreys",
                               "Jeffreys", "Uniform", "Fix", "Fix", "Fix"]
    assert list(inp.priors_switch) == [4, 4, 7, 7, 0, 0, 0, 1, 1, 0, 4, 4, 1, 0, 0, 0]   # primepriors_ctrl.list
    P = inp.priors
    assert np.array_equal(P[:, 0], [1.0, 10000.0, -9999.0, -9999.0])       # "Height Jeffreys 1.0 1 10000": values from the 2nd
    d0 = 0.01 * abs(99.64391 - 98.76572)
    assert np.allclose(P[:, 2], [98.76572, 99.64391, d0, d0], rtol=1e-14)   # GUG from the eigen table window
    d2 = 0.01 * abs(98.67162 - 97.43274)
    assert np.allclose(P[:, 3], [97.43274, 98.67162, d2, d2], rtol=1e-14)
    assert np.allclose(P[:, 7], [0.0, np.sqrt(1.5), -9999.0, -9999.0]) and np.array_equal(P[:, 7], P[:, 8])
    assert np.allclose(P[:, 10], [resol, 10.6795 / 3.0, -9999.0, -9999.0])  # Width Fix_Auto
    assert np.allclose(P[:, 12, ][:2], [0.5 * min(_harvey(94.30), _harvey(102.20)), 1.5 * max(_harvey(94.30), _harvey(102.20))])
    assert np.all(P[:, [4, 5, 6, 9, 13, 14, 15]] == -9999.0)                # fixed parameters carry no prior values
    assert np.allclose(inp.extra_priors[:4], [0, 0, 0.2, 0])
    # the assembled vector has a finite prior under the local prior class
    from tamcmc_c_amd import sampler
    star, _ = inputs.load_local_star(MODEL, DATA, 0)
    lp, st = sampler.log_prior(star)
    assert st == 0 and np.isfinite(lp)


def test_local_model_other_slices(inputs):
    s1 = inputs.LocalInputs(MODEL, 1, 0.008)       # 106.65 - 112.74: l=0 109.318, l=2 107.867, l=3 111.445
    assert list(s1.plength) == [3, 0, 1, 0, 1, 1, 6, 3, 1, 1, 2]
    assert np.allclose(s1.params[:3], [1563.15512, 828.47217, 95.92497]) and np.allclose(s1.params[3:6], [109.318, 107.867, 111.445])
    s7 = inputs.LocalInputs(MODEL, 7, 0.008)       # 170.29 - 175.19: l=0 173.685, l=2 172.639
    assert list(s7.plength[:6]) == [2, 0, 1, 0, 1, 0]
    with pytest.raises(Exception):
        inputs.LocalInputs(MODEL, 8, 0.008)        # no ninth '*' range
    with pytest.raises(Exception):
        inputs.LocalInputs(os.path.join(GOLD, "missing.model"), 0, 0.008)


def test_load_local_star_cuts_the_data(inputs):
    star, inp = inputs.load_local_star(MODEL, DATA, 0)
    assert star.x.size == star.y.size == 973                   # the committed slice lies inside [94.30, 102.20)
    assert star.x[0] >= 94.30 and star.x[-1] < 102.20
    assert star.nvars == 9 and star.prior_class == 3


SUN = os.path.join(GOLD, "Sun_19992002_incfix_fast_Priorevalrange.model")   # reference: test/inputs/Sun/fast/ (unchanged copy)


def test_global_model_aj(inputs, pkg):
    """model_MS_Global_aj_HarveyLike from the reference's Sun sample (io_ms_global.cpp rules, hand-derived expectations)."""
    resol = 0.0105
    inp = inputs.GlobalInputs(SUN, resol)
    assert inp.model_name == "model_MS_Global_aj_HarveyLike" and inp.model_id == pkg.MODEL_MS_GLOBAL_AJ and inp.prior_class == 2
    assert inp.freq_range == (2330.0, 3660.0) and inp.dnu == 135.00258
    assert list(inp.plength) == [10, 3, 10, 10, 10, 10, 14, 10, 10, 1, 2] and inp.params.size == 90   # io_ms_global.cpp:1313-1324
    o = np.cumsum([0] + list(inp.plength))
    # heights / widths: the l=0 rows of the eigen table; "Height Jeffreys 1 1000" -> values from the FIRST number (:880-886)
    assert np.allclose(inp.params[:3], [0.35844, 0.64263, 1.00745]) and np.array_equal(inp.priors[:, 0], [1.0, 1000.0, -9999.0, -9999.0])
    assert inp.names[o[1]:o[2]] == ["Visibility_l1", "Visibility_l2", "Visibility_l3"]
    assert np.allclose(inp.params[o[1]:o[2]], [1.5, 0.53, 0.08]) and np.allclose(inp.priors[:2, o[1] + 1], [0.53, 0.03])
    f = inp.params[o[2]:o[6]]
    assert np.isclose(f[0], 2362.80591) and np.isclose(f[10], 2425.59204) and np.isclose(f[39], 3625.77271)
    k = o[2] + 10                                                          # first l=1 frequency: GUG from its window, sigma = Dnu/100
    assert inp.prior_names[k] == "GUG" and np.allclose(inp.priors[:, k], [2424.64893, 2426.53516, 1.3500258, 1.3500258])
    s = o[6]
    assert inp.names[s:s + 14] == ["a1_0", "a1_1", "a2_0", "a2_1", "a3_0", "a3_1", "a4_0", "a4_1", "a5_0", "a5_1", "a6_0", "a6_1",
                                    "eta0_switch", "Lorentzian_asymetry"]
    assert list(inp.relax[s:s + 14]) == [1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1]
    assert np.allclose(inp.priors[:2, s], [0.1, 0.6]) and inp.params[s] == 0.4 and inp.params[s + 2] == -0.02
    assert inp.prior_names[s + 13] == "Jeffreys_abs" and np.allclose(inp.priors[:2, s + 13], [5.0, 200.0])
    wd = o[7]
    assert inp.names[wd] == "Width_l" and np.allclose(inp.priors[:2, wd], [resol, 135.00258 / 3.0]) and inp.params[wd] == 0.75319
    nz = o[8]                                                              # set_noise_params, io_ms_global.cpp:1447-1536
    assert np.allclose(inp.params[nz:nz + 10], [0, 0, 1, 1.2709550, 49.575897, 2.0, 2.6596093, 1.5192964, 2.0, 0.0051731238])
    assert list(inp.relax[nz:nz + 10]) == [0, 0, 0, 0, 0, 0, 1, 1, 1, 1]
    assert inp.prior_names[nz + 6:nz + 10] == ["Gaussian"] * 4
    assert np.allclose(inp.priors[:2, nz + 6], [2.6596093, 2.6596093 * 0.05])          # 3/2 (err- + err+) = 2.4 % -> 5 % floor
    assert np.allclose(inp.priors[:2, nz + 7], [1.5192964, (0.0074767880 + 0.0075137648) * 1.5])
    assert np.allclose(inp.priors[:2, nz + 8], [2.0, 0.2])                              # p without errors -> 10 %
    assert np.allclose(inp.priors[:2, nz + 9], [0.0051731238, 0.00051731238])           # Gaussian N0: 10 %
    assert inp.names[o[9]] == "Inclination" and inp.params[o[9]] == 88.18647 and inp.relax[o[9]] == 0
    assert inp.params[o[10]] == 50.0 and inp.params[o[10] + 1] == 0.0                   # trunc_c, do_amp
    assert np.allclose(inp.extra_priors, [1, 2, 1e6, 0.5, 0.2, 0.15, 0.05, 0.05, 0, 9])
    from tamcmc_c_amd import sampler
    star = inputs.star_from_inputs(inp, np.arange(2330.0, 3660.0, resol))
    lp, st = sampler.log_prior(star)
    assert st == 0 and np.isfinite(lp) and star.nvars == int(inp.relax.sum()) == 71
    with pytest.raises(Exception):
        inputs.GlobalInputs(MODEL, resol)           # eight '*' ranges: not a global-fit file (io_ms_global.cpp:93-104)
    with pytest.raises(Exception):
        inputs.LocalInputs(SUN, 0, resol)           # the loaders refuse each other's model_fullname


def test_cfg_dialect_and_sampler_settings(inputs):
    """`.cfg` reader (Config::format_line / read_cfg_file, config.cpp:1062-1112, :1223-1732) on a fixture written for this test."""
    c = inputs.Cfg(os.path.join(GOLD, "sampler_test.cfg"))
    assert c.string("MALA", "proposal_type") == "Random"
    assert c.numbers("MALA", "lambda_temp")[0] == 3.5                  # "3.50 #1.70; ..." : the value is the leading number
    assert list(c.numbers("MALA", "Nt_learn")) == [200, 600, 4000] and list(c.numbers("MALA", "periods_learn")) == [1, 5]
    assert c.numbers("Data", "ysig_col")[0] == -1 and c.string("Outputs", "file_format") == "binary"
    with pytest.raises(KeyError):
        c.string("MALA", "ignored")                                        # after /END;
    with pytest.raises(KeyError):
        c.string("MALA", "no_such_key")
    kw, rest = c.sampler_kwargs()
    assert kw == dict(nchains=4, lambda_temp=3.5, use_drift=0, p=1.0, target_acceptance=0.234, c0=5.0, epsilon1=1e-12, epsilon2=1e-12,
                      A1=1e14, delta=0.0, delta_x=1e-10, dN_mixing=1, Nt_learn=(200, 600, 4000), periods_learn=(1, 5))
    assert rest == dict(Nsamples=150, Nbuffer=10000, prior_class=3, likelihood_id=0)   # io_local, chi(2,2p)
    c.close()
    with pytest.raises(Exception):
        inputs.Cfg(os.path.join(GOLD, "missing.cfg"))


def test_initial_errors_from_the_errors_file(inputs):
    """err = A*value + B by parameter name, 1 without a match (Config::read_defautlerrors + MALA::init_proposal)."""
    star, inp = inputs.load_local_star(MODEL, DATA, 0)
    e = inputs.init_errors(os.path.join(GOLD, "errors_test.cfg"), star)
    idx = list(star.index_to_relax)
    names = [star.names[i] for i in idx]
    assert names == ["Height_l", "Height_l", "Frequency_l", "Frequency_l", "sqrt(splitting_a1).cosi", "sqrt(splitting_a1).sini", "Width_l",
                     "Width_l", "White_Noise_N0"]
    v = star.params[idx]
    want = [0.02 * v[0] + 0.01, 0.02 * v[1] + 0.01, 0.07, 0.07, 0.1 * v[4] + 0.05, 1.0, 0.015 * v[6] + 0.005, 0.015 * v[7] + 0.005,
            0.015 * v[8] + 0.0002]                                          # ".sini" has no row in this fixture -> 1
    assert np.allclose(e, want, rtol=1e-15)


def test_parsers_survive_malformed_input_under_sanitizers(tmp_path):
    """The front end's C++ (csrc/host_io.cpp, csrc/host_cfg.cpp) built for the CPU with AddressSanitizer + UBSan, fed every prefix,
    line deletion and junk-line substitution of the fixtures: errors must be codes, never memory errors (the reference exits or
    reads out of bounds on such input)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tamcmc-c_amd", "csrc")
    exe = str(tmp_path / "fuzz")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
           os.path.join(root, "tests", "parser_fuzz_driver.cpp"), os.path.join(src, "host_io.cpp"), os.path.join(src, "host_cfg.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    files = [MODEL, SUN, os.path.join(GOLD, "RGB_10722175.model"), DATA, os.path.join(GOLD, "sampler_test.cfg"), os.path.join(GOLD, "errors_test.cfg")]
    r = subprocess.run([exe, str(tmp_path / "variant.txt")] + files, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "variants" in r.stdout
