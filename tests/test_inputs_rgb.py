"""Input front end, red-giant dialect (row N2 of SURVEY 8f): build_init_asymptotic (tamcmc/sources/io_asymptotic.cpp:32-955) restated in
csrc/host_io.cpp::build_asymptotic.  Fixtures: the reference's own example star test/inputs/RGB/v1.86.0/10722175{,_nobias}.model +
.data (KIC 10722175), copied as data under tests/golden/.  No expected numbers ship with the reference for this path (parity unpinned
by it): the loader is checked against the rules read from its source, the resulting vector against the oracle's model and prior."""
import os
import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
MODEL, NOBIAS, DATA = (os.path.join(G, n) for n in ("RGB_10722175.model", "RGB_10722175_nobias.model", "RGB_10722175.data"))


def test_asymptotic_model_file_of_the_reference(pkg, oracle):
    from tamcmc_c_amd import inputs, sampler
    star, inp = inputs.load_asymptotic_star(MODEL, DATA)
    assert inp.model_id == 25 and inp.prior_class == 4 and inp.model_name == "model_RGB_asympt_aj_AppWidth_HarveyLike_v4"
    # 5 radial orders; the ONE l=1 line of the file is a placeholder whose slot holds 8 global parameters + 2 x 16 spline nodes
    assert list(inp.plength) == [5, 3, 5, 8 + 2 * 16, 4, 2, 10, 6, 10, 1, 6]
    assert star.x[0] >= 80.0 and star.x[-1] <= 128.0 and star.x.size == 6099                      # '* 80.00 128.00'
    o = np.cumsum([0] + list(inp.plength))
    names = inp.names
    assert names[:5] == ["Height_l0_rgb"] * 5 and names[o[2]:o[3]] == ["Frequency_RGB_l"] * 5
    assert names[o[3]:o[3] + 8] == ["delta01", "DP1", "alpha_g", "q", "Empty", "Empty", "Wfactor", "Hfactor"]
    assert names[o[3] + 8:o[3] + 24] == ["fref_bias"] * 16 and names[o[3] + 24:o[4]] == ["ferr_bias"] * 16
    assert names[o[6]:o[7]] == ["rot_env", "rot_core", "a2_core", "a2_env", "a3_env", "a4_env", "a5_env", "a6_env", "eta0_switch",
                                "Lorentzian_asymetry"]
    p, pr, pn = inp.params, inp.priors, inp.prior_names
    dnu = 9.54
    assert np.allclose(p[o[2]:o[3]], [87.7427, 96.9129, 106.411, 115.777, 125.34])
    assert pn[o[2]] == "GUG" and np.allclose(pr[:, o[2]], [87.64, 87.84, 0.0025 * dnu, 0.0025 * dnu])          # default GUG wings
    assert pn[o[3]] == "Uniform" and np.isclose(p[o[3]], 0.5 * dnu / 100) and np.allclose(pr[:2, o[3]], [-dnu / 100, dnu / 100])  # delta01 Fix_Auto
    assert np.isclose(p[o[3] + 1], 73.75) and np.allclose(pr[:2, o[3] + 1], [50, 150])                          # DP1
    assert pn[o[3] + 6] == "GU" and np.allclose(pr[:3, o[3] + 6], [0.8, 1.0, 0.1])                               # Wfactor
    assert np.allclose(p[o[3] + 8:o[3] + 24][[0, 1, -1]], [92.2, 93.2, 121.1]) and not inp.relax[o[3] + 8:o[3] + 24].any()
    assert set(pn[o[3] + 24:o[4]]) == {"Gaussian"} and np.allclose(pr[:2, o[3] + 24:o[4]].T, [0, 0.1]) and np.all(p[o[3] + 24:o[4]] == 0)
    assert np.allclose(p[o[4]:o[6]], [95.6329, 105.166, 114.664, 124.161, 98.4849, 108.441])                    # l=2 then l=3 (eigen table)
    # Appourchaux width law started from numax (set_width_App2016_params_v2, io_ms_global.cpp:1625-1720)
    numax, enumax = 113.784460254, 0.220633701471
    w = p[o[7]:o[8]]
    assert np.allclose(w, [numax, numax, abs(4 / 2150 * numax + 1 - 4000 / 2150), abs(0.8 / 2150 * numax + 4.5 - 800 / 2150) / 5,
                           abs(3400 / 2150 * numax + 1000 - 3.4e6 / 2150), abs(2.8 / 2200 * numax + 1 - 2.8 / 2200)])
    assert np.allclose(pr[:2, o[7]], [numax, enumax]) and pn[o[7] + 2] == "Uniform" and np.allclose(pr[:2, o[7] + 4], [w[4], 0.25 * w[4]])
    # noise: two switched-off Harvey profiles, the third + white noise Gaussian around the previous step's fit
    assert np.allclose(p[o[8]:o[9]], [0, 0, 1, 0, 0, 1, 0.6571252, 8.9101468, 4.8003148, 0.0020856]) and list(inp.relax[o[8]:o[9]]) == [0] * 6 + [1] * 4
    assert np.allclose(p[o[10]:], [25.0, 0.0, dnu / 10, 0.0, 1.0, 16.0])            # trunc_c, do_amp, sigma_limit, model_type, bias_type, Nferr
    assert np.allclose(inp.extra_priors[:5], [0.5, 1.0, 0.2, 0.0, 3.0])             # 'freq_smoothness bool 0.5 1.0'; v4 prior switch
    assert inp.relax.sum() == 5 + 3 + 5 + 6 + 16 + 6 + 2 + 6 + 4 + 1
    # the vector drives the oracle's model and prior: finite spectrum above the background, finite prior
    st, m = oracle.call_model(inp.model_id, p, inp.plength, star.x)
    assert st == 0 and np.all(np.isfinite(m)) and m.min() > 0.0020856 and m.max() > 5.0
    lp, st_p = sampler.log_prior(star)
    ref = oracle.call_prior(star)
    assert st_p == 0 and np.isfinite(lp) and np.isclose(lp, ref, rtol=1e-12)
    errs = sampler.default_errors(star)
    assert errs.size == inp.relax.sum() and np.all(errs > 0)


def test_asymptotic_variants_and_refusals(pkg, tmp_path):
    from tamcmc_c_amd import inputs
    nb = inputs.AsymptoticInputs(NOBIAS, 0.00787)
    o = np.cumsum([0] + list(nb.plength))
    # every node 'Fix': no bias whatever bias_type says (io_asymptotic.cpp:462-468)
    assert nb.params[o[10] + 4] == 0.0 and nb.params[o[10] + 5] == 16 and not nb.relax[o[3] + 8:o[4]].any()
    text = open(MODEL).read()

    def variant(name, **subs):
        t = text
        for a, b in subs.values():
            assert a in t
            t = t.replace(a, b)
        q = tmp_path / name
        q.write_text(t)
        return str(q)
    # one-column hyper priors: Uniform bias values, +-Dnu/20 inside and Dnu/2 outwards at the two ends (:350-365)
    lines = text.split("\n")
    one = "\n".join(ln.split()[0] if ("Gaussian" in ln and ln.strip()[0].isdigit()) else ln for ln in lines)
    q = tmp_path / "onecol.model"
    q.write_text(one)
    oc = inputs.AsymptoticInputs(str(q), 0.00787)
    k = o[3] + 24
    assert set(oc.prior_names[k:o[4]]) == {"Uniform"}
    assert np.allclose(oc.priors[:2, k], [-9.54 / 2, 9.54 / 20]) and np.allclose(oc.priors[:2, k + 5], [-9.54 / 20, 9.54 / 20])
    assert np.allclose(oc.priors[:2, o[4] - 1], [-9.54 / 20, 9.54 / 2])
    # constant-width variant: one width = mean of the listed l=0 widths, Jeffreys(resol, Dnu/3) (:633-648); id 27
    ct = inputs.AsymptoticInputs(variant("cte.model", m=("model_RGB_asympt_aj_AppWidth_HarveyLike_v4", "model_RGB_asympt_aj_CteWidth_HarveyLike_v4")), 0.00787)
    assert ct.model_id == 27 and ct.plength[7] == 1
    oc7 = np.cumsum([0] + list(ct.plength))[7]
    assert np.isclose(ct.params[oc7], np.mean([0.803, 0.8035, 0.804, 0.8045, 0.8045])) and ct.prior_names[oc7] == "Jeffreys"
    assert np.allclose(ct.priors[:2, oc7], [0.00787, 9.54 / 3])
    # refusals (the reference exits): no l=1 placeholder, a missing rotation keyword, nodes out of order, model_type without bias_type,
    # a main-sequence model name
    for name, sub in (("nol1.model", ("p  1  100.000000  1      1       1\n", "")),
                      ("norot.model", ("                 rot_core            Uniform            1.00000          0.000000          2.000000    \n", "")),
                      ("order.model", ("93.2000000    Gaussian", "91.2000000    Gaussian")),
                      ("nobt.model", ("          bias_type                       Fix          1.\n", "")),
                      ("ms.model", ("model_RGB_asympt_aj_AppWidth_HarveyLike_v4", "model_MS_Global_aj_HarveyLike"))):
        with pytest.raises(inputs.TamcmcError):
            inputs.AsymptoticInputs(variant(name, s=sub), 0.00787)
    # and the global main-sequence loader still reads a file whose hyper-prior section is empty
    sun = inputs.GlobalInputs(os.path.join(G, "Sun_19992002_incfix_fast_Priorevalrange.model"), 0.01)
    assert sun.model_id == 23
