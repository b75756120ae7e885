"""The red-giant pre-step (row N1) pinned on outputs of the REFERENCE's own solver: tests/golden/armm_scanner/out_{0..10}.res are
the files external/ARMM/tests/scanner/out/ holds -- written by external/ARMM/do_solve.cpp:114-121, i.e.
    solve_mm_asymptotic_O2from_l0(fl0, l, delta0l, DPl, alpha_g, q, 0, step, true, false, fmin, fmax)   solver_mm.cpp:624-760
    ksi_fct2(nu_m, nu_p, nu_g, dnu_p, DPg, q, "precise")                                                  bump_DP.cpp:125-188
    h_l_rgb(zeta)                                                                                         bump_DP.cpp:235-254
with every input printed in the `!` header (q = 0, 0.1, ..., 1) and nu_p, dnu_p, nu_g, DPg, nu_m, zeta_pg, H1/H0 printed with six
significant digits.  CPU part: the oracle (oracle/armm_oracle.c) reproduces all of them to the printed precision.  GPU part: the
product's device pre-step (csrc/rgb_prestep.hip) on the same star."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "armm_scanner")


def parse_res(path):
    d = {}
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("!"):
            line = line[1:]
        k, v = line.split("=", 1)
        d[k.strip()] = np.array([float(t) for t in v.split()])
    return d


def fixtures():
    files = sorted(glob.glob(os.path.join(GOLD, "out_*.res")), key=lambda f: int(os.path.basename(f)[4:-4]))
    assert len(files) == 11
    return [(os.path.basename(f), parse_res(f)) for f in files]


def printed_equal(got, ref, digits=6):
    """|got - ref| within the rounding of a number printed with `digits` significant digits (+ a hair for the tie cases)."""
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    if got.shape != ref.shape:
        return False
    ulp = 10.0 ** (np.floor(np.log10(np.maximum(np.abs(ref), 1e-300))) - (digits - 1))
    return bool(np.all(np.abs(got - ref) <= 0.5 * ulp * 1.02 + 1e-15))


def solve_like_do_solve(oracle, d):
    return oracle.armm_solve_O2from_l0(d["fl0"], int(d["l"][0]), float(d["delta0l"][0]), float(d["DPl"][0]), float(d["alpha_g"][0]),
                                       float(d["q_star"][0]), float(d["step"][0]), float(d["fmin"][0]), float(d["fmax"][0]))


def test_oracle_reproduces_the_reference_solver_outputs(oracle):
    seen_q = []
    for name, d in fixtures():
        q = float(d["q_star"][0])
        seen_q.append(q)
        assert (d["step"][0], d["fmin"][0], d["fmax"][0], d["l"][0], d["DPl"][0], d["alpha_g"][0], d["delta0l"][0]) == (0.00814215, 50, 200, 1, 150, 0.5, -1)
        rc, s = solve_like_do_solve(oracle, d)
        assert rc == 0, name
        # the p and g ladders the solver builds from the l=0 list
        assert printed_equal(s["nu_p"], d["nu_p"]) and printed_equal(s["dnup"], d["dnu_p"]), name
        assert printed_equal(s["nu_g"], d["nu_g"]) and printed_equal(s["dPg"], d["DPg"]), name
        # the mixed modes: same count, same frequencies
        assert s["nu_m"].size == d["nu_m"].size, (name, s["nu_m"].size, d["nu_m"].size)
        if q == 0:
            assert s["nu_m"].size == 0          # no coupling: the reference finds nothing either
            continue
        assert printed_equal(s["nu_m"], d["nu_m"]), (name, np.max(np.abs(s["nu_m"] - d["nu_m"])))
        z = oracle.ksi_precise(s["nu_m"], s["nu_p"], s["dnup"], s["nu_g"], s["dPg"], q)
        assert printed_equal(z, d["zeta_pg"]), (name, np.max(np.abs(z - d["zeta_pg"])))
        h = np.sqrt(1.0 - z)                     # h_l_rgb with its default factor 1 (bump_DP.h:112)
        h[np.abs(h) < 1e-5] = 1e-10
        assert printed_equal(h, d["H1/H0"]), (name, np.max(np.abs(h - d["H1/H0"])))
    assert np.allclose(seen_q, np.arange(11) / 10.0)


def scanner_star(synth, d, cte_width=True):
    """A red-giant parameter vector whose pre-step is the fixture's solver call: the l=0 ladder 100..170, delta0l -1, DPl 150,
    alpha_g 0.5, the fixture's q, model_type 1 (= solve_mm_asymptotic_O2from_l0, models.cpp:4861), no bias, Hfactor 1; spectrum grid of
    the fixture's step.  The model keeps the mixed modes inside [min fl0, max fl0] (models.cpp:4861: fmin, fmax of the l=0 list)."""
    rng = np.random.default_rng(1)
    fl0 = d["fl0"]
    params, plength = synth.make_params_rgb_model(rng, nmax=fl0.size, dnu=10.0, n_first=10, delta0l=float(d["delta0l"][0]), DPl=float(d["DPl"][0]),
                                                  alpha_g=float(d["alpha_g"][0]), q=float(d["q_star"][0]), nferr=0, bias_type=0, model_type=1,
                                                  cte_width=cte_width)
    o = np.cumsum([0] + list(plength))
    params[o[2]:o[3]] = fl0
    params[o[3] + 7] = 1.0                       # Hfactor
    params[o[4]:o[5]] = fl0[1:] - 1.2            # l=2, l=3 lists follow the ladder
    params[o[5]:o[6]] = fl0[:-1] + 2.1
    step = float(d["step"][0])
    x = 85.0 + step * np.arange(int(100.0 / step))
    return params, plength, x


def test_oracle_model_prestep_is_the_fixture_inside_the_l0_range(oracle, synth):
    """orc_rgb_v4_cte_modes (models.cpp:4377-4470 + the solver) on the fixture's star returns the fixture's modes inside [100, 170]."""
    for name, d in fixtures()[1::3]:
        params, plength, x = scanner_star(synth, d)
        rc, md = oracle.rgb_modes(params, plength, x[2] - x[1], cte_width=True)
        assert rc == 0
        keep = (d["nu_m"] >= 100.0) & (d["nu_m"] <= 170.0)
        assert printed_equal(md["fl1"], d["nu_m"][keep]) and printed_equal(md["ksi"], d["zeta_pg"][keep]), name


@pytest.mark.gpu
def test_device_prestep_reproduces_the_reference_solver_outputs(pkg, oracle, synth):
    """The product path: k_armm_scan -> k_armm_sort_unique -> k_zeta on the fixture's star, read back through
    tamcmc_hip_rgb_mixed_modes: frequencies, zeta and H1/H0 equal the reference's printed values (six significant digits) for every q,
    on both red-giant model ids; and equal the oracle's far below that."""
    for model_cte in (True, False):
        for name, d in fixtures():
            q = float(d["q_star"][0])
            params, plength, x = scanner_star(synth, d, cte_width=model_cte)
            ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
            ctx.set_spectrum(x, np.ones_like(x))
            mid = pkg.MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4 if model_cte else pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4
            if q == 0:
                # q = 0: zeta's front factor divides by q -- the reference's model would propagate NaN; the solver part finds no mode
                nu, z, h = ctx.rgb_mixed_modes(mid, params, plength)
                assert nu.size == 0
                ctx.close()
                continue
            nu, z, h = ctx.rgb_mixed_modes(mid, params, plength)
            ctx.close()
            keep = (d["nu_m"] >= 100.0) & (d["nu_m"] <= 170.0)
            assert nu.size == int(keep.sum()), (name, nu.size, int(keep.sum()))
            assert printed_equal(nu, d["nu_m"][keep]), (name, np.max(np.abs(nu - d["nu_m"][keep])))
            assert printed_equal(z, d["zeta_pg"][keep]), (name, np.max(np.abs(z - d["zeta_pg"][keep])))
            assert printed_equal(h, d["H1/H0"][keep]), (name, np.max(np.abs(h - d["H1/H0"][keep])))
            rc, md = oracle.rgb_modes(params, plength, x[2] - x[1], cte_width=model_cte)
            assert rc == 0 and md["fl1"].size == nu.size
            assert np.max(np.abs(nu - md["fl1"])) < 1e-10 and np.max(np.abs(z - md["ksi"])) < 1e-10   # (muHz; red-giant tolerance, include/tamcmc_hip.h)
