"""The mixed-mode solver against a file the reference's own synthetic-star generator wrote: external/ARMM/tests/make_star/out/star_params.in
(copied as data under tests/golden/make_star/).  Its header lists every input of the call
    solve_mm_asymptotic_O2p(Dnu_star, epsilon_star, el = 1, delta0l_star = -el (el + 1) delta0l_percent / 100, alpha_p, nmax_star, DPl, alpha_g,
                            q, sigma_p, fmin, fmax, resol = 1e6 / (4 x 365 x 86400))                bump_DP.cpp:636, :697-700
and its mode list holds the l = 1 frequencies that call returned with TEN significant digits (the scanner files of
tests/test_armm_scanner_fixtures.py print six).  Pinned here: the oracle's solver (oracle/armm_oracle.c), the oracle's v4 red-giant model
on an equivalent parameter vector (its own linear fit of the l = 0 ladder, ladders, first-g-mode search, solver), and -- on the GPU --
the device pre-step (csrc/rgb_prestep.hip) through tamcmc_hip_rgb_mixed_modes."""
import os

import numpy as np
import pytest

FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_star", "star_params.in")


def _read():
    hdr, p, gm, modes = {}, [], [], []
    section = None
    for ln in open(FILE):
        s = ln.strip()
        if not s:
            continue
        if s.startswith("#"):
            if "numax and min and max" in s:
                section = "range"
            elif "l=1 p and g modes" in s:
                section = "ladders"
            elif "Input mode parameters. degree" in s:
                section = "modes"
            elif "H0 , tau_0" in s:
                section = "noise"
            continue
        if section is None and "=" in s:
            k, v = s.split("=", 1)
            hdr[k.strip()] = v.strip()
        elif section == "range":
            hdr["range"] = [float(t) for t in s.split()]
        elif section == "ladders":
            t = s.split()
            (p if t[0] == "p" else gm).append(float(t[2]))
        elif section == "modes":
            t = s.split()
            modes.append((int(t[0]), float(t[1]), t[1]))
    return hdr, np.array(p), np.array(gm), modes


def _printed_equal(values, printed_strings):
    """Does each value, written with the digits the file shows, give the file's text?"""
    for v, txt in zip(values, printed_strings):
        digits = len(txt.replace(".", "").lstrip("0"))
        if float("%.*g" % (digits, v)) != float(txt):
            return False
    return True


def _inputs(hdr):
    Dnu, eps = float(hdr["Dnu_star"]), float(hdr["epsilon_p"])
    d0l = -1 * 2 * float(hdr["delta0l_percent"]) / 100.0                       # bump_DP.cpp:697 with el = 1
    numax, fmin, fmax, nmax = hdr["range"]
    resol = 1e6 / (4 * 365.0 * 86400.0)                                          # bump_DP.cpp:636
    return dict(Dnu=Dnu, eps=eps, d0l=d0l, DPl=float(hdr["DPl"]), alpha_g=float(hdr["epsilon_g"]), q=float(hdr["q_star"]), fmin=fmin, fmax=fmax,
                nmax=nmax, resol=resol)


def test_oracle_solver_reproduces_the_generators_mixed_modes(oracle):
    hdr, p_listed, g_listed, modes = _read()
    assert hdr["alpha_p"] == "None" and float(hdr["beta_p"]) == 0            # no curvature: nmax_star does not enter
    c = _inputs(hdr)
    l1 = [(f, txt) for (l, f, txt) in modes if l == 1]
    assert len(l1) == 15 and len(p_listed) == 15 and len(g_listed) == 3
    rc, sol = oracle.armm_solve_O2p(c["Dnu"], c["eps"], 1, c["d0l"], 0.0, c["nmax"], c["DPl"], c["alpha_g"], c["q"], c["fmin"], c["fmax"], c["resol"])
    assert rc == 0
    # the ladders the file lists (asympt_nu_p, asympt_nu_g: solver_mm.cpp:200-260)
    assert np.allclose(sol["nu_p"][:15], p_listed, rtol=0, atol=5e-10) and _printed_equal(sol["nu_g"][:3], ["2597.4026", "1360.5442", "921.65899"])
    # every listed mixed mode is among the solutions, to the last printed digit (ten significant)
    got = np.array([sol["nu_m"][np.argmin(np.abs(sol["nu_m"] - f))] for f, _ in l1])
    assert _printed_equal(got, [t for _, t in l1]), np.max(np.abs(got - np.array([f for f, _ in l1])))
    assert np.max(np.abs(got - np.array([f for f, _ in l1]))) < 5e-7
    # (the generator keeps as many l = 1 modes as it made l = 0 modes; the solver's next root, 1459.96, is beyond them)
    assert sol["nu_m"].size == 16


def _v4_vector(synth, c):
    class Flat:                                                                  # no scatter of the l = 0 ladder: it IS (n + eps) Dnu
        def uniform(self, a, b, n):
            return np.zeros(n)
    return synth.make_params_rgb_model(Flat(), nmax=15, dnu=c["Dnu"], epsilon=c["eps"], n_first=12, delta0l=c["d0l"], DPl=c["DPl"], alpha_g=c["alpha_g"],
                                       q=c["q"], model_type=0, bias_type=0)


def test_oracle_v4_model_prestep_reproduces_them_too(oracle, synth):
    """The same star as a parameter vector of model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (models.cpp:4684-5079): Dnu and epsilon now come
    out of the model's own linear fit of the l = 0 frequencies (665.5 ... 1435.5, the file's l = 0 rows), the p ladder, the g ladder and
    the search zones out of its unpacking -- the mixed modes must still be the file's."""
    hdr, _, _, modes = _read()
    c = _inputs(hdr)
    params, pl = _v4_vector(synth, c)
    o = np.cumsum([0] + list(pl))
    assert _printed_equal(params[o[2]:o[3]], [t for (l, f, t) in modes if l == 0])
    rc, m = oracle.rgb_modes(params, pl, c["resol"])
    assert rc == 0
    l1 = [(f, txt) for (l, f, txt) in modes if l == 1]
    got = np.array([m["fl1"][np.argmin(np.abs(m["fl1"] - f))] for f, _ in l1])
    assert _printed_equal(got, [t for _, t in l1]), np.max(np.abs(got - np.array([f for f, _ in l1])))


@pytest.mark.gpu
def test_device_prestep_reproduces_the_generators_mixed_modes(pkg, oracle, synth):
    hdr, _, _, modes = _read()
    c = _inputs(hdr)
    params, pl = _v4_vector(synth, c)
    o = np.cumsum([0] + list(pl))
    fl0 = params[o[2]:o[3]]
    x = fl0.min() - 60.0 + c["resol"] * np.arange(int((fl0.max() - fl0.min() + 120.0) / c["resol"]))    # the solver's step is the grid's (models.cpp:4719)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(x, np.ones_like(x))
    nu, z, h = ctx.rgb_mixed_modes(pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4, params, pl)
    ctx.close()
    l1 = [(f, txt) for (l, f, txt) in modes if l == 1]
    got = np.array([nu[np.argmin(np.abs(nu - f))] for f, _ in l1])
    assert _printed_equal(got, [t for _, t in l1]), np.max(np.abs(got - np.array([f for f, _ in l1])))
    rc, m = oracle.rgb_modes(params, pl, x[2] - x[1])
    assert rc == 0 and m["fl1"].size == nu.size and np.max(np.abs(nu - m["fl1"])) < 1e-10
