"""How far is the device red-giant pre-step from the oracle?  Prints, for the cases of tests/test_gpu_rgb.py, the largest deviation of the
mixed-mode frequencies (muHz), of zeta, of the model rows (relative L2 and per-bin maximum) and of logL (relative).  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import oracle_lib  # noqa: E402

pkg = g.load_package()
from tamcmc_c_amd import synth  # noqa: E402

orc = oracle_lib.Oracle()


def modes_dev(ctx, model_id, params, pl):
    import ctypes as C
    L = pkg.lib()
    p = np.ascontiguousarray(params, dtype=np.float64)
    plc = np.ascontiguousarray(pl, dtype=np.int32)
    nu, ze, hh = np.zeros(4096), np.zeros(4096), np.zeros(4096)
    n = C.c_int(0)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.tamcmc_hip_rgb_mixed_modes.restype = C.c_int
    L.tamcmc_hip_rgb_mixed_modes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.c_int,
                                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    rc = L.tamcmc_hip_rgb_mixed_modes(ctx._h, int(model_id), dp(p), p.size, plc.ctypes.data_as(C.POINTER(C.c_int32)), 4096, dp(nu), dp(ze), dp(hh),
                                      C.byref(n))
    return rc, nu[:n.value], ze[:n.value], hh[:n.value]


def case(name, model_id, P, pl, x, y, T):
    ref, m_o, st_o = orc.loglike_batch(model_id, P, pl, x, y, 1.0, T, want_model=True)
    out = [name]
    for prec in (pkg.PRECISION_STRICT, pkg.PRECISION_FAST):
        ctx = pkg.HipContext(0, precision=prec)
        ctx.set_spectrum(x, y)
        got, m_d, st_d = ctx.loglike_params_batch(model_id, P, pl, T, want_model=True)
        ok = (st_o == 0) & (st_d == 0)
        rel = np.linalg.norm(m_d[ok] - m_o[ok], axis=1) / np.linalg.norm(m_o[ok], axis=1)
        relbin = np.max(np.abs(m_d[ok] - m_o[ok]) / m_o[ok])
        dl = np.max(np.abs(got[ok] / ref[ok] - 1))
        out.append("prec %d: rows L2 %.2e  per-bin %.2e  logL %.2e" % (prec, rel.max(), relbin, dl))
        if prec == pkg.PRECISION_STRICT:
            dnu, dz = 0.0, 0.0
            for b in range(min(P.shape[0], 3)):
                rc, nu, ze, hh = modes_dev(ctx, model_id, P[b], pl)
                rco, mo = orc.rgb_modes(P[b], pl, x[2] - x[1], cte_width=(model_id == synth.MODEL_RGB_CTE_V4))
                if rc == 0 and rco == 0 and mo["fl1"].size == nu.size:
                    dnu = max(dnu, np.max(np.abs(nu - mo["fl1"])) if nu.size else 0.0)
                    dz = max(dz, np.max(np.abs(ze - mo["ksi"])) if nu.size else 0.0)
                else:
                    out.append("   MODE COUNT %d vs %d (rc %d %d)" % (nu.size, mo["fl1"].size if mo else -1, rc, rco))
            out.append("modes: max |d nu| %.2e muHz   max |d zeta| %.2e   (n=%d)" % (dnu, dz, nu.size))
        ctx.close()
    print(" | ".join(out), flush=True)


rng = np.random.default_rng(5)
for bias_type, model_type, cte in ((0, 0, False), (1, 0, False), (2, 1, False), (0, 1, False), (1, 0, True), (0, 1, True)):
    rng = np.random.default_rng(5)
    params, pl = synth.make_params_rgb_model(rng, bias_type=bias_type, model_type=model_type, cte_width=cte)
    model_id = synth.MODEL_RGB_CTE_V4 if cte else synth.MODEL_RGB_V4
    x = 110.0 + 0.05 * np.arange(3400)
    st, m0 = orc.call_model(model_id, params, pl, x)
    y = m0 * np.random.default_rng(2).exponential(1.0, m0.size)
    B = 5
    P = np.tile(params, (B, 1))
    o = np.cumsum([0] + list(pl))
    P[1:, :pl[0]] *= 1 + 0.05 * rng.standard_normal((B - 1, pl[0]))
    P[1:, o[3] + 1] *= 1 + 0.002 * rng.standard_normal(B - 1)
    P[1:, o[3] + 3] *= 1 + 0.05 * rng.standard_normal(B - 1)
    case("basic b%d m%d cte%d" % (bias_type, model_type, cte), model_id, P, pl, x, y, 1.4 ** np.arange(B))

for k, c in enumerate([dict(nmax=3, dnu=12.0, DPl=90.0, q=0.2, step=0.05, B=1), dict(nmax=6, dnu=18.0, DPl=310.0, q=0.9, step=0.05, B=5, alpha_g=0.9),
                       dict(nmax=5, dnu=9.0, DPl=70.0, q=0.05, step=0.2, B=3),
                       dict(nmax=8, dnu=7.0, DPl=75.0, q=0.15, step=0.02, B=2, model_type=1, bias_type=2, nferr=9),
                       dict(nmax=6, dnu=15.0, DPl=85.0, q=0.3, step=0.05, B=4, cte=True, model_type=1)]):
    rng = np.random.default_rng(23)
    cte = c.get("cte", False)
    params, pl = synth.make_params_rgb_model(rng, nmax=c["nmax"], dnu=c["dnu"], DPl=c["DPl"], q=c["q"], alpha_g=c.get("alpha_g", 0.0),
                                             model_type=c.get("model_type", 0), bias_type=c.get("bias_type", 1), nferr=c.get("nferr", 4), cte_width=cte)
    model_id = synth.MODEL_RGB_CTE_V4 if cte else synth.MODEL_RGB_V4
    o = np.cumsum([0] + list(pl))
    fl0 = params[o[2]:o[3]]
    lo = fl0.min() - 1.3 * c["dnu"]
    x = lo + c["step"] * np.arange(int((fl0.max() - fl0.min() + 2.6 * c["dnu"]) / c["step"]))
    B = c["B"]
    P = np.tile(params, (B, 1))
    if B > 1:
        P[1:, o[3] + 1] *= 1 + 0.004 * rng.standard_normal(B - 1)
        P[1:, o[3] + 3] *= 1 + 0.05 * rng.standard_normal(B - 1)
    st, m0 = orc.call_model(model_id, params, pl, x)
    y = m0 * np.random.default_rng(4).exponential(1.0, m0.size)
    case("awkward %d" % k, model_id, P, pl, x, y, 1.1 ** np.arange(B))

star = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
st, m0 = orc.call_model(star.model_id, star.params, star.plength, star.x)
y = star.set_spectrum_from_model(m0, 7)
P = np.tile(star.params, (2, 1))
o = np.cumsum([0] + list(star.plength))
P[1, o[3] + 1] *= 1.0007
case("C5 full size", star.model_id, P, star.plength, star.x, y, np.array([1.0, 1.15]))
