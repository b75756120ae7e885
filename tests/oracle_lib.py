"""ctypes binding of the CPU oracle (oracle/libtamcmc_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load(fast=False):
    name = "libtamcmc_oracle_fast.so" if fast else "libtamcmc_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    ld = C.c_longdouble
    lib.orc_Pslm.restype = ld
    lib.orc_Pslm.argtypes = [C.c_int] * 3
    lib.orc_Qlm.restype = C.c_double
    lib.orc_Qlm.argtypes = [C.c_int] * 2
    lib.orc_amplitude_ratio.restype = None
    lib.orc_amplitude_ratio.argtypes = [C.c_int, C.c_double, c_dp]
    lib.orc_lin_interpol.restype = C.c_double
    lib.orc_lin_interpol.argtypes = [c_dp, c_dp, C.c_long, C.c_double]
    lib.orc_linfit.restype = None
    lib.orc_linfit.argtypes = [c_dp, c_dp, C.c_long, c_dp]
    lib.orc_eta0_from_dnu.restype = C.c_double
    lib.orc_eta0_from_dnu.argtypes = [C.c_double]
    lib.orc_eta0_fct.restype = C.c_double
    lib.orc_eta0_fct.argtypes = [c_dp, C.c_long]
    lib.orc_set_imin_imax.restype = C.c_int
    lib.orc_set_imin_imax.argtypes = [c_dp, C.c_long, C.c_int] + [C.c_double] * 5 + [c_ip]
    lib.orc_nu_nlm_aj.restype = C.c_double
    lib.orc_nu_nlm_aj.argtypes = [C.c_double] * 8 + [C.c_int] * 2
    lib.orc_nu_nlm_a1etaa3.restype = C.c_double
    lib.orc_nu_nlm_a1etaa3.argtypes = [C.c_double] * 4 + [C.c_int] * 2
    lib.orc_build_l_mode_aj.restype = None
    lib.orc_build_l_mode_aj.argtypes = [c_dp, C.c_long] + [C.c_double] * 11 + [C.c_int, c_dp, c_dp]
    lib.orc_build_l_mode_a1etaa3.restype = None
    lib.orc_build_l_mode_a1etaa3.argtypes = [c_dp, C.c_long] + [C.c_double] * 7 + [C.c_int, c_dp, c_dp]
    lib.orc_harvey_like.restype = None
    lib.orc_harvey_like.argtypes = [c_dp, C.c_long, c_dp, c_dp, C.c_long, C.c_int]
    lib.orc_likelihood_chi22p.restype = ld
    lib.orc_likelihood_chi22p.argtypes = [c_dp, c_dp, C.c_long, C.c_long]
    lib.orc_likelihood_chi22p_ld.restype = ld
    lib.orc_likelihood_chi22p_ld.argtypes = [c_dp, c_dp, C.c_long, C.c_long]
    lib.orc_call_model.restype = C.c_int
    lib.orc_call_model.argtypes = [C.c_int, c_dp, c_ip, c_dp, C.c_long, c_dp]
    lib.orc_call_likelihood.restype = C.c_double
    lib.orc_call_likelihood.argtypes = [c_dp, c_dp, C.c_long, C.c_double, C.c_double]
    lib.orc_loglike_batch.restype = C.c_int
    lib.orc_loglike_batch.argtypes = [C.c_int, C.c_int, c_dp, C.c_long, c_ip, c_dp, c_dp, C.c_long, C.c_double,
                                      c_dp, c_dp, c_dp, c_ip]
    lib.orc_fd_gradient.restype = C.c_int
    lib.orc_fd_gradient.argtypes = [C.c_int, c_dp, C.c_long, c_ip, c_ip, C.c_int, c_dp, c_dp, c_dp, C.c_long,
                                    C.c_double, C.c_double, c_dp, c_dp]
    for nm, n in (("orc_logP_uniform", 3), ("orc_logP_gaussian", 3), ("orc_logP_jeffrey", 3),
                  ("orc_logP_uniform_abs", 3), ("orc_logP_jeffrey_abs", 3), ("orc_logP_gaussian_uniform", 4),
                  ("orc_logP_uniform_gaussian", 4), ("orc_logP_gug", 5)):
        f = getattr(lib, nm)
        f.restype = ld
        f.argtypes = [C.c_double] * n
    lib.orc_apply_generic_priors.restype = ld
    lib.orc_apply_generic_priors.argtypes = [c_dp, C.c_long, C.c_long, c_dp, C.c_long, c_ip]
    lib.orc_call_prior.restype = C.c_double
    lib.orc_call_prior.argtypes = [C.c_int, c_dp, c_ip, c_dp, c_ip, c_dp]
    # red-giant model and pre-step (armm_oracle.c)
    class EigenSols(C.Structure):
        _fields_ = [("n_m", C.c_long), ("n_p", C.c_long), ("n_g", C.c_long), ("nu_m", c_dp), ("nu_p", c_dp), ("nu_g", c_dp),
                    ("dnup", c_dp), ("dPg", c_dp)]

    class RgbModes(C.Structure):
        _fields_ = [("N0", C.c_long), ("N1", C.c_long), ("fl0", c_dp), ("Wl0", c_dp), ("Hl0", c_dp), ("fl1", c_dp), ("Wl1", c_dp),
                    ("Hl1", c_dp), ("a1_l1", c_dp), ("ksi", c_dp), ("g", C.c_double * 6)]

    lib.EigenSols, lib.RgbModes = EigenSols, RgbModes
    lib.orc_eigensols_free.restype = None
    lib.orc_eigensols_free.argtypes = [C.POINTER(EigenSols)]
    lib.orc_armm_solve_O2p.restype = C.c_int
    lib.orc_armm_solve_O2p.argtypes = [C.c_double, C.c_double, C.c_int] + [C.c_double] * 9 + [C.POINTER(EigenSols)]
    lib.orc_armm_solve_O2from_l0.restype = C.c_int
    lib.orc_armm_solve_O2from_l0.argtypes = [c_dp, C.c_long, C.c_int] + [C.c_double] * 7 + [C.POINTER(EigenSols)]
    lib.orc_ksi_fct2_precise.restype = None
    lib.orc_ksi_fct2_precise.argtypes = [c_dp, C.c_long, c_dp, c_dp, C.c_long, c_dp, c_dp, C.c_long, C.c_double, c_dp]
    lib.orc_spline_eval.restype = C.c_double
    lib.orc_spline_eval.argtypes = [c_dp, c_dp, C.c_long, C.c_int, C.c_double]
    lib.orc_rgb_v4_modes.restype = C.c_int
    lib.orc_rgb_v4_modes.argtypes = [c_dp, c_ip, C.c_double, C.POINTER(RgbModes)]
    lib.orc_quad_interpol.restype = None
    lib.orc_quad_interpol.argtypes = [c_dp, C.c_long, C.c_long, c_dp]
    lib.orc_evidence_calc.restype = C.c_double
    lib.orc_evidence_calc.argtypes = [c_dp, C.c_long, c_dp, C.c_long, C.c_int, c_dp, c_dp, c_dp, c_dp]
    lib.orc_rgb_v4_cte_modes.restype = C.c_int
    lib.orc_rgb_v4_cte_modes.argtypes = [c_dp, c_ip, C.c_double, C.POINTER(RgbModes)]
    lib.orc_rgb_modes_free.restype = None
    lib.orc_rgb_modes_free.argtypes = [C.POINTER(RgbModes)]

    # one sampler iteration with explicit draws (sampler_oracle.c)
    class SamplerStar(C.Structure):
        _fields_ = [("model_id", C.c_int), ("prior_class", C.c_int), ("Nparams", C.c_long), ("Nvars", C.c_long), ("Nx", C.c_long),
                    ("Nchains", C.c_long), ("plength", c_ip), ("index_to_relax", c_ip), ("priors_switch", c_ip), ("priors", c_dp),
                    ("extra_priors", c_dp), ("x", c_dp), ("y", c_dp), ("Tcoefs", c_dp), ("init_logL", c_dp),
                    ("likelihood_params", C.c_double), ("epsilon1", C.c_double), ("epsi2", C.c_double), ("A1", C.c_double),
                    ("target_acceptance", C.c_double), ("c0", C.c_double)]

    lib.SamplerStar = SamplerStar
    lib.orc_p1_fct.restype = ld
    lib.orc_p1_fct.argtypes = [ld, ld, ld]
    lib.orc_p2_fct.restype = None
    lib.orc_p2_fct.argtypes = [c_dp, C.c_long, C.c_double]
    lib.orc_p3_fct.restype = None
    lib.orc_p3_fct.argtypes = [c_dp, C.c_long, C.c_double]
    lib.orc_update_proposal.restype = None
    lib.orc_update_proposal.argtypes = [c_dp, c_dp, c_dp, c_dp, C.c_long, ld, ld, ld, ld, ld]
    lib.orc_new_prop_values.restype = C.c_int
    lib.orc_new_prop_values.argtypes = [c_dp, C.c_double, C.c_double, c_dp, c_dp, C.c_long, c_dp, c_dp]
    lib.orc_mh_accept.restype = C.c_int
    lib.orc_mh_accept.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, c_dp]
    lib.orc_parallel_tempering.restype = C.c_int
    lib.orc_parallel_tempering.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp, C.c_long, C.c_long, C.c_int, C.c_double, C.c_int, c_dp]
    lib.orc_learn_at.restype = C.c_int
    lib.orc_learn_at.argtypes = [C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_long]
    lib.orc_sampler_iteration.restype = C.c_int
    lib.orc_sampler_iteration.argtypes = [C.POINTER(SamplerStar), C.c_long, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, c_dp, c_dp, c_dp, c_dp,
                                          c_dp, c_dp, c_dp, c_ip, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp]
    # the Langevin step (sampler_oracle.c, second half)
    lib.orc_mvn_logpdf.restype = ld
    lib.orc_mvn_logpdf.argtypes = [c_dp, c_dp, c_dp, C.c_long]
    lib.orc_fd_gradient_posterior.restype = C.c_int
    lib.orc_fd_gradient_posterior.argtypes = [C.POINTER(SamplerStar), c_dp, C.c_double, c_dp, c_dp, c_dp]
    lib.orc_langevin_drift.restype = None
    lib.orc_langevin_drift.argtypes = [c_dp, C.c_double, C.c_double, C.c_double, c_dp, C.c_long, c_dp]
    lib.orc_langevin_iteration.restype = C.c_int
    lib.orc_langevin_iteration.argtypes = [C.POINTER(SamplerStar), C.c_long, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, c_dp, c_dp, C.c_double,
                                           C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp, c_dp, c_ip, c_dp]
    return lib


class Oracle:
    """Thin numpy-facing wrapper."""

    def __init__(self, fast=False):
        self.lib = load(fast)

    # ---- red-giant pre-step (armm_oracle.c) ----
    def _sols(self, e):
        take = lambda p, n: np.array([p[i] for i in range(n)])
        out = dict(nu_m=take(e.nu_m, e.n_m), nu_p=take(e.nu_p, e.n_p), nu_g=take(e.nu_g, e.n_g), dnup=take(e.dnup, e.n_p),
                   dPg=take(e.dPg, e.n_g))
        self.lib.orc_eigensols_free(C.byref(e))
        return out

    def armm_solve_O2p(self, Dnu_p, epsilon, el, delta0l, alpha_p, nmax, DPl, alpha, q, fmin, fmax, resol):
        e = self.lib.EigenSols()
        rc = self.lib.orc_armm_solve_O2p(Dnu_p, epsilon, el, delta0l, alpha_p, nmax, DPl, alpha, q, fmin, fmax, resol, C.byref(e))
        return rc, (self._sols(e) if rc == 0 else None)

    def armm_solve_O2from_l0(self, nu_l0, el, delta0l, DPl, alpha, q, resol, fmin, fmax):
        e = self.lib.EigenSols()
        v = np.ascontiguousarray(nu_l0, dtype=np.float64)
        rc = self.lib.orc_armm_solve_O2from_l0(_dp(v), v.size, el, delta0l, DPl, alpha, q, resol, fmin, fmax, C.byref(e))
        return rc, (self._sols(e) if rc == 0 else None)

    def ksi_precise(self, nu, nu_p, dnup, nu_g, dPg, q):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (nu, nu_p, dnup, nu_g, dPg)]
        out = np.zeros(a[0].size)
        self.lib.orc_ksi_fct2_precise(_dp(a[0]), a[0].size, _dp(a[1]), _dp(a[2]), a[1].size, _dp(a[3]), _dp(a[4]), a[3].size, q, _dp(out))
        return out

    def spline_eval(self, xn, yn, kind, x):
        xn, yn = np.ascontiguousarray(xn, dtype=np.float64), np.ascontiguousarray(yn, dtype=np.float64)
        return np.array([self.lib.orc_spline_eval(_dp(xn), _dp(yn), xn.size, kind, float(v)) for v in np.atleast_1d(x)])

    def quad_interpol(self, a, m):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.zeros(m)
        self.lib.orc_quad_interpol(_dp(a), a.size, m, _dp(b))
        return b

    def evidence(self, Tcoefs, logL, interp_factor):
        T = np.ascontiguousarray(Tcoefs, dtype=np.float64)
        Lk = np.ascontiguousarray(logL, dtype=np.float64)
        n = T.size
        beta, Lb, bi, Li = np.zeros(n), np.zeros(n), np.zeros(n * interp_factor), np.zeros(n * interp_factor)
        ev = self.lib.orc_evidence_calc(_dp(T), n, _dp(Lk), Lk.shape[0], interp_factor, _dp(beta), _dp(Lb), _dp(bi), _dp(Li))
        return ev, beta, Lb, bi, Li

    def rgb_modes(self, params, plength, step, cte_width=False):
        p, pl = np.ascontiguousarray(params, dtype=np.float64), np.ascontiguousarray(plength, dtype=np.int32)
        m = self.lib.RgbModes()
        rc = (self.lib.orc_rgb_v4_cte_modes if cte_width else self.lib.orc_rgb_v4_modes)(_dp(p), _ip(pl), float(step), C.byref(m))
        if rc != 0:
            return rc, None
        take = lambda q, n: np.array([q[i] for i in range(n)])
        out = dict(fl0=take(m.fl0, m.N0), Wl0=take(m.Wl0, m.N0), Hl0=take(m.Hl0, m.N0), fl1=take(m.fl1, m.N1), Wl1=take(m.Wl1, m.N1),
                   Hl1=take(m.Hl1, m.N1), a1_l1=take(m.a1_l1, m.N1), ksi=take(m.ksi, m.N1), g=np.array(list(m.g)))
        self.lib.orc_rgb_modes_free(C.byref(m))
        return rc, out

    # ---- one sampler iteration with explicit draws (sampler_oracle.c) ----
    def _sampler_star(self, star, y, Tcoefs, init_logL, nparams, nvars, p, epsilon1, epsilon2, A1, target_acceptance, c0):
        f = lambda a: np.ascontiguousarray(np.array(a, dtype=np.float64, copy=True))
        T = f(Tcoefs)
        keep = dict(pl=np.ascontiguousarray(star.plength, dtype=np.int32), idx=np.ascontiguousarray(star.index_to_relax, dtype=np.int32),
                    sw=np.ascontiguousarray(star.priors_switch, dtype=np.int32), pr=f(star.priors), ex=f(np.resize(np.append(star.extra_priors, np.zeros(10)), 10)),
                    x=f(star.x), y=f(y), T=T, il=f(init_logL if init_logL is not None else np.zeros(T.size)))
        S = self.lib.SamplerStar(int(star.model_id), int(star.prior_class), int(nparams), int(nvars), keep["x"].size, T.size, _ip(keep["pl"]),
                                 _ip(keep["idx"]), _ip(keep["sw"]), _dp(keep["pr"]), _dp(keep["ex"]), _dp(keep["x"]), _dp(keep["y"]), _dp(keep["T"]),
                                 _dp(keep["il"]), float(p), epsilon1, epsilon2, A1, target_acceptance, c0)
        return S, keep

    def sampler_iteration(self, star, y, Tcoefs, init_logL, state, law, i, z, u_mh, learn=False, do_swap=False, ind_A=0, u_swap=1.0,
                          literal_444=False, p=1.0, epsilon1=1e-12, epsilon2=1e-12, A1=1e14, target_acceptance=0.234, c0=10.0,
                          use_drift=False, fd_step_rel=1e-7, delta=0.0, chain_mask=None, prop_given=None):
        """One pass of MALA::execute's loop body.  state = dict(params, vars, logL, logPrior, logPost) [copied], law = (mu, cov, sigma)
        [copied].  Returns (new state incl. moved / Pmove / swapped / prop_vars / prop_stats, new law, rc).
        use_drift: the Langevin step (orc_langevin_iteration); the state then also carries diag [Nchains x 4] = log q(x'|x), log q(x|x'),
        |drift(x)|, |drift(x')|; chain_mask [Nchains]: advance the flagged chains only; prop_given [Nchains x Nvars]: test at these
        proposals (prop_vars still returns the oracle's own)."""
        f = lambda a: np.ascontiguousarray(np.array(a, dtype=np.float64, copy=True))
        st = {k: f(state[k]) for k in ("params", "vars", "logL", "logPrior", "logPost")}
        mu, cov, sigma = f(law[0]), f(law[1]), f(law[2])
        C_, Nv = st["vars"].shape
        S, keep = self._sampler_star(star, y, Tcoefs, init_logL, st["params"].shape[1], Nv, p, epsilon1, epsilon2, A1, target_acceptance, c0)
        z, u = f(z), f(u_mh)
        moved, Pmove, swapped = np.zeros(C_, dtype=np.int32), np.zeros(C_), np.zeros(1, dtype=np.int32)
        pv, ps = np.zeros((C_, Nv)), np.zeros((C_, 3))
        if use_drift:
            diag = np.zeros((C_, 4))
            rc = self.lib.orc_langevin_iteration(C.byref(S), int(i), int(learn), int(do_swap), int(ind_A), float(u_swap), int(literal_444), _dp(z),
                                                 _dp(u), float(fd_step_rel), float(delta), _dp(st["params"]), _dp(st["vars"]), _dp(st["logL"]),
                                                 _dp(st["logPrior"]), _dp(st["logPost"]), _ip(moved), _dp(Pmove), _dp(mu), _dp(cov), _dp(sigma),
                                                 _ip(swapped), _dp(pv), _dp(ps), _dp(diag),
                                                 _ip(np.ascontiguousarray(chain_mask, dtype=np.int32)) if chain_mask is not None else None,
                                                 _dp(f(prop_given)) if prop_given is not None else None)
            st["diag"] = diag
        else:
            rc = self.lib.orc_sampler_iteration(C.byref(S), int(i), int(learn), int(do_swap), int(ind_A), float(u_swap), int(literal_444), _dp(z), _dp(u),
                                                _dp(st["params"]), _dp(st["vars"]), _dp(st["logL"]), _dp(st["logPrior"]), _dp(st["logPost"]), _ip(moved),
                                                _dp(Pmove), _dp(mu), _dp(cov), _dp(sigma), _ip(swapped), _dp(pv), _dp(ps))
        st.update(moved=moved, Pmove=Pmove, swapped=int(swapped[0]), prop_vars=pv, prop_stats=ps)
        return st, (mu, cov, sigma), rc

    def mvn_logpdf(self, v, mean, M):
        v, mean, M = (np.ascontiguousarray(a, dtype=np.float64) for a in (v, mean, M))
        return float(self.lib.orc_mvn_logpdf(_dp(v), _dp(mean), _dp(M), v.size))

    def langevin_drift(self, cov, sigma, epsi2, delta, grad):
        cov, grad = np.ascontiguousarray(cov, dtype=np.float64), np.ascontiguousarray(grad, dtype=np.float64)
        out = np.zeros(grad.size)
        self.lib.orc_langevin_drift(_dp(cov), float(sigma), float(epsi2), float(delta), _dp(grad), grad.size, _dp(out))
        return out

    def fd_gradient_posterior(self, star, y, params, Tcoef, h, p=1.0):
        """(status of the base point, gradient of logL/T + logPrior, the prior's share) by forward differences with the steps h."""
        params = np.ascontiguousarray(params, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64)
        S, keep = self._sampler_star(star, y, [Tcoef], None, params.size, h.size, p, 1e-12, 1e-12, 1e14, 0.234, 10.0)
        g, gp = np.zeros(h.size), np.zeros(h.size)
        st = self.lib.orc_fd_gradient_posterior(C.byref(S), _dp(params), float(Tcoef), _dp(h), _dp(g), _dp(gp))
        return st, g, gp

    def amplitude_ratio(self, l, inc_deg):
        v = np.zeros(2 * l + 1)
        self.lib.orc_amplitude_ratio(l, float(inc_deg), _dp(v))
        return v

    def set_imin_imax(self, x, l, fc, gamma, f_s, c, step):
        x = np.ascontiguousarray(x, dtype=np.float64)
        iv = np.zeros(2, dtype=np.int32)
        st = self.lib.orc_set_imin_imax(_dp(x), x.size, l, fc, gamma, f_s, c, step, _ip(iv))
        return st, int(iv[0]), int(iv[1])

    def call_model(self, model_id, params, plength, x):
        params = np.ascontiguousarray(params, dtype=np.float64)
        plength = np.ascontiguousarray(plength, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        m = np.zeros(x.size)
        st = self.lib.orc_call_model(model_id, _dp(params), _ip(plength), _dp(x), x.size, _dp(m))
        return st, m

    def loglike_batch(self, model_id, params, plength, x, y, p=1.0, Tcoefs=None, want_model=False):
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.ndim == 1:
            params = params[None, :]
        B, Np = params.shape
        plength = np.ascontiguousarray(plength, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        T = np.ones(B) if Tcoefs is None else np.ascontiguousarray(Tcoefs, dtype=np.float64)
        logL = np.zeros(B)
        model = np.zeros((B, x.size)) if want_model else None
        status = np.zeros(B, dtype=np.int32)
        self.lib.orc_loglike_batch(model_id, B, _dp(params), Np, _ip(plength), _dp(x), _dp(y), x.size, float(p),
                                   _dp(T), _dp(logL), _dp(model), _ip(status))
        return logL, model, status

    def chi22p_ld(self, y, model, p=1):
        y = np.ascontiguousarray(y, dtype=np.float64)
        model = np.ascontiguousarray(model, dtype=np.float64)
        return float(self.lib.orc_likelihood_chi22p_ld(_dp(y), _dp(model), y.size, int(p)))

    def chi22p(self, y, model, p=1):
        y = np.ascontiguousarray(y, dtype=np.float64)
        model = np.ascontiguousarray(model, dtype=np.float64)
        return float(self.lib.orc_likelihood_chi22p(_dp(y), _dp(model), y.size, int(p)))

    def fd_gradient(self, model_id, params, plength, index_to_relax, hstep, x, y, p=1.0, Tcoef=1.0):
        params = np.ascontiguousarray(params, dtype=np.float64)
        plength = np.ascontiguousarray(plength, dtype=np.int32)
        idx = np.ascontiguousarray(index_to_relax, dtype=np.int32)
        h = np.ascontiguousarray(hstep, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        g = np.zeros(idx.size)
        l0 = C.c_double(0.0)
        st = self.lib.orc_fd_gradient(model_id, _dp(params), params.size, _ip(plength), _ip(idx), idx.size, _dp(h),
                                      _dp(x), _dp(y), x.size, float(p), float(Tcoef), C.byref(l0), _dp(g))
        return st, l0.value, g

    def call_prior(self, star, params=None):
        p = np.ascontiguousarray(star.params if params is None else params, dtype=np.float64)
        pl = np.ascontiguousarray(star.plength, dtype=np.int32)
        pr = np.ascontiguousarray(star.priors, dtype=np.float64)
        sw = np.ascontiguousarray(star.priors_switch, dtype=np.int32)
        ex = np.ascontiguousarray(star.extra_priors, dtype=np.float64)
        return float(self.lib.orc_call_prior(int(star.prior_class), _dp(p), _ip(pl), _dp(pr), _ip(sw), _dp(ex)))
