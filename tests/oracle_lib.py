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
    return lib


class Oracle:
    """Thin numpy-facing wrapper."""

    def __init__(self, fast=False):
        self.lib = load(fast)

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
