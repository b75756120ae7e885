"""ctypes binding of include/tamcmc_sampler.h (the host-side mirror of the reference's MALA + Model_def)."""
import ctypes as C

import numpy as np

from . import ABI, EXTRA_ABI, HipContext, TamcmcError, OK, lib, _dp, _ip, _vp, _f64, _i32, _p

_i64p = C.POINTER(C.c_int64)


class SamplerConfig(C.Structure):
    """struct tamcmc_sampler_config"""
    _fields_ = [("model_id", C.c_int32), ("prior_class", C.c_int32), ("likelihood_id", C.c_int32), ("use_drift", C.c_int32),
                ("likelihood_params", C.c_double), ("Nparams", C.c_int64), ("inputs", _dp), ("relax", _ip), ("plength", _ip),
                ("priors", _dp), ("priors_switch", _ip), ("extra_priors", _dp), ("n_extra", C.c_int32),
                ("Nchains", C.c_int32), ("lambda_temp", C.c_double), ("target_acceptance", C.c_double), ("c0", C.c_double),
                ("epsilon1", C.c_double), ("epsilon2", C.c_double), ("A1", C.c_double), ("delta", C.c_double),
                ("delta_x", C.c_double), ("Nt_learn", _i64p), ("periods_learn", _i64p), ("n_Nt_learn", C.c_int32),
                ("engine", C.c_int32), ("dN_mixing", C.c_int64), ("init_errors", _dp), ("seed", C.c_uint64),
                ("fd_step_rel", C.c_double), ("chain_groups", C.c_int32), ("swap_rule", C.c_int32)]


EXTRA_ABI += [
    ("tamcmc_sampler_create", C.c_int, [C.POINTER(_vp), _vp, C.POINTER(SamplerConfig)]),
    ("tamcmc_sampler_destroy", None, [_vp]),
    ("tamcmc_sampler_nvars", C.c_int64, [_vp]),
    ("tamcmc_sampler_get_info", C.c_int, [_vp, _i64p, C.c_int32]),
    ("tamcmc_sampler_get_move_counts", C.c_int, [_vp, _i64p]),
    ("tamcmc_sampler_get_gradient", C.c_int, [_vp, _dp, _dp, _ip]),
    ("tamcmc_sampler_get_last_test", C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    ("tamcmc_outputs_write_acceptance", C.c_int, [C.c_char_p, C.c_double, _dp, C.c_int32, C.c_int32]),
    ("tamcmc_outputs_read_acceptance", C.c_int, [C.c_char_p, _ip, C.c_int64, _i64p, _dp, _dp]),
    ("tamcmc_sampler_run", C.c_int, [_vp, C.c_int64, _dp, _dp]),
    ("tamcmc_sampler_run_packed", C.c_int, [C.POINTER(_vp), C.c_int32, C.c_int64, C.POINTER(_dp), C.POINTER(_dp)]),
    ("tamcmc_sampler_draws", C.c_int, [_vp, C.c_int64, _dp, _dp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ("tamcmc_sampler_get_state", C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _i64p]),
    ("tamcmc_sampler_get_proposal", C.c_int, [_vp, C.c_int32, _dp, _dp]),
    ("tamcmc_sampler_set_proposal", C.c_int, [_vp, C.c_int32, _dp, _dp, C.c_double]),
    ("tamcmc_sampler_set_state", C.c_int, [_vp, _dp, C.c_int64]),
    ("tamcmc_outputs_write_restore", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_char_p), _dp, _dp, _dp, _dp]),
    ("tamcmc_outputs_read_restore", C.c_int, [C.c_char_p, _ip, _ip, _i64p, _dp, _dp, _dp, _dp]),
    ("tamcmc_sampler_write_restore", C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_char_p)]),
    ("tamcmc_sampler_read_restore", C.c_int, [_vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]),
    ("tamcmc_outputs_write_params", C.c_int, [C.c_char_p, _dp, C.c_int64, C.c_int32, C.c_int32, C.c_int64, _ip, _ip, C.c_int32, C.c_int64, _dp,
                                             C.POINTER(C.c_char_p), C.c_int32]),
    ("tamcmc_outputs_write_stat_criteria", C.c_int, [C.c_char_p, _dp, C.c_int64, C.c_int32, C.c_int32]),
    ("tamcmc_outputs_read_params", C.c_int, [C.c_char_p, C.c_int32, _dp, C.c_int64, _i64p, _ip, _ip]),
    ("tamcmc_params_summary", C.c_int, [_dp, C.c_int64, C.c_int32, C.c_int64, _dp, _dp, _dp]),
    ("tamcmc_evidence_calc", C.c_int, [_dp, C.c_int32, _dp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _dp, _dp, _dp, _dp, C.POINTER(C.c_double)]),
    ("tamcmc_outputs_write_evidence", C.c_int, [C.c_char_p, C.c_int64, C.c_int32, _dp, _dp, C.c_int32, C.c_double, C.c_int32]),
    ("tamcmc_log_prior", C.c_double, [C.c_int, _dp, C.c_int64, _ip, _dp, _ip, _dp, C.c_int32, _ip]),
]


def _rebind():
    L = lib()
    for name, res, args in EXTRA_ABI:
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


class Sampler:
    """Parallel-tempered adaptive MH / Langevin sampler of one star on one GPU context.

    Defaults follow Config/default/config_default.cfg (!MALA section)."""

    def __init__(self, ctx: HipContext, star, nchains=5, lambda_temp=3.5, use_drift=0, seed=20240229, p=1.0,
                 target_acceptance=0.234, c0=10.0, epsilon1=1e-12, epsilon2=1e-12, A1=1e14, delta=0.0, delta_x=1e-10,
                 Nt_learn=(1000, 1500, 100000), periods_learn=(1, 1), dN_mixing=1, init_errors=None, fd_step_rel=1e-7,
                 engine="host", chain_groups=0, swap_rule=0):
        self._L = _rebind()
        self.ctx = ctx
        self.nchains = int(nchains)
        keep = self._keep = {}
        keep["inputs"] = _f64(star.params)
        keep["relax"] = _i32(star.relax)
        keep["plength"] = _i32(star.plength)
        keep["priors"] = _f64(star.priors)
        keep["sw"] = _i32(star.priors_switch)
        keep["extra"] = _f64(star.extra_priors)
        keep["Nt"] = np.ascontiguousarray(Nt_learn, dtype=np.int64)
        keep["per"] = np.ascontiguousarray(periods_learn, dtype=np.int64)
        nv = int((keep["relax"] == 1).sum())
        if init_errors is None:
            init_errors = default_errors(star)
        keep["err"] = _f64(init_errors)
        assert keep["err"].size == nv
        cfg = SamplerConfig()
        cfg.model_id, cfg.prior_class, cfg.likelihood_id, cfg.use_drift = int(star.model_id), int(star.prior_class), 0, int(use_drift)
        cfg.likelihood_params, cfg.Nparams = float(p), keep["inputs"].size
        cfg.inputs, cfg.relax, cfg.plength = _p(keep["inputs"]), _p(keep["relax"], _ip), _p(keep["plength"], _ip)
        cfg.priors, cfg.priors_switch = _p(keep["priors"]), _p(keep["sw"], _ip)
        cfg.extra_priors, cfg.n_extra = _p(keep["extra"]), keep["extra"].size
        cfg.Nchains, cfg.lambda_temp, cfg.target_acceptance, cfg.c0 = self.nchains, lambda_temp, target_acceptance, c0
        cfg.epsilon1, cfg.epsilon2, cfg.A1, cfg.delta, cfg.delta_x = epsilon1, epsilon2, A1, delta, delta_x
        cfg.Nt_learn, cfg.periods_learn, cfg.n_Nt_learn = _p(keep["Nt"], _i64p), _p(keep["per"], _i64p), keep["Nt"].size
        cfg.dN_mixing, cfg.init_errors, cfg.seed, cfg.fd_step_rel = int(dN_mixing), _p(keep["err"]), int(seed), fd_step_rel
        cfg.engine = {"host": 0, "device": 1}[engine]
        cfg.chain_groups = int(chain_groups)
        cfg.swap_rule = int(swap_rule)
        h = _vp()
        st = self._L.tamcmc_sampler_create(C.byref(h), ctx._h, C.byref(cfg))
        if st != OK:
            raise TamcmcError(st, "tamcmc_sampler_create: " + self._L.tamcmc_hip_last_error(ctx._h).decode())
        self._h = h
        ctx._samplers.add(self)  # the context closes its samplers before itself
        self.nvars = int(self._L.tamcmc_sampler_nvars(h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.tamcmc_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, n_iter, record=True, stats=False, out=None):
        """Advances the chains by n_iter iterations; returns (samples [n x Nchains x Nvars] or None, stats [n x Nchains x 3] or None).
        out = (samples, stats): caller-owned C-contiguous float64 arrays of those shapes to fill instead of new ones (a run that
        writes buffer after buffer, like the reference's Nbuffer ring, reuses them; pinned_empty() arrays avoid the staged copy)."""
        n_iter = int(n_iter)
        if out is not None:
            smp, stt = out
            record, stats = smp is not None, stt is not None
            for arr, shape, what in ((smp, (n_iter, self.nchains, self.nvars), "samples"), (stt, (n_iter, self.nchains, 3), "stats")):
                if arr is not None and not (isinstance(arr, np.ndarray) and arr.flags.c_contiguous and arr.dtype == np.float64 and arr.shape == shape):
                    raise ValueError(f"out: {what} must be a C-contiguous float64 array of shape {shape}")  # (the library writes that many bytes)
        else:
            smp = np.zeros((n_iter, self.nchains, self.nvars)) if record else None
            stt = np.zeros((n_iter, self.nchains, 3)) if stats else None
        st = self._L.tamcmc_sampler_run(self._h, n_iter, _p(smp), _p(stt))
        if st != OK:
            raise TamcmcError(st, self._L.tamcmc_hip_last_error(self.ctx._h).decode())
        return smp, stt

    def state(self):
        nc, nv = self.nchains, self.nvars
        out = {"vars": np.zeros((nc, nv)), "logL": np.zeros(nc), "logPrior": np.zeros(nc), "logPost": np.zeros(nc),
               "Pmove": np.zeros(nc), "sigma": np.zeros(nc)}
        cnt = np.zeros(4, dtype=np.int64)
        self._L.tamcmc_sampler_get_state(self._h, _p(out["vars"]), _p(out["logL"]), _p(out["logPrior"]), _p(out["logPost"]),
                                         _p(out["Pmove"]), _p(out["sigma"]), _p(cnt, _i64p))
        out.update(iteration=int(cnt[0]), accepted0=int(cnt[1]), swap_attempts=int(cnt[2]), swaps=int(cnt[3]))
        return out

    def info(self):
        """tamcmc_sampler_get_info as a dict (engine, sizes, which side of the size limits, iterations per scheme)."""
        v = np.zeros(9, dtype=np.int64)
        rc = self._L.tamcmc_sampler_get_info(self._h, _p(v, _i64p), 9)
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_get_info")
        keys = ("engine", "nvars", "nparams", "nchains", "adapt_in_lds", "fused_available", "chain_groups", "iter_fused", "iter_lockstep")
        return dict(zip(keys, (int(x) for x in v)))

    def gradient(self):
        """(grad, grad_prior [Nchains x Nvars], valid [Nchains]) the sampler holds for the chains' positions (use_drift = 1)."""
        g, gp = np.zeros((self.nchains, self.nvars)), np.zeros((self.nchains, self.nvars))
        v = np.zeros(self.nchains, dtype=np.int32)
        rc = self._L.tamcmc_sampler_get_gradient(self._h, _p(g), _p(gp), _p(v, _ip))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_get_gradient")
        return g, gp, v

    def last_test(self):
        """Host engine: (proposals [Nchains x Nvars], their logL / logPrior / logPosterior [Nchains x 3], log q(x'|x) and log q(x|x') [Nchains x 2],
        gradient at the proposals [Nchains x Nvars])."""
        v, st, lq = np.zeros((self.nchains, self.nvars)), np.zeros((self.nchains, 3)), np.zeros((self.nchains, 2))
        g = np.zeros((self.nchains, self.nvars))
        rc = self._L.tamcmc_sampler_get_last_test(self._h, _p(v), _p(st), _p(lq), _p(g))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_get_last_test")
        return v, st, lq, g

    def move_counts(self):
        """Per chain: iterations since creation whose record carries moved = 1 (tamcmc_sampler_get_move_counts)."""
        v = np.zeros(self.nchains, dtype=np.int64)
        rc = self._L.tamcmc_sampler_get_move_counts(self._h, _p(v, _i64p))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_get_move_counts")
        return v

    def draws(self, iteration):
        """The random numbers iteration `iteration` consumes: (z [Nchains x Nvars], u_accept [Nchains], u_swap, ind_A)."""
        z, u = np.zeros((self.nchains, self.nvars)), np.zeros(self.nchains)
        us, ia = C.c_double(0), C.c_int32(0)
        rc = self._L.tamcmc_sampler_draws(self._h, int(iteration), _p(z), _p(u), C.byref(us), C.byref(ia))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_draws")
        return z, u, us.value, ia.value

    def proposal_law(self):
        """(mu [Nchains x Nvars], covarmat [Nchains x Nvars x Nvars], sigma [Nchains]) of every chain."""
        mu, cov = np.zeros((self.nchains, self.nvars)), np.zeros((self.nchains, self.nvars, self.nvars))
        for m in range(self.nchains):
            mu[m], cov[m] = self.get_proposal(m)
        return mu, cov, self.state()["sigma"]

    def set_state(self, vars, iteration=-1):
        v = _f64(vars)
        rc = self._L.tamcmc_sampler_set_state(self._h, _p(v), int(iteration))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_set_state")

    def write_restore(self, root, names=None):
        """Checkpoint: <root>1.dat / 2.dat / 3.dat in the reference's restore-file layout."""
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        rc = self._L.tamcmc_sampler_write_restore(self._h, str(root).encode(), arr)
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_write_restore")

    def read_restore(self, root, variables=True, proposal=True, last_index=True):
        rc = self._L.tamcmc_sampler_read_restore(self._h, str(root).encode(), int(variables), int(proposal), int(last_index))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_sampler_read_restore")

    def get_proposal(self, m):
        mu, cov = np.zeros(self.nvars), np.zeros((self.nvars, self.nvars))
        self._L.tamcmc_sampler_get_proposal(self._h, int(m), _p(mu), _p(cov))
        return mu, cov

    def set_proposal(self, m, mu=None, cov=None, sigma=0.0):
        mu = _f64(mu) if mu is not None else None
        cov = _f64(cov) if cov is not None else None
        self._L.tamcmc_sampler_set_proposal(self._h, int(m), _p(mu), _p(cov), float(sigma))


def default_errors(star):
    """Initial proposal standard deviations per free parameter, in the spirit of Config/default/errors_default.cfg
    (error = fraction*value + offset per parameter family)."""
    rules = {"Height": (0.05, 0.0), "Visibility": (0.03, 0.0), "Frequency": (0.0, 0.05), "Width": (0.05, 0.0),
             "a1_0": (0.05, 0.01), "sqrt(splitting_a1)": (0.03, 0.01), "Harvey-Noise_H": (0.02, 0.0),
             "Harvey-Noise_tc": (0.02, 0.0), "White_Noise_N0": (0.01, 0.0), "Inclination": (0.0, 1.0),
             # red-giant parameters: the mixed-mode pattern is very sensitive to the period spacing
             "DP1": (0.0, 0.002), "delta01": (0.0, 0.01), "q": (0.02, 0.0), "Wfactor": (0.0, 0.02), "Hfactor": (0.0, 0.02),
             "rot_": (0.05, 0.005), "ferr_bias": (0.0, 0.005), "numax": (0.005, 0.0), "nudip": (0.005, 0.0), "alpha": (0.01, 0.0),
             "Gamma_alpha": (0.02, 0.0), "Wdip": (0.01, 0.0), "DeltaGammadip": (0.02, 0.0)}
    err = []
    for i in np.flatnonzero(star.relax == 1):
        nm = star.names[i]
        frac, off = 0.02, 1e-3
        for key, (f, o) in rules.items():
            if nm.startswith(key):
                frac, off = f, o
        err.append(abs(star.params[i]) * frac + off)
    return np.array(err)


def log_prior(star, params=None):
    """Host log-prior (Model_def::call_prior) of `star`'s model class at `params` (default: the star's own vector)."""
    L = _rebind()
    p = _f64(star.params if params is None else params)
    pl, pr, sw, ex = _i32(star.plength), _f64(star.priors), _i32(star.priors_switch), _f64(star.extra_priors)
    st = C.c_int32(0)
    v = L.tamcmc_log_prior(int(star.prior_class), _p(p), p.size, _p(pl, _ip), _p(pr), _p(sw, _ip), _p(ex), ex.size, C.byref(st))
    return float(v), int(st.value)


def write_outputs(root, star, samples, stats=None, nsamples_total=None, append=False):
    """Writes the reference's params.hdr / params_chain-<m>.bin (+ stat_criteria) files for samples [n x Nchains x Nvars]."""
    L = _rebind()
    smp = _f64(samples)
    n, nc, nv = smp.shape
    names = (C.c_char_p * len(star.names))(*[s.encode() for s in star.names])
    pl = _i32(star.plength)
    st = L.tamcmc_outputs_write_params(str(root).encode(), _p(smp), n, nc, nv, int(nsamples_total or n), _p(_i32(star.relax), _ip),
                                       _p(pl, _ip), pl.size, star.params.size, _p(_f64(star.params)), names, int(append))
    if st != OK:
        raise TamcmcError(st, "tamcmc_outputs_write_params")
    if stats is not None:
        stt = _f64(stats)
        st = L.tamcmc_outputs_write_stat_criteria(str(root).encode(), _p(stt), stt.shape[0], nc, int(append))
        if st != OK:
            raise TamcmcError(st, "tamcmc_outputs_write_stat_criteria")


def write_acceptance(file, xaxis, rates, first):
    """One line of the reference's acceptance diagnostic file (Outputs::write_txt_acceptance)."""
    L = _rebind()
    r = _f64(rates)
    rc = L.tamcmc_outputs_write_acceptance(str(file).encode(), float(xaxis), _p(r), r.size, int(bool(first)))
    if rc != OK:
        raise TamcmcError(rc, "tamcmc_outputs_write_acceptance")


def read_acceptance(file):
    """(xaxis [n], rates [n x Nchains]) of an acceptance diagnostic file."""
    L = _rebind()
    nc, n = C.c_int32(0), C.c_int64(0)
    rc = L.tamcmc_outputs_read_acceptance(str(file).encode(), C.byref(nc), 0, C.byref(n), None, None)
    if rc != OK:
        raise TamcmcError(rc, "tamcmc_outputs_read_acceptance")
    x, r = np.zeros(n.value), np.zeros((n.value, nc.value))
    L.tamcmc_outputs_read_acceptance(str(file).encode(), C.byref(nc), n.value, C.byref(n), _p(x), _p(r))
    return x, r


def read_params(root, chain):
    L = _rebind()
    n, nc, nv = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    st = L.tamcmc_outputs_read_params(str(root).encode(), int(chain), None, 0, C.byref(n), C.byref(nc), C.byref(nv))
    if st != OK:
        raise TamcmcError(st, "tamcmc_outputs_read_params")
    out = np.zeros((n.value, nv.value))
    L.tamcmc_outputs_read_params(str(root).encode(), int(chain), _p(out), n.value, C.byref(n), C.byref(nc), C.byref(nv))
    return out


def run_packed(samplers, n_iter, record=True, stats=False):
    """Several stars at once (tamcmc_sampler_run_packed): every sampler advances n_iter iterations, one host thread each inside the
    library; the samplers must sit on different contexts (which may share a GPU).  Returns ([samples_k], [stats_k])."""
    L = _rebind()
    n_iter, S = int(n_iter), len(samplers)
    smp = [np.zeros((n_iter, s.nchains, s.nvars)) if record else None for s in samplers]
    stt = [np.zeros((n_iter, s.nchains, 3)) if stats else None for s in samplers]
    hs = (_vp * S)(*[s._h for s in samplers])
    ps = (_dp * S)(*[_p(a) for a in smp])
    pt = (_dp * S)(*[_p(a) for a in stt])
    rc = L.tamcmc_sampler_run_packed(hs, S, n_iter, ps, pt)
    if rc != OK:
        raise TamcmcError(rc, "tamcmc_sampler_run_packed: " + "; ".join(L.tamcmc_hip_last_error(s.ctx._h).decode() for s in samplers))
    return smp, stt


def params_summary(samples2d):
    """mean / median / stddev per variable of [n x Nvars] samples (bin2txt's summary)."""
    L = _rebind()
    a = _f64(samples2d)
    n, nv = a.shape
    mean, med, sd = np.zeros(nv), np.zeros(nv), np.zeros(nv)
    st = L.tamcmc_params_summary(_p(a), n, nv, nv, _p(mean), _p(med), _p(sd))
    if st != OK:
        raise TamcmcError(st, "tamcmc_params_summary")
    return mean, med, sd


def evidence(Tcoefs, stats, interp_factor=1000, out_file=None, first=True):
    """Evidence diagnostic of the tempered ladder (Diagnostics::evidence_calc, diagnostics.cpp:980-1019) from the [n x Nchains x 3]
    statistics block of Sampler.run; optionally appends the reference's text line (write_evidence, :1021-1066).
    Returns (evidence, beta, L_beta, beta_interp, L_beta_interp)."""
    L = _rebind()
    T = _f64(Tcoefs)
    st3 = _f64(stats)
    n, nc = st3.shape[0], st3.shape[1]
    beta, Lb = np.zeros(nc), np.zeros(nc)
    bi, Li = np.zeros(nc * interp_factor), np.zeros(nc * interp_factor)
    ev = C.c_double(0)
    rc = L.tamcmc_evidence_calc(_p(T), nc, _p(st3), n, 3 * nc, 3, int(interp_factor), _p(beta), _p(Lb), _p(bi), _p(Li), C.byref(ev))
    if rc != OK:
        raise TamcmcError(rc, "tamcmc_evidence_calc")
    if out_file is not None:
        rc = L.tamcmc_outputs_write_evidence(str(out_file).encode(), n, nc, _p(beta), _p(Lb), int(interp_factor), ev.value, int(bool(first)))
        if rc != OK:
            raise TamcmcError(rc, "tamcmc_outputs_write_evidence")
    return ev.value, beta, Lb, bi, Li
