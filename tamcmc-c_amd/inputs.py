"""Input front end (include/tamcmc_io.h): `.data` and local-fit `.model` files -> the `Star` the sampler takes.
ctypes binding of csrc/host_io.cpp; mirrors Config::read_inputs_priors_local + the data range cut of Config::setup
(tamcmc/sources/config.cpp:709-723, :312-347 of the reference)."""
import ctypes as C

import numpy as np

from . import EXTRA_ABI, TamcmcError, lib, _dp, _ip, _vp, _p
from .synth import Star

_i64p = C.POINTER(C.c_int64)

EXTRA_ABI += [
    ("tamcmc_io_last_error", C.c_char_p, []),
    ("tamcmc_io_read_data", C.c_int, [C.c_char_p, C.POINTER(_dp), _i64p, _i64p]),
    ("tamcmc_io_free", None, [_vp]),
    ("tamcmc_io_select_range", C.c_int, [_dp, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double, _i64p, _i64p]),
    ("tamcmc_io_load_model_local", C.c_int, [C.c_char_p, C.c_int, C.c_double, C.POINTER(_vp)]),
    ("tamcmc_io_load_model_global", C.c_int, [C.c_char_p, C.c_double, C.POINTER(_vp)]),
    ("tamcmc_io_load_model_asymptotic", C.c_int, [C.c_char_p, C.c_double, C.POINTER(_vp)]),
    ("tamcmc_inputs_free", None, [_vp]),
    ("tamcmc_cfg_last_error", C.c_char_p, []),
    ("tamcmc_cfg_open", C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    ("tamcmc_cfg_free", None, [_vp]),
    ("tamcmc_cfg_string", C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    ("tamcmc_cfg_numbers", C.c_int, [_vp, C.c_char_p, C.c_char_p, _dp, C.c_int, C.POINTER(C.c_int)]),
    ("tamcmc_cfg_sampler", C.c_int, [_vp, _vp, _i64p, _i64p, C.c_int, _i64p, _i64p]),
    ("tamcmc_io_init_errors", C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), _dp, C.c_int64, _dp]),
    ("tamcmc_inputs_nparams", C.c_int, [_vp]),
    ("tamcmc_inputs_get", C.c_int, [_vp, _dp, _ip, _dp, _ip, _ip, _dp, _dp, _ip, _ip, _dp, _dp]),
    ("tamcmc_inputs_name", C.c_char_p, [_vp, C.c_int]),
    ("tamcmc_inputs_prior_name", C.c_char_p, [_vp, C.c_int]),
    ("tamcmc_inputs_model_name", C.c_char_p, [_vp]),
]


def _L():
    L = lib()
    for name, res, args in EXTRA_ABI:
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


def _check(rc, what):
    if rc != 0:
        raise TamcmcError(rc, "%s: %s" % (what, _L().tamcmc_io_last_error().decode()))


def read_data(path):
    """Whole `.data` table as an [nrows, ncols] array (config.cpp:907-1060)."""
    L = _L()
    tab = _dp()
    nr, nc = C.c_int64(0), C.c_int64(0)
    _check(L.tamcmc_io_read_data(str(path).encode(), C.byref(tab), C.byref(nr), C.byref(nc)), "read_data")
    try:
        return np.ctypeslib.as_array(tab, shape=(nr.value, nc.value)).copy()
    finally:
        L.tamcmc_io_free(tab)


def select_range(table, xmin, xmax, x_col=0):
    """Row range [imin, imax) the reference keeps for a fit over [xmin, xmax) (config.cpp:312-347)."""
    L = _L()
    t = np.ascontiguousarray(table, dtype=np.float64)
    a, b = C.c_int64(0), C.c_int64(0)
    _check(L.tamcmc_io_select_range(_p(t), t.shape[0], t.shape[1], int(x_col), float(xmin), float(xmax), C.byref(a), C.byref(b)),
           "select_range")
    return a.value, b.value


class ModelInputs:
    """Input_Data built from a `.model` file: params, relax, priors (4 x N), prior ids, plength, extra priors, names."""

    def __init__(self, handle):
        L = _L()
        h = handle
        try:
            n = L.tamcmc_inputs_nparams(h)
            self.params = np.zeros(n)
            self.relax = np.zeros(n, dtype=np.int32)
            self.priors = np.zeros((4, n))
            self.priors_switch = np.zeros(n, dtype=np.int32)
            self.plength = np.zeros(11, dtype=np.int32)
            self.extra_priors = np.zeros(10)
            rng = np.zeros(2)
            mid, pc = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
            dnu, cl = np.zeros(1), np.zeros(1)
            _check(L.tamcmc_inputs_get(h, _p(self.params), _p(self.relax, _ip), _p(self.priors), _p(self.priors_switch, _ip),
                                       _p(self.plength, _ip), _p(self.extra_priors), _p(rng), _p(mid, _ip), _p(pc, _ip), _p(dnu),
                                       _p(cl)), "inputs_get")
            self.freq_range = (float(rng[0]), float(rng[1]))
            self.model_id, self.prior_class = int(mid[0]), int(pc[0])
            self.dnu, self.c_l = float(dnu[0]), float(cl[0])
            self.names = [L.tamcmc_inputs_name(h, i).decode() for i in range(n)]
            self.prior_names = [L.tamcmc_inputs_prior_name(h, i).decode() for i in range(n)]
            self.model_name = L.tamcmc_inputs_model_name(h).decode()
        finally:
            L.tamcmc_inputs_free(h)


class LocalInputs(ModelInputs):
    """Local fit (model_MS_local_basic), slice `slice_ind` of the `.model` file."""

    def __init__(self, model_path, slice_ind, resol):
        h = _vp()
        _check(_L().tamcmc_io_load_model_local(str(model_path).encode(), int(slice_ind), float(resol), C.byref(h)), "load_model_local")
        super().__init__(h)


class GlobalInputs(ModelInputs):
    """Global main-sequence fit (model_MS_Global_aj_HarveyLike, model_MS_Global_a1etaa3_HarveyLike_Classic)."""

    def __init__(self, model_path, resol):
        h = _vp()
        _check(_L().tamcmc_io_load_model_global(str(model_path).encode(), float(resol), C.byref(h)), "load_model_global")
        super().__init__(h)


class AsymptoticInputs(ModelInputs):
    """Red-giant fit (model_RGB_asympt_aj_AppWidth_HarveyLike_v4, model_RGB_asympt_aj_CteWidth_HarveyLike_v4; io_asymptotic.cpp)."""

    def __init__(self, model_path, resol):
        h = _vp()
        _check(_L().tamcmc_io_load_model_asymptotic(str(model_path).encode(), float(resol), C.byref(h)), "load_model_asymptotic")
        super().__init__(h)


def star_from_inputs(inp, x, y=None):
    star = Star(inp.model_id, inp.params, inp.plength, x, inp.relax, inp.priors, inp.priors_switch, inp.names, inp.prior_class,
                inp.extra_priors)
    star.y = y
    return star


def load_global_star(model_path, data_path, x_col=0, y_col=1):
    """`.model` + `.data` of a global fit -> (Star with x, y cut to the file's range, GlobalInputs)."""
    tab = read_data(data_path)
    inp = GlobalInputs(model_path, tab[2, x_col] - tab[1, x_col])  # config.cpp:682
    a, b = select_range(tab, inp.freq_range[0], inp.freq_range[1], x_col)
    return star_from_inputs(inp, np.ascontiguousarray(tab[a:b, x_col]), np.ascontiguousarray(tab[a:b, y_col])), inp


def load_asymptotic_star(model_path, data_path, x_col=0, y_col=1):
    """`.model` + `.data` of a red-giant fit -> (Star with x, y cut to the file's range, AsymptoticInputs)."""
    tab = read_data(data_path)
    inp = AsymptoticInputs(model_path, tab[2, x_col] - tab[1, x_col])
    a, b = select_range(tab, inp.freq_range[0], inp.freq_range[1], x_col)
    return star_from_inputs(inp, np.ascontiguousarray(tab[a:b, x_col]), np.ascontiguousarray(tab[a:b, y_col])), inp


def load_local_star(model_path, data_path, slice_ind=0, x_col=0, y_col=1):
    """`.model` + `.data` of a local fit -> (Star with x, y cut to the slice's range, LocalInputs)."""
    tab = read_data(data_path)
    resol = tab[2, x_col] - tab[1, x_col]  # config.cpp:720
    inp = LocalInputs(model_path, slice_ind, resol)
    a, b = select_range(tab, inp.freq_range[0], inp.freq_range[1], x_col)
    x = np.ascontiguousarray(tab[a:b, x_col])
    star = Star(inp.model_id, inp.params, inp.plength, x, inp.relax, inp.priors, inp.priors_switch, inp.names, inp.prior_class,
                inp.extra_priors)
    star.y = np.ascontiguousarray(tab[a:b, y_col])
    return star, inp


class Cfg:
    """A `.cfg` file of the reference's dialect (`!Group:` / `key=value; comment`)."""

    def __init__(self, path):
        self._L = _L()
        self._h = _vp()
        rc = self._L.tamcmc_cfg_open(str(path).encode(), C.byref(self._h))
        if rc != 0:
            raise TamcmcError(rc, "cfg_open: " + self._L.tamcmc_cfg_last_error().decode())

    def close(self):
        if self._h:
            self._L.tamcmc_cfg_free(self._h)
            self._h = None

    def string(self, group, key):
        buf = C.create_string_buffer(512)
        rc = self._L.tamcmc_cfg_string(self._h, group.encode(), key.encode(), buf, 512)
        if rc != 0:
            raise KeyError("%s/%s: %s" % (group, key, self._L.tamcmc_cfg_last_error().decode()))
        return buf.value.decode()

    def numbers(self, group, key, max_n=16):
        out = np.zeros(max_n)
        n = C.c_int(0)
        rc = self._L.tamcmc_cfg_numbers(self._h, group.encode(), key.encode(), _p(out), max_n, C.byref(n))
        if rc != 0:
            raise KeyError("%s/%s: %s" % (group, key, self._L.tamcmc_cfg_last_error().decode()))
        return out[:n.value].copy()

    def sampler_kwargs(self):
        """The !MALA / !Modeling / !Outputs settings as keyword arguments of `Sampler` (+ Nsamples, Nbuffer, prior_class)."""
        from .sampler import SamplerConfig
        sc = SamplerConfig()
        nt = np.zeros(16, dtype=np.int64)
        pe = np.zeros(16, dtype=np.int64)
        ns, nb = C.c_int64(0), C.c_int64(0)
        rc = self._L.tamcmc_cfg_sampler(self._h, C.addressof(sc), _p(nt, _i64p), _p(pe, _i64p), 16, C.byref(ns), C.byref(nb))
        if rc != 0:
            raise TamcmcError(rc, "cfg_sampler: " + self._L.tamcmc_cfg_last_error().decode())
        n = sc.n_Nt_learn
        return dict(nchains=sc.Nchains, lambda_temp=sc.lambda_temp, use_drift=sc.use_drift, p=sc.likelihood_params,
                    target_acceptance=sc.target_acceptance, c0=sc.c0, epsilon1=sc.epsilon1, epsilon2=sc.epsilon2, A1=sc.A1,
                    delta=sc.delta, delta_x=sc.delta_x, dN_mixing=int(sc.dN_mixing), Nt_learn=tuple(int(v) for v in nt[:n]),
                    periods_learn=tuple(int(v) for v in pe[:n - 1])), dict(Nsamples=ns.value, Nbuffer=nb.value,
                                                                            prior_class=sc.prior_class, likelihood_id=sc.likelihood_id)


def init_errors(errors_path, star):
    """Initial proposal standard deviations of the star's free parameters from an errors_default.cfg-style file."""
    L = _L()
    idx = star.index_to_relax
    names = (C.c_char_p * len(idx))(*[star.names[i].encode() for i in idx])
    vals = np.ascontiguousarray(star.params[idx], dtype=np.float64)
    out = np.zeros(len(idx))
    _check(L.tamcmc_io_init_errors(str(errors_path).encode(), names, _p(vals), len(idx), _p(out)), "init_errors")
    return out
