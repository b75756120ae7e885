"""tamcmc-c_amd: MI355X-native hot path of the TAMCMC parallel-tempered sampler.

Python here is plumbing only: a ctypes binding of the C ABI declared in
include/tamcmc_hip.h (built in-tree into libtamcmc_hip.so by `make`), used by the
tests, bench.py and __graft_entry__.  The compute path is the HIP library; there is
no CPU fallback -- a missing library or a missing GPU raises.

The directory name contains a hyphen, so import it through `load_package()` of
__graft_entry__.py (it registers this package as module `tamcmc_c_amd`).
"""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtamcmc_hip.so")

OK = 0
ERR_HIP, ERR_EMPTY_WINDOW, ERR_NAN_WINDOW, ERR_BAD_MODEL, ERR_BAD_ARG, ERR_NO_SPECTRUM, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7
MODEL_MS_GLOBAL_A1ETAA3_CLASSIC, MODEL_MS_LOCAL_BASIC, MODEL_MS_GLOBAL_AJ = 3, 11, 23
MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4 = 27   # constant-width variant of 25, same path
MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4 = 25  # batched device path only (tamcmc_hip_loglike_params_batch): needs the ARMM pre-step
PRECISION_STRICT, PRECISION_FAST, PRECISION_FAST_DIRECT = 0, 1, 2
OPT_PRECISION, OPT_TIMING, OPT_BINS_PER_THREAD, OPT_WORKGROUP, OPT_FD_WINDOWED, OPT_STEP_SCHEME, OPT_ARMM_DENSE_SCAN = 1, 2, 3, 4, 5, 6, 7


class Multiplet(C.Structure):
    """struct tamcmc_multiplet (include/tamcmc_hip.h), 152 bytes."""
    _fields_ = [("l", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("flags", C.c_int32),
                ("fc", C.c_double), ("gamma", C.c_double), ("asym", C.c_double),
                ("nu", C.c_double * 7), ("hv", C.c_double * 7)]


MULT_DTYPE = np.dtype([("l", "<i4"), ("i0", "<i4"), ("i1", "<i4"), ("flags", "<i4"), ("fc", "<f8"), ("gamma", "<f8"),
                       ("asym", "<f8"), ("nu", "<f8", (7,)), ("hv", "<f8", (7,))])
assert MULT_DTYPE.itemsize == 152 and C.sizeof(Multiplet) == 152

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# every symbol include/tamcmc_hip.h declares: (name, restype, argtypes)
ABI = [
    ("tamcmc_hip_version", C.c_char_p, []),
    ("tamcmc_hip_create", C.c_int, [C.POINTER(_vp), C.c_int]),
    ("tamcmc_hip_destroy", None, [_vp]),
    ("tamcmc_hip_last_error", C.c_char_p, [_vp]),
    ("tamcmc_hip_set_option", C.c_int, [_vp, C.c_int, C.c_int64]),
    ("tamcmc_hip_host_alloc", C.c_void_p, [C.c_size_t]),
    ("tamcmc_hip_host_free", None, [_vp]),
    ("tamcmc_hip_set_spectrum", C.c_int, [_vp, _dp, _dp, C.c_int64]),
    ("tamcmc_hip_loglike_batch", C.c_int, [_vp, C.c_int, _vp, _ip, _dp, C.c_int, _ip, _ip, _dp, C.c_double, _dp, _dp]),
    ("tamcmc_build_mode_table", C.c_int, [C.c_int, _dp, _ip, _dp, C.c_int64, _vp, C.c_int, C.POINTER(C.c_int), _dp,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("tamcmc_hip_loglike_params_batch", C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int64, _ip, _dp, C.c_double, _dp, _dp, _ip]),
    ("tamcmc_hip_fd_gradient", C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int64, _ip, _ip, C.c_int, _dp, _dp, C.c_double, _dp, _dp]),
    ("tamcmc_hip_fd_gradient_posterior", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, C.c_int64, _ip, _ip, C.c_int, _dp, _dp, C.c_double,
                                                  _dp, _ip, _dp, _dp, _dp, _dp, _dp]),
    ("tamcmc_hip_rgb_mixed_modes", C.c_int, [_vp, C.c_int, _dp, C.c_int64, _ip, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    ("tamcmc_hip_get_kernel_stats", C.c_int, [_vp, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("tamcmc_hip_reset_kernel_stats", C.c_int, [_vp]),
    ("tamcmc_hip_get_fd_stats", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("tamcmc_hip_get_fd_full_tables", C.c_int, [_vp, C.POINTER(C.c_int64)]),
]

_lib = None


def lib():
    """Loads libtamcmc_hip.so (never builds it silently, never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C {_HERE}` (or __graft_entry__.build()) first; "
                               "there is no CPU fallback for the product path")
        _lib = C.CDLL(LIB_PATH)
        for name, res, args in ABI + EXTRA_ABI:
            f = getattr(_lib, name)
            f.restype = res
            f.argtypes = args
    return _lib


EXTRA_ABI = []  # filled by sampler.py (include/tamcmc_sampler.h)


class TamcmcError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"tamcmc_hip error {code}: {msg}")
        self.code = code


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t=_dp):
    return a.ctypes.data_as(t) if a is not None else None


def pinned_empty(shape, dtype=np.float64):
    """A numpy array in page-locked host memory (tamcmc_hip_host_alloc): the record buffers of Sampler.run(out=...) are then filled by
    an asynchronous DMA.  Freed when the array (and every view of it) is garbage-collected."""
    L = lib()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = L.tamcmc_hip_host_alloc(max(n, 1))
    if not p:
        raise MemoryError("tamcmc_hip_host_alloc failed")
    buf = (C.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, L.tamcmc_hip_host_free, p)
    return arr


def build_mode_table(model_id, params, plength, x):
    """Host-side table builder (no GPU needed). Returns (status, mults[structured], noise_abs, nharvey)."""
    L = lib()
    params, plength, x = _f64(params), _i32(plength), _f64(x)
    cap = 4096
    mults = np.zeros(cap, dtype=MULT_DTYPE)
    noise = np.zeros(max(int(plength[8]), 1))
    n, nh, nn = C.c_int(0), C.c_int(0), C.c_int(0)
    st = L.tamcmc_build_mode_table(int(model_id), _p(params), _p(plength, _ip), _p(x), x.size, mults.ctypes.data, cap,
                                   C.byref(n), _p(noise), C.byref(nh), C.byref(nn))
    return st, mults[:min(n.value, cap)].copy(), noise[:max(nn.value, 0)].copy(), nh.value


class HipContext:
    """One context = one GPU + one HIP stream + the resident spectrum (tamcmc_hip_ctx)."""

    def __init__(self, device=0, precision=PRECISION_STRICT, timing=False, bins_per_thread=None, workgroup=None):
        self._L = lib()
        h = _vp()
        st = self._L.tamcmc_hip_create(C.byref(h), int(device))
        if st != OK:
            raise TamcmcError(st, "tamcmc_hip_create failed (no GPU / HIP runtime?) -- there is no CPU fallback")
        self._h = h
        self._samplers = weakref.WeakSet()  # samplers borrow the context: they are destroyed first, whatever order the caller (or the GC) uses
        self.Nx = 0
        self.set_option(OPT_PRECISION, precision)
        self.set_option(OPT_TIMING, 1 if timing else 0)
        if workgroup:
            self.set_option(OPT_WORKGROUP, workgroup)
        if bins_per_thread:
            self.set_option(OPT_BINS_PER_THREAD, bins_per_thread)

    def close(self):
        if getattr(self, "_h", None):
            for smp in list(getattr(self, "_samplers", ())):
                smp.close()
            self._L.tamcmc_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, tolerate=()):
        if st != OK and st not in tolerate:
            raise TamcmcError(st, self._L.tamcmc_hip_last_error(self._h).decode())
        return st

    def set_option(self, opt, value):
        self._chk(self._L.tamcmc_hip_set_option(self._h, int(opt), int(value)))

    def set_spectrum(self, x, y):
        x, y = _f64(x), _f64(y)
        assert x.shape == y.shape and x.ndim == 1
        self._chk(self._L.tamcmc_hip_set_spectrum(self._h, _p(x), _p(y), x.size))
        self.Nx = x.size

    def loglike_batch(self, mults, offsets, noise, nharvey, nnoise, Tcoefs=None, p=1.0, want_model=False):
        mults = np.ascontiguousarray(mults, dtype=MULT_DTYPE)
        offsets, nharvey, nnoise = _i32(offsets), _i32(nharvey), _i32(nnoise)
        noise = _f64(noise)
        B = nharvey.size
        noise = noise.reshape(B, -1)
        T = _f64(Tcoefs) if Tcoefs is not None else None
        logL = np.zeros(B)
        model = np.zeros((B, self.Nx)) if want_model else None
        st = self._L.tamcmc_hip_loglike_batch(self._h, B, mults.ctypes.data, _p(offsets, _ip), _p(noise), noise.shape[1],
                                             _p(nharvey, _ip), _p(nnoise, _ip), _p(T), float(p), _p(logL), _p(model))
        self._chk(st)
        return logL, model

    def loglike_params_batch(self, model_id, params, plength, Tcoefs=None, p=1.0, want_model=False):
        params = _f64(params)
        if params.ndim == 1:
            params = params[None, :]
        B, Np = params.shape
        plength = _i32(plength)
        T = _f64(Tcoefs) if Tcoefs is not None else None
        logL = np.zeros(B)
        model = np.zeros((B, self.Nx)) if want_model else None
        status = np.zeros(B, dtype=np.int32)
        st = self._L.tamcmc_hip_loglike_params_batch(self._h, int(model_id), B, _p(params), Np, _p(plength, _ip), _p(T),
                                                    float(p), _p(logL), _p(model), _p(status, _ip))
        # a per-vector table failure comes back as the call's code AND in status[b] (that vector's logL is NaN): not an exception
        self._chk(st, tolerate=(ERR_EMPTY_WINDOW, ERR_NAN_WINDOW) + ((int(st),) if (status != OK).any() else ()))
        return logL, model, status

    def fd_gradient(self, model_id, params, plength, index_to_relax, hstep, Tcoefs=None, p=1.0):
        params = _f64(params)
        if params.ndim == 1:
            params = params[None, :]
        Cn, Np = params.shape
        plength, idx, h = _i32(plength), _i32(index_to_relax), _f64(hstep)
        T = _f64(Tcoefs) if Tcoefs is not None else None
        l0 = np.zeros(Cn)
        g = np.zeros((Cn, idx.size))
        st = self._L.tamcmc_hip_fd_gradient(self._h, int(model_id), Cn, _p(params), Np, _p(plength, _ip), _p(idx, _ip),
                                           idx.size, _p(h), _p(T), float(p), _p(l0), _p(g))
        self._chk(st, tolerate=(ERR_EMPTY_WINDOW, ERR_NAN_WINDOW))
        return l0, g

    def fd_gradient_posterior(self, star, params, hstep, Tcoefs=None, p=1.0):
        """Gradient of the tempered log-posterior of `star`'s model at each row of params (device-built FD batch)."""
        params = _f64(params)
        if params.ndim == 1:
            params = params[None, :]
        Cn, Np = params.shape
        plength, idx, h = _i32(star.plength), _i32(star.index_to_relax), _f64(hstep)
        pri, sw, ex = _f64(star.priors), _i32(star.priors_switch), _f64(star.extra_priors)
        T = _f64(Tcoefs) if Tcoefs is not None else None
        l0, pr0, g, gp = np.zeros(Cn), np.zeros(Cn), np.zeros((Cn, idx.size)), np.zeros((Cn, idx.size))
        st = self._L.tamcmc_hip_fd_gradient_posterior(self._h, int(star.model_id), int(star.prior_class), Cn, _p(params), Np,
                                                     _p(plength, _ip), _p(idx, _ip), idx.size, _p(h), _p(T), float(p), _p(pri),
                                                     _p(sw, _ip), _p(ex), _p(l0), _p(pr0), _p(g), _p(gp))
        self._chk(st, tolerate=(ERR_EMPTY_WINDOW, ERR_NAN_WINDOW))
        self.last_grad_prior = gp
        return l0, pr0, g

    def rgb_mixed_modes(self, model_id, params, plength, max_modes=1024):
        """l=1 mixed modes of one red-giant vector from the device pre-step: (nu_m, zeta, H1/H0) -- what ARMM's do_solve prints."""
        params, plength = _f64(params), _i32(plength)
        nu, z, h = np.zeros(max_modes), np.zeros(max_modes), np.zeros(max_modes)
        n = C.c_int(0)
        self._chk(self._L.tamcmc_hip_rgb_mixed_modes(self._h, int(model_id), _p(params), params.size, _p(plength, _ip), max_modes,
                                                     _p(nu), _p(z), _p(h), C.byref(n)))
        k = min(n.value, max_modes)
        return nu[:k].copy(), z[:k].copy(), h[:k].copy()

    def kernel_stats(self):
        ms, n, e = C.c_double(0), C.c_int64(0), C.c_int64(0)
        self._chk(self._L.tamcmc_hip_get_kernel_stats(self._h, C.byref(ms), C.byref(n), C.byref(e)))
        return ms.value, n.value, e.value

    def fd_stats(self):
        """(bins inside the affected ranges, delta evaluations) of the windowed finite-difference launches since the last reset."""
        b, e = C.c_int64(0), C.c_int64(0)
        self._chk(self._L.tamcmc_hip_get_fd_stats(self._h, C.byref(b), C.byref(e)))
        return b.value, e.value

    def fd_full_tables(self):
        """Delta evaluations since the last reset that were evaluated as whole perturbed tables (timing on)."""
        n = C.c_int64(0)
        self._chk(self._L.tamcmc_hip_get_fd_full_tables(self._h, C.byref(n)))
        return n.value

    def reset_kernel_stats(self):
        self._chk(self._L.tamcmc_hip_reset_kernel_stats(self._h))


from . import sampler  # noqa: E402,F401  (registers the tamcmc_sampler_* symbols in EXTRA_ABI)
from . import inputs  # noqa: E402,F401  (include/tamcmc_io.h)
from .sampler import Sampler  # noqa: E402,F401
