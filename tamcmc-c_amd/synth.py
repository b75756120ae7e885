"""Deterministic synthetic stars for the parity tests and bench.py.

The generators follow the reference's own random-input recipe
(make_params_aj_model, test/lorentzian_test/unit_tests/test_build_l_mode.cpp:769-874:
H~U(10,20), Gamma~U(0.5,2), a1~U(0.1,5), |a2|<0.1 a1, ..., inc~U(0,90), trunc_c=50, do_amp=0)
with fixed seeds.  They only produce parameter vectors / layouts / grids; the spectrum itself is
y = M(theta_true) * Exp(1) with M supplied by the caller (GPU path in bench.py, oracle in CPU tests).
"""
import numpy as np

MODEL_CLASSIC, MODEL_LOCAL, MODEL_AJ, MODEL_RGB_V4, MODEL_RGB_CTE_V4 = 3, 11, 23, 25, 27
KEPLER_4YR_RESOL = 1e6 / (4.0 * 365.0 * 86400.0)  # test_build_l_mode.cpp:107


def make_params_aj_model(rng, lmax=3, nfreqs=5, dnu=129.5, epsilon=0.02, d0l=-1.3, asym=None, noise=None,
                         eta_switch=0.0, n_first=0, dl_shift=(0, 0, 0, 0)):
    """params / plength of model_MS_Global_aj_HarveyLike (layout: SURVEY App. B; models.cpp:1207-1270)."""
    fl = []
    for el in range(lmax + 1):
        for en in range(nfreqs):
            scatter = rng.uniform(-dnu / 100, dnu / 100)
            fl.append((n_first + en + dl_shift[el] + epsilon + el / 2.0) * dnu + d0l * el * (el + 1) + scatter)
    a1 = rng.uniform(0.1, 5.0)
    aj = np.zeros(13)
    aj[0] = a1
    aj[2] = rng.uniform(-0.1 * a1, 0.1 * a1)
    aj[4] = rng.uniform(-0.025 * a1, 0.025 * a1)
    aj[6] = rng.uniform(-0.025 * a1, 0.025 * a1)
    aj[8] = rng.uniform(-0.01 * a1, 0.01 * a1)
    aj[10] = rng.uniform(-0.005 * a1, 0.005 * a1)
    aj[12] = eta_switch
    if asym is None:
        asym = rng.uniform(-100, 100) if rng.integers(0, 2) == 1 else 0.0
    vis = np.array([1.5, 0.53, 0.07][:lmax])
    height = rng.uniform(10, 20, nfreqs)
    width = rng.uniform(0.5, 2, nfreqs)
    if noise is None:
        noise = np.array([0, 1, 1, 0, 1, 1, 0.1])
    inc = rng.uniform(0.0, 90.0)
    params = np.concatenate([height, vis, np.array(fl), aj, [asym], width, noise, [inc], [50.0, 0.0]])
    nfl = [nfreqs if el <= lmax else 0 for el in range(4)]
    plength = np.array([nfreqs, lmax, nfl[0], nfl[1], nfl[2], nfl[3], 14, nfreqs, len(noise), 1, 2], dtype=np.int32)
    assert params.size == plength.sum()
    return params, plength


def grid(nx, fmin, step):
    return fmin + step * np.arange(nx, dtype=np.float64)


class Star:
    """Everything the sampler needs for one star (the content of Config.modeling.inputs + Data)."""

    def __init__(self, model_id, params, plength, x, relax, priors, priors_switch, names, prior_class, extra_priors=None):
        self.model_id = model_id
        self.params = np.asarray(params, dtype=np.float64)
        self.plength = np.asarray(plength, dtype=np.int32)
        self.x = x
        self.relax = np.asarray(relax, dtype=np.int32)
        self.priors = np.asarray(priors, dtype=np.float64)  # 4 x Nparams
        self.priors_switch = np.asarray(priors_switch, dtype=np.int32)
        self.names = names
        self.prior_class = prior_class  # 2 = io_MS_Global, 3 = io_local (Config/default/priors_ctrl.list)
        self.extra_priors = np.zeros(10) if extra_priors is None else np.asarray(extra_priors, dtype=np.float64)
        self.y = None

    @property
    def index_to_relax(self):
        return np.flatnonzero(self.relax == 1).astype(np.int32)

    @property
    def nvars(self):
        return int((self.relax == 1).sum())

    def set_spectrum_from_model(self, model, seed):
        rng = np.random.default_rng(seed)
        self.y = np.asarray(model, dtype=np.float64) * rng.exponential(1.0, size=model.size)
        return self.y


# primitive prior ids: Config/default/primepriors_ctrl.list
P_FIX, P_UNIFORM, P_GAUSS, P_JEFF = 0, 1, 2, 4


def _prior_tables(names, params, relax, rules):
    n = len(names)
    pr = np.full((4, n), -9999.0)
    sw = np.zeros(n, dtype=np.int32)
    for i, nm in enumerate(names):
        if not relax[i]:
            continue
        kind, fn = rules[nm]
        sw[i] = kind
        vals = fn(params[i])
        pr[:len(vals), i] = vals
    return pr, sw


def make_c3_star(seed=20240229, nx=100000, nmax=14, lmax=3, step=0.02, fmin=1950.0):
    """BASELINE config C3: global MS fit, model_MS_Global_aj_HarveyLike, 14 orders x l<=3 = 56 multiplets,
    111 parameters (93 free), 1e5 bins of 0.02 muHz over [1950, 3950) muHz, two active Harvey terms."""
    rng = np.random.default_rng(seed)
    noise = np.array([1.27, 49.6, 2.0, 2.66, 1.52, 2.0, 0.005])  # test/inputs/Sun/fast/..Priorevalrange.model:94-98
    params, plength = make_params_aj_model(rng, lmax=lmax, nfreqs=nmax, dnu=135.1, epsilon=0.4, d0l=-1.5, asym=0.0,
                                           noise=noise, eta_switch=0.0, n_first=15, dl_shift=(0, 0, -1, -1))
    o_inc = plength[:9].sum()
    params[o_inc] = 60.0 + rng.uniform(-15, 15)
    names = (["Height_l0"] * nmax + ["Visibility_l%d" % (l + 1) for l in range(lmax)] + ["Frequency_l"] * (4 * nmax) +
             ["a1_0", "a1_1", "a2_0", "a2_1", "a3_0", "a3_1", "a4_0", "a4_1", "a5_0", "a5_1", "a6_0", "a6_1", "eta0_switch",
              "Lorentzian_asymetry"] + ["Width_l0"] * nmax +
             ["Harvey-Noise_H", "Harvey-Noise_tc", "Harvey-Noise_p", "Harvey-Noise_H", "Harvey-Noise_tc", "Harvey-Noise_p",
              "White_Noise_N0"] + ["Inclination", "Truncation_parameter", "do_amp"])
    assert len(names) == params.size
    relax = np.zeros(params.size, dtype=np.int32)
    relax[:nmax] = 1                                   # heights
    relax[nmax:nmax + lmax] = 1                        # visibilities
    relax[nmax + lmax:nmax + lmax + 4 * nmax] = 1      # frequencies
    o_split = nmax + lmax + 4 * nmax
    relax[o_split] = 1                                 # a1_0
    o_w = o_split + 14
    relax[o_w:o_w + nmax] = 1                          # widths
    o_n = o_w + nmax
    relax[[o_n + 3, o_n + 4, o_n + 6]] = 1             # second Harvey H, tc and the white noise
    relax[o_n] = 1                                     # first Harvey H
    relax[o_inc] = 1                                   # inclination
    rules = {
        "Height_l0": (P_JEFF, lambda v: (1.0, 1.0e4)),
        "Visibility_l1": (P_GAUSS, lambda v: (1.5, 0.15)),
        "Visibility_l2": (P_GAUSS, lambda v: (0.53, 0.05)),
        "Visibility_l3": (P_GAUSS, lambda v: (0.07, 0.02)),
        "Frequency_l": (P_UNIFORM, lambda v: (v - 8.0, v + 8.0)),
        "a1_0": (P_UNIFORM, lambda v: (0.0, 8.0)),
        "Width_l0": (P_JEFF, lambda v: (0.05, 40.0)),
        "Harvey-Noise_H": (P_UNIFORM, lambda v: (0.0, 50.0)),
        "Harvey-Noise_tc": (P_UNIFORM, lambda v: (0.0, 100.0)),
        "White_Noise_N0": (P_UNIFORM, lambda v: (0.0, 5.0)),
        "Inclination": (P_UNIFORM, lambda v: (0.0, 90.0)),
    }
    pr, sw = _prior_tables(names, params, relax, rules)
    x = grid(nx, fmin, step)
    # extra_priors (io_ms_global.cpp): [smooth switch, smooth coef, |aj/a1| limits x6, impose_normHnlm, model index 9 = aj]
    extra = np.array([1.0, 2.0, 0.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.0, 9.0])
    return Star(MODEL_AJ, params, plength, x, relax, pr, sw, names, prior_class=2, extra_priors=extra)


def make_c2_star(seed=20240229, nx=10000):
    """BASELINE config C2: local slice, model_MS_local_basic, 6 multiplets (l=0,1,2 x 2 orders), 28 parameters
    (21 free), 1e4 bins at 1-yr resolution from 2900 muHz, white noise only."""
    rng = np.random.default_rng(seed)
    step = 1e6 / (365.0 * 86400.0)
    x = grid(nx, 2900.0, step)
    dnu, eps = 135.1, 0.45
    f = []
    for l, d in ((0, 0.0), (1, -2.5), (2, -9.0)):
        for n in (21, 22):
            nn = n - 1 if l == 2 else n
            f.append((nn + eps + l / 2.0) * dnu + d + rng.uniform(-0.5, 0.5))
    heights = rng.uniform(10, 20, 6)
    widths = rng.uniform(0.5, 2, 6)
    a1, inc = 1.0, 60.0
    split = [0.0, 0.0, 0.0, np.sqrt(a1) * np.cos(np.radians(inc)), np.sqrt(a1) * np.sin(np.radians(inc)), 0.0]
    params = np.concatenate([heights, f, split, widths, [0.1], [0.0], [50.0, 0.0]])
    plength = np.array([6, 0, 2, 2, 2, 0, 6, 6, 1, 1, 2], dtype=np.int32)
    assert params.size == plength.sum() == 28
    names = (["Height_l"] * 6 + ["Frequency_l"] * 6 +
             ["Splitting_a1", "Asphericity_eta", "Splitting_a3", "sqrt(splitting_a1).cosi", "sqrt(splitting_a1).sini",
              "Lorentzian_asymetry"] + ["Width_l"] * 6 + ["White_Noise_N0", "Inclination", "Truncation_parameter", "do_amp"])
    relax = np.zeros(28, dtype=np.int32)
    relax[0:12] = 1
    relax[[15, 16]] = 1
    relax[18:24] = 1
    relax[24] = 1
    rules = {
        "Height_l": (P_JEFF, lambda v: (1.0, 1.0e4)),
        "Frequency_l": (P_UNIFORM, lambda v: (v - 5.0, v + 5.0)),
        "sqrt(splitting_a1).cosi": (P_UNIFORM, lambda v: (0.0, 2.5)),
        "sqrt(splitting_a1).sini": (P_UNIFORM, lambda v: (0.0, 2.5)),
        "Width_l": (P_JEFF, lambda v: (0.05, 40.0)),
        "White_Noise_N0": (P_UNIFORM, lambda v: (0.0, 5.0)),
    }
    pr, sw = _prior_tables(names, params, relax, rules)
    extra = np.array([0.0, 0.0, 0.2, 0.0, 0, 0, 0, 0, 0, 0])  # priors_local: a3/a1 limit at [2]
    return Star(MODEL_LOCAL, params, plength, x, relax, pr, sw, names, prior_class=3, extra_priors=extra)


def aj_to_classic(params, plength):
    """Re-packs an aj-layout vector into the Classic (a1, eta, a3, -, -, asym) layout (Nsplit=6)."""
    nmax, lmax = int(plength[0]), int(plength[1])
    nf = int(plength[2:6].sum())
    o = nmax + lmax + nf
    sp = params[o:o + 14]
    split = np.array([sp[0], 0.0, sp[4], 0.0, 0.0, sp[13]])
    out = np.concatenate([params[:o], split, params[o + 14:]])
    pl = plength.copy()
    pl[6] = 6
    return out, pl


def make_params_rgb_model(rng, nmax=6, dnu=20.0, epsilon=0.2, n_first=6, delta0l=-0.6, DPl=80.0, alpha_g=0.0, q=0.15, nferr=4,
                          bias_type=0, model_type=0, rot_env=0.1, rot_core=0.6, inclination=55.0, trunc_c=20.0, ferr_scale=0.05, cte_width=False):
    """Parameter vector of model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (layout: SURVEY App. B; generator in the spirit of
    make_params_RGB_model, test/lorentzian_test/unit_tests/test_build_l_mode.cpp:584-767): nmax radial orders, the l=1 block
    [delta0l, DPl, alpha_g, q, -, -, Wfactor, Hfactor, fref x nferr, ferr x nferr], l=2/l=3 lists of nmax-1 modes, ten
    rotation/asymmetry slots, the six-parameter Appourchaux width law, two Harvey profiles + white noise.
    cte_width: the layout of model_RGB_asympt_aj_CteWidth_HarveyLike_v4 (id 27) instead: a single width parameter."""
    n = np.arange(nmax)
    fl0 = (n_first + n + epsilon) * dnu + rng.uniform(-0.01, 0.01, nmax) * dnu
    numax = fl0.mean()
    heights = 40.0 * np.exp(-0.5 * ((fl0 - numax) / (1.2 * dnu)) ** 2) + 2.0
    vis = np.array([1.5, 0.53, 0.08])
    fref = (n_first + 1 + np.arange(nferr) * max((nmax - 2) / max(nferr - 1, 1), 1) + epsilon + 0.5) * dnu
    ferr = rng.uniform(-ferr_scale, ferr_scale, nferr) if bias_type != 0 else np.zeros(nferr)
    l1 = np.concatenate([[delta0l, DPl, alpha_g, q, 0.0, 0.0, 0.9, 0.9], fref, ferr])   # Wfactor, Hfactor < 1: zeta = 1 keeps a finite mode
    fl2 = fl0[1:] - 0.12 * dnu
    fl3 = fl0[:-1] + 0.21 * dnu
    split = np.array([rot_env, rot_core, 0.0, 0.0, 0.01, 0.0, 0.0, 0.0, 1.0, 0.0])
    width = np.array([numax, numax, 1.5, 0.15, 0.8 * numax, 2.5])          # nu_max, nu_dip, alpha, Gamma_alpha, W_dip, DeltaGamma_dip
    if cte_width:
        width = np.array([0.14])
    noise = np.array([30.0, 40.0, 2.0, 10.0, 8.0, 2.0, 0.4])
    cfg = np.array([trunc_c, 0.0, 0.0, float(model_type), float(bias_type), float(nferr)])
    params = np.concatenate([heights, vis, fl0, l1, fl2, fl3, split, width, noise, [inclination], cfg])
    plength = np.array([nmax, 3, nmax, l1.size, fl2.size, fl3.size, 10, width.size, 7, 1, 6], dtype=np.int32)
    assert params.size == plength.sum()
    return params, plength


def make_c5_star(seed=20240229, nx=200000, nmax=10, dnu=10.0, bias_type=1, model_type=0, nferr=6, margin=15.0, cte_width=False):
    """BASELINE config C5 family: red giant, model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (id 25), prior class io_asymptotic (4).
    The l=1 mixed modes are not parameters: they follow from (delta0l, DPl, alpha_g, q) through the ARMM solver."""
    rng = np.random.default_rng(seed)
    params, plength = make_params_rgb_model(rng, nmax=nmax, dnu=dnu, n_first=6, DPl=80.0, q=0.15, nferr=nferr, bias_type=bias_type,
                                            model_type=model_type, cte_width=cte_width)
    o = np.cumsum([0] + list(plength))
    names = (["Height_l0"] * nmax + ["Visibility_l1", "Visibility_l2", "Visibility_l3"] + ["Frequency_l"] * nmax +
             ["delta01", "DP1", "alpha_g", "q", "sigma_H_l1", "sigma_g_l1", "Wfactor", "Hfactor"] + ["fref_bias"] * nferr + ["ferr_bias"] * nferr +
             ["Frequency_l"] * (plength[4] + plength[5]) +
             ["rot_env", "rot_core", "a2_env", "a2_core", "a3_env", "a4_env", "a5_env", "a6_env", "eta0_switch", "Lorentzian_asymetry"] +
             (["Width_l0"] if cte_width else ["numax", "nudip", "alpha", "Gamma_alpha", "Wdip", "DeltaGammadip"]) +
             ["Harvey-Noise_H", "Harvey-Noise_tc", "Harvey-Noise_p", "Harvey-Noise_H", "Harvey-Noise_tc", "Harvey-Noise_p", "White_Noise_N0"] +
             ["Inclination", "Truncation_parameter", "do_amp", "sigma_limit", "model_type", "bias_type", "Nferr"])
    assert len(names) == params.size
    relax = np.zeros(params.size, dtype=np.int32)
    relax[:nmax] = 1                                        # heights
    relax[o[2]:o[3]] = 1                                    # l=0 frequencies
    relax[[o[3], o[3] + 1, o[3] + 3, o[3] + 6, o[3] + 7]] = 1   # delta01, DP1, q, Wfactor, Hfactor
    if bias_type != 0:
        relax[o[3] + 8 + nferr:o[4]] = 1                    # bias values at the spline nodes
    relax[o[4]:o[6]] = 1                                    # l=2, l=3 frequencies
    relax[[o[6], o[6] + 1]] = 1                             # envelope and core rotation
    relax[o[7]:o[8]] = 1                                    # width law
    relax[[o[8], o[8] + 3, o[8] + 6]] = 1                   # Harvey heights, white noise
    relax[o[9]] = 1                                         # inclination
    rules = {
        "Height_l0": (P_JEFF, lambda v: (0.1, 1.0e4)),
        "Width_l0": (P_JEFF, lambda v: (0.01, 5.0)),
        "Frequency_l": (P_UNIFORM, lambda v: (v - 0.3 * dnu / 10.0 * 3, v + 0.3 * dnu / 10.0 * 3)),
        "delta01": (P_UNIFORM, lambda v: (v - 1.0, v + 1.0)),
        "DP1": (P_UNIFORM, lambda v: (v - 1.0, v + 1.0)),
        "q": (P_UNIFORM, lambda v: (0.0, 1.0)),
        "Wfactor": (P_UNIFORM, lambda v: (0.0, 1.0)), "Hfactor": (P_UNIFORM, lambda v: (0.0, 1.0)),
        "ferr_bias": (P_GAUSS, lambda v: (0.0, 0.1)),
        "rot_env": (P_UNIFORM, lambda v: (0.0, 1.0)), "rot_core": (P_UNIFORM, lambda v: (0.0, 3.0)),
        "numax": (P_GAUSS, lambda v: (v, 0.05 * v)), "nudip": (P_GAUSS, lambda v: (v, 0.05 * v)), "alpha": (P_GAUSS, lambda v: (v, 0.3)),
        "Gamma_alpha": (P_GAUSS, lambda v: (v, 0.05)), "Wdip": (P_GAUSS, lambda v: (v, 0.1 * v)), "DeltaGammadip": (P_GAUSS, lambda v: (v, 0.3)),
        "Harvey-Noise_H": (P_UNIFORM, lambda v: (0.0, 500.0)), "White_Noise_N0": (P_UNIFORM, lambda v: (0.0, 50.0)),
        "Inclination": (P_UNIFORM, lambda v: (0.0, 90.0)),
    }
    pr, sw = _prior_tables(names, params, relax, rules)
    fl0 = params[o[2]:o[3]]
    lo, hi = fl0.min() - margin, fl0.max() + margin
    x = lo + (hi - lo) / nx * np.arange(nx)
    # extra_priors (io_asymptotic.cpp:422-431): [smooth switch, smooth coef, |a3/a1| limit, impose_normHnlm, model switch 3 = v4 models]
    extra = np.array([1.0, 2.0, 0.2, 0.0, 3.0, 0, 0, 0, 0, 0])
    return Star(MODEL_RGB_CTE_V4 if cte_width else MODEL_RGB_V4, params, plength, x, relax, pr, sw, names, prior_class=4, extra_priors=extra)
