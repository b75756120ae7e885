"""Host logic of the product (no GPU): the C-ABI library loads and exports every symbol of include/*.h, and the
mode-table builders reproduce the oracle's models bit-for-bit when the table is evaluated with the reference's
per-bin operation order (tests/strict_numpy.py)."""
import os
import re

import numpy as np
import pytest

from strict_numpy import eval_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()
    declared = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        src = open(os.path.join(inc, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(tamcmc_[a-z0-9_]+)\s*\(", src))
    assert len(declared) >= 12
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    bound = {n for n, _, _ in pkg.ABI + pkg.EXTRA_ABI}
    assert declared <= bound, declared - bound


def test_no_gpu_means_error_not_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.TamcmcError) as e:
        pkg.HipContext(0)
    assert e.value.code in (pkg.ERR_NO_DEVICE, pkg.ERR_HIP)


def _check_model(pkg, oracle, model_id, params, plength, x, exact=True):
    st_o, m_o = oracle.call_model(model_id, params, plength, x)
    st, mults, noise, nh = pkg.build_mode_table(model_id, params, plength, x)
    assert st == st_o == 0
    m = eval_table(mults, noise, nh, x)
    if exact:
        assert np.array_equal(m, m_o), float(np.max(np.abs(m - m_o)))
    else:
        assert np.max(np.abs(m - m_o) / m_o) < 1e-14
    return mults


@pytest.mark.parametrize("seed", range(6))
def test_aj_table_bit_exact(pkg, oracle, synth, seed):
    """The reference's own accuracy test recipe (test_build_l_mode.cpp:100-137): random aj models, lmax in {2,3},
    5 orders, 4-yr Kepler grid from 0 to Nfreqs*130+250 muHz; noise [0,1,1,0,1,1,0.1] makes the Harvey terms exact zeros."""
    rng = np.random.default_rng(100 + seed)
    lmax = int(rng.integers(2, 4))
    p, pl = synth.make_params_aj_model(rng, lmax=lmax, nfreqs=5, dnu=rng.uniform(129, 130), epsilon=rng.uniform(0, 0.05),
                                       d0l=rng.uniform(-2.6, 0))
    nx = int(np.ceil((5 * 130 + 250) / synth.KEPLER_4YR_RESOL))
    x = np.linspace(0.0, 5 * 130 + 250.0, nx)
    mults = _check_model(pkg, oracle, 23, p, pl, x)
    assert len(mults) == 5 * (lmax + 1)


def test_aj_with_eta_and_amplitudes(pkg, oracle, synth):
    rng = np.random.default_rng(7)
    p, pl = synth.make_params_aj_model(rng, lmax=3, nfreqs=6, asym=12.5, eta_switch=1.0, n_first=10)
    p[-1] = 1.0  # do_amp
    x = synth.grid(60000, 1200.0, 0.02)
    _check_model(pkg, oracle, 23, p, pl, x)


def test_classic_and_local_tables(pkg, oracle, synth):
    rng = np.random.default_rng(11)
    p, pl = synth.make_params_aj_model(rng, lmax=3, nfreqs=6, asym=-30.0, n_first=12)
    pc, plc = synth.aj_to_classic(p, pl)
    pc[pl[0] + pl[1] + pl[2:6].sum() + 2] = 0.01  # a3
    x = synth.grid(50000, 1400.0, 0.02)
    _check_model(pkg, oracle, 3, pc, plc, x)
    star = synth.make_c2_star()
    _check_model(pkg, oracle, 11, star.params, star.plength, star.x)
    s3 = synth.make_c3_star(nx=20000, step=0.1)
    _check_model(pkg, oracle, 23, s3.params, s3.plength, s3.x, exact=False)  # active Harvey pow(): numpy vs libm


def test_window_edge_cases_match_oracle(pkg, oracle, synth):
    """Modes near / beyond the grid edges, gamma and a1 on both sides of 1, tiny c."""
    rng = np.random.default_rng(3)
    x = synth.grid(5000, 100.0, 0.05)  # 100..350
    for trial in range(40):
        p, pl = synth.make_params_aj_model(rng, lmax=3, nfreqs=3, dnu=60.0, epsilon=rng.uniform(0.5, 2.5), d0l=-0.5,
                                           asym=0.0, n_first=1)
        o = pl[0] + pl[1] + pl[2:6].sum()
        p[o] = rng.choice([0.2, 1.0, 3.0])                        # a1 below / at / above 1
        p[o + 14:o + 14 + 3] = rng.choice([0.3, 1.0, 2.5], 3)     # widths below / at / above 1
        p[-2] = rng.choice([0.5, 5.0, 50.0])                      # trunc_c
        st_o, m_o = oracle.call_model(23, p, pl, x)
        st, mults, noise, nh = pkg.build_mode_table(23, p, pl, x)
        assert st == st_o
        if st == 0:
            assert np.array_equal(eval_table(mults, noise, nh, x), m_o)


def test_bad_inputs_give_status_codes(pkg, synth):
    star = synth.make_c2_star()
    st, *_ = pkg.build_mode_table(99, star.params, star.plength, star.x)
    assert st == pkg.ERR_BAD_MODEL
    p = star.params.copy()
    p[18] = float("nan")  # a width
    st, *_ = pkg.build_mode_table(11, p, star.plength, star.x)
    assert st == pkg.ERR_NAN_WINDOW
