"""Golden log-likelihoods (tests/golden/logl_golden.json, written by tests/golden/make_logl_golden.py from the CPU oracle): the
oracle must keep reproducing them (CPU test) and the device path must match them without running the oracle (GPU test)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logl_golden.json")


def _cases():
    return json.load(open(GOLD))["cases"]


def _inputs(c, model_row0_from):
    x = c["x0"] + c["step"] * np.arange(c["nx"])
    P = np.array(c["params"])
    pl = np.array(c["plength"], dtype=np.int32)
    m0 = model_row0_from(c["model_id"], P[0], pl, x)
    y = m0 * np.random.default_rng(c["y_seed"]).exponential(1.0, m0.size)
    return x, y, P, pl, m0


@pytest.mark.parametrize("i", range(4))
def test_oracle_reproduces_its_golden_numbers(oracle, i):
    c = _cases()[i]

    def row0(mid, p, pl, x):
        st, m = oracle.call_model(mid, p, pl, x)
        assert st == 0
        return m
    x, y, P, pl, m0 = _inputs(c, row0)
    assert np.allclose(m0[c["model_bins"]], c["model_row0"], rtol=1e-13)
    logL, _, st = oracle.loglike_batch(c["model_id"], P, pl, x, y, 1.0, np.array(c["T"]))
    assert (st == 0).all() and np.allclose(logL, c["logL"], rtol=1e-12), c["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(4))
def test_device_matches_the_golden_numbers(pkg, i):
    c = _cases()[i]
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)

    def row0(mid, p, pl, x):
        ctx.set_spectrum(x, np.ones_like(x))
        _, m, st = ctx.loglike_params_batch(mid, p, pl, want_model=True)
        assert st[0] == 0
        return m[0]
    x, y, P, pl, m0 = _inputs(c, row0)
    rgb = c["model_id"] == 25          # its mixed-mode frequencies come from device tan/atan: red-giant tolerance, see test_gpu_rgb.py
    assert np.allclose(m0[c["model_bins"]], c["model_row0"], rtol=1e-9 if rgb else 1e-13), c["name"]
    # the spectrum is built from the DEVICE's own STRICT row (bit-identical to the oracle's for the main-sequence models)
    for prec, tol in ((pkg.PRECISION_STRICT, 1e-11 if rgb else 1e-12), (pkg.PRECISION_FAST, 1e-11)):
        ctx.set_option(pkg.OPT_PRECISION, prec)
        ctx.set_spectrum(x, y)
        logL, _, st = ctx.loglike_params_batch(c["model_id"], P, pl, np.array(c["T"]))
        assert (st == 0).all() and np.allclose(logL, c["logL"], rtol=tol), (c["name"], prec, np.abs(logL / np.array(c["logL"]) - 1).max())
    ctx.close()
