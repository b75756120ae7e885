#!/usr/bin/env python3
"""Writes tests/golden/logl_golden.json: log-likelihoods (and a few model bins) of the CPU oracle for fixed synthetic cases of every
model of the path, so that later rounds notice any drift of the oracle or of the device path without re-deriving the numbers.
Run from the repo root:  python tests/golden/make_logl_golden.py   (needs oracle/libtamcmc_oracle.so; no GPU)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
from tamcmc_c_amd import synth  # noqa: E402
import oracle_lib  # noqa: E402

orc = oracle_lib.Oracle()
cases = []


def add(name, model_id, params, plength, x, seed, T):
    x = float(x[0]) + float(x[1] - x[0]) * np.arange(x.size)   # exactly the grid the tests rebuild from (x0, step, nx)
    st, m0 = orc.call_model(model_id, params, plength, x)
    assert st == 0, name
    y = m0 * np.random.default_rng(seed).exponential(1.0, m0.size)
    rng = np.random.default_rng(seed + 1)
    B = len(T)
    P = np.tile(params, (B, 1))
    free = np.flatnonzero(np.abs(params) > 0)[: max(3, params.size // 4)]
    P[1:, free] *= 1 + 1e-3 * rng.standard_normal((B - 1, free.size))
    logL, models, status = orc.loglike_batch(model_id, P, plength, x, y, 1.0, np.asarray(T), want_model=True)
    assert (status == 0).all(), name
    pick = np.linspace(0, x.size - 1, 9).astype(int)
    cases.append(dict(name=name, model_id=int(model_id), plength=[int(v) for v in plength], params=P.tolist(), x0=float(x[0]),
                      step=float(x[1] - x[0]), nx=int(x.size), y_seed=int(seed), y_from_row0=True, T=[float(t) for t in T],
                      logL=[float(v) for v in logL], model_bins=[int(i) for i in pick], model_row0=[float(models[0][i]) for i in pick]))


c3 = synth.make_c3_star(nx=20000, step=0.1)
add("C3-like aj (20000 bins)", c3.model_id, c3.params, c3.plength, c3.x, 11, 1.3 ** np.arange(4))
c2 = synth.make_c2_star(nx=5000)
add("C2 local (5000 bins)", c2.model_id, c2.params, c2.plength, c2.x, 12, 1.7 ** np.arange(3))
pc, plc = synth.aj_to_classic(c3.params, c3.plength)
add("Classic a1etaa3 (20000 bins)", synth.MODEL_CLASSIC, pc, plc, c3.x, 13, [1.0, 2.0])
pr, plr = synth.make_params_rgb_model(np.random.default_rng(5), bias_type=1, model_type=0)
add("RGB v4 (3400 bins)", synth.MODEL_RGB_V4, pr, plr, 110.0 + 0.05 * np.arange(3400), 14, [1.0, 1.4, 1.96])
json.dump(dict(generator="tests/golden/make_logl_golden.py", oracle="oracle/tamcmc_oracle.c + armm_oracle.c (-O2 -ffp-contract=off)",
               cases=cases), open(os.path.join(ROOT, "tests", "golden", "logl_golden.json"), "w"))
print("wrote", len(cases), "cases")
