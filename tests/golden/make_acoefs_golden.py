#!/usr/bin/env python3
"""Generates tests/golden/acoefs_py.json from the reference's own python helper
/root/reference/test/lorentzian_test/acoefs.py (imported, read-only; nothing is
copied).  Run ONLY in the authoring container (the reference does not travel to
the GPU box); the JSON it writes is the committed fixture.

Values: Pslm(s,l,m) for the (s,l) pairs the helper itself evaluates
(nunlm_from_acoefs: s<=2 for l=1, s<=4 for l=2, s<=6 for l=3) and the split
frequencies nu_nlm for seeded random (nu_c, a1..a6) draws that follow the
ranges of make_params_aj_model (test_build_l_mode.cpp:791-807).
"""
import json, os, sys, random
import matplotlib
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference/test/lorentzian_test")
import acoefs  # noqa: E402

out = {"source": "test/lorentzian_test/acoefs.py (Pslm, nunlm_from_acoefs, eval_acoefs)", "pslm": [], "nunlm": []}
smax = {1: 2, 2: 4, 3: 6}
for l in (1, 2, 3):
    for s in range(1, smax[l] + 1):
        for m in range(-l, l + 1):
            out["pslm"].append({"s": s, "l": l, "m": m, "value": float(acoefs.Pslm(s, l, m))})
rng = random.Random(20240229)
for case in range(24):
    l = 1 + case % 3
    nu = rng.uniform(1500.0, 3500.0)
    a1 = rng.uniform(0.1, 5.0)
    a = [a1, rng.uniform(-0.1, 0.1) * a1, rng.uniform(-0.025, 0.025) * a1, rng.uniform(-0.025, 0.025) * a1,
         rng.uniform(-0.01, 0.01) * a1, rng.uniform(-0.005, 0.005) * a1]
    if l < 2:
        a[2] = a[3] = 0.0
    if l < 3:
        a[4] = a[5] = 0.0
    nus = [float(v) for v in acoefs.nunlm_from_acoefs(nu, l, *a)]
    back = [float(v) for v in acoefs.eval_acoefs(l, nus)]
    out["nunlm"].append({"l": l, "nu_c": nu, "a": a, "nu_nlm": nus, "eval_acoefs": back})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "acoefs_py.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(out["pslm"]), "Pslm values,", len(out["nunlm"]), "multiplets")
