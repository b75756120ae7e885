"""Learning-phase iteration time (lockstep scheme: k_iterate with the Robbins-Monro update and the Cholesky factorisation, then
k_loglike) on the headline star.  python tools/learn_probe.py [iterations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
if os.environ.get("TAMCMC_PROBE_ADAPT"):    # probe build: adapt_chain leaves after phase N (tamcmc-c_amd: make probe)
    pkg.LIB_PATH = os.path.join(ROOT, "tamcmc-c_amd", "libtamcmc_hip_probe.so")
from tamcmc_c_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
star = synth.make_c3_star()
ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
ctx.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
star.set_spectrum_from_model(m0[0], seed=20240301)
ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
ctx.set_spectrum(star.x, star.y)
s = pkg.Sampler(ctx, star, nchains=20, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(50, 10**9), periods_learn=(1,), c0=2.0)
s.run(100, record=False)
t0 = time.perf_counter()
s.run(n, record=False)
el = time.perf_counter() - t0
print(f"learning iterations: {1e6 * el / n:.1f} us each ({n / el:.0f} per second)")
