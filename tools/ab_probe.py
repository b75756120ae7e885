"""Acquire-phase iteration time of the device engine at a few star sizes with a given build of the library (A/B of two builds in two
processes).  python tools/ab_probe.py <path to libtamcmc_hip.so> [iterations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
pkg.LIB_PATH = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
from tamcmc_c_amd import synth

for nx, C in ((100000, 20), (10000, 10), (10000, 4), (40000, 8)):
    star = synth.make_c3_star(nx=nx, step=2000.0 / nx)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    star.set_spectrum_from_model(m0[0], seed=20240301)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    s = pkg.Sampler(ctx, star, nchains=C, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(10, 200), periods_learn=(1,), c0=2.0)
    s.run(400, record=False)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s.run(n, record=False)
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    s.close()
    ctx.close()
    print(f"{os.path.basename(sys.argv[1])}: Nx={nx} chains={C}: {best:6.2f} us per iteration", flush=True)
