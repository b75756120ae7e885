"""Acquire-phase iteration time with one launch per iteration (chain_groups=1) against the default two chain groups, for a few star sizes.
python tools/groups_probe.py [bins_per_lane]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
from tamcmc_c_amd import synth

kbins = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # bins per lane of the likelihood tile (0 = library default)
for nx, chains in ((100000, 20), (100000, 8), (10000, 10), (10000, 20), (400000, 20)):
    star = synth.make_c3_star(nx=nx, step=2000.0 / nx)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    star.set_spectrum_from_model(m0[0], seed=20240301)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    if kbins:
        ctx.set_option(pkg.OPT_BINS_PER_THREAD, kbins)
    ctx.set_spectrum(star.x, star.y)
    out = []
    for groups in (1, 0):
        s = pkg.Sampler(ctx, star, nchains=chains, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(10, 200), periods_learn=(1,), c0=2.0, chain_groups=groups)
        s.run(400, record=False)
        t0 = time.perf_counter()
        s.run(3000, record=False)
        out.append((time.perf_counter() - t0) / 3000 * 1e6)
        s.close()
    print(f"Nx={nx} chains={chains}: one launch per iteration {out[0]:.2f} us, two chain groups {out[1]:.2f} us")
    ctx.close()
