import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from tamcmc_c_amd import synth
star = synth.make_c3_star(seed=20240229, nx=100000, step=0.02)
for timing in (True, False):
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT, timing=timing)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    y = star.set_spectrum_from_model(m0[0], seed=20240301)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, y)
    s = pkg.Sampler(ctx, star, nchains=20, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(100, 200), periods_learn=(1,), dN_mixing=1)
    s.run(300, record=False)
    for n in (1, 5, 20, 100, 1000):
        for rec in (False, True):
            ts = []
            for rep in range(8):
                t0 = time.perf_counter(); s.run(n, record=rec, stats=rec); ts.append(time.perf_counter() - t0)
            print(f"timing={timing} n={n:5d} record={rec}: best {min(ts)*1e6:8.1f} us  median {sorted(ts)[4]*1e6:8.1f} us  -> per-iter {min(ts)/n*1e6:7.2f} us", flush=True)
    s.close(); ctx.close()
