"""Fixed cost of one tamcmc_sampler_run call on the device engine (C3 star): many short calls against one long one."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from tamcmc_c_amd import synth
star = synth.make_c3_star()
c = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
c.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = c.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
y = star.set_spectrum_from_model(m0[0], 1)
c.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
c.set_spectrum(star.x, y)
s = pkg.Sampler(c, star, nchains=20, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(10, 20), periods_learn=(1,))
s.run(50, record=False)
for k in (1, 5, 20, 100, 2000):
    reps = max(2000 // k, 3)
    t0 = time.perf_counter()
    for _ in range(reps):
        s.run(k, record=False)
    dt = (time.perf_counter() - t0) / reps
    print(f"run({k}): {dt * 1e6:8.1f} us per call = {dt * 1e6 / k:7.1f} us per iteration; fixed part ~ {dt * 1e6 - 40.4 * k:7.1f} us", flush=True)
