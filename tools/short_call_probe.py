"""What a short run() call costs on the headline star (the driver's bench window is 20 iterations after 5 warm-up ones):
wall time of calls of n iterations, against n x the steady-state iteration time.  python tools/short_call_probe.py [n] [calls] [timing 0/1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
from tamcmc_c_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
timing = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True   # sampled launches bracketed by events (what bench.py runs with)
star = synth.make_c3_star()
ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT, timing=timing)
ctx.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
star.set_spectrum_from_model(m0[0], seed=20240301)
ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
ctx.set_spectrum(star.x, star.y)
s = pkg.Sampler(ctx, star, nchains=20, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(100, 1100), periods_learn=(1,), c0=2.0)
smp, stt = pkg.pinned_empty((n, 20, s.nvars)), pkg.pinned_empty((n, 20, 3))   # before the set-up phase: no idle gap in front of the calls
s.run(1500 - n, record=False)
s.run(n, out=(smp, stt))        # the set-up phase's last iterations are recorded into the same buffers (allocations, argument image, TLBs)
s.run(5, out=(smp[:5], stt[:5]))
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    s.run(n, out=(smp, stt))
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
t0 = time.perf_counter()
for _ in range(200):
    s.run(0, out=(smp[:0], stt[:0]))
print(f"an empty call through the Python binding: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
big = pkg.pinned_empty((3000, 20, s.nvars)), pkg.pinned_empty((3000, 20, 3))
t0 = time.perf_counter()
s.run(3000, out=big)
steady = (time.perf_counter() - t0) / 3000 * 1e6
print("first calls (us):", np.round(ts[:8], 1))
print(f"calls of {n} iterations: median {np.median(ts):.1f} us, min {ts.min():.1f} us, first {ts[0]:.1f} us; steady state {steady:.2f} us per iteration "
      f"-> {n * steady:.1f} us; overhead per call {np.median(ts) - n * steady:.1f} us ({n / np.median(ts) * 1e6:.0f} samples/s in such calls)")
