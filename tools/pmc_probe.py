"""Minimal GPU program for rocprofv3 --pmc passes: N launches of a C3 batch through the C ABI.  B = evaluations per launch
(default 10: the device sampler launches the 20 chains as two chain groups of 10)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from tamcmc_c_amd import synth
prec = pkg.PRECISION_FAST if (len(sys.argv) < 2 or sys.argv[1] == "fast") else pkg.PRECISION_STRICT
K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
star = synth.make_c3_star()
c = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
c.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = c.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
y = star.set_spectrum_from_model(m0[0], 1)
c.set_option(pkg.OPT_PRECISION, prec)
if K:
    c.set_option(pkg.OPT_BINS_PER_THREAD, K)
c.set_spectrum(star.x, y)
rng = np.random.default_rng(0)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 10
P = np.tile(star.params, (B, 1)); idx = star.index_to_relax
P[1:, idx] *= 1 + 0.002 * rng.standard_normal((B - 1, idx.size))
T = 1.3 ** np.arange(B)
for _ in range(n):
    logL, _, _ = c.loglike_params_batch(star.model_id, P, star.plength, T)
print("done", logL[:3])
