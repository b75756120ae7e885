"""Scratch GPU probe: raw kernel timing of the C3 batch for every (precision, K); prints one line each."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_package()
from tamcmc_c_amd import synth
import oracle_lib
orc = oracle_lib.Oracle()
star = synth.make_c3_star()
_, m0 = orc.call_model(star.model_id, star.params, star.plength, star.x)
y = star.set_spectrum_from_model(m0, 1)
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = np.tile(star.params, (B, 1)); idx = star.index_to_relax
P[1:, idx] *= 1 + 0.002 * rng.standard_normal((B - 1, idx.size))
T = 1.35 ** (np.arange(B) % 20)
st, mults, _, _ = pkg.build_mode_table(star.model_id, star.params, star.plength, star.x)
W = int(((mults["i1"] - mults["i0"]) * (2 * mults["l"] + 1)).sum())
print("component-bin evals per model:", W, flush=True)
GEOMS = [(256, 4), (64, 8)] if B != 20 else [(256, 1), (256, 2), (256, 4), (64, 4), (64, 8), (64, 16)]
for prec in (pkg.PRECISION_STRICT, pkg.PRECISION_FAST_DIRECT, pkg.PRECISION_FAST):
    for wg, K in GEOMS:
        c = pkg.HipContext(0, precision=prec, timing=True, workgroup=wg, bins_per_thread=K)
        c.set_spectrum(star.x, y)
        for _ in range(3):
            c.loglike_params_batch(star.model_id, P, star.plength, T)
        c.reset_kernel_stats()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            logL, _, _ = c.loglike_params_batch(star.model_id, P, star.plength, T)
        wall = (time.perf_counter() - t0) / n
        if prec == pkg.PRECISION_STRICT:
            ref = logL.copy()
        ms, nl, ne = c.kernel_stats()
        k_us = ms / nl * 1e3
        print(f"prec={prec} wg={wg} K={K} B={B}: kernel {k_us:9.1f} us  wall/call {wall*1e6:9.1f} us  "
              f"comp-evals/s {W*B/(k_us*1e-6):.3e}  algGB/s {16*star.x.size*B/(k_us*1e-6)/1e9:.1f}  max|dlogL/logL| vs strict {np.max(np.abs(logL-ref)/np.abs(ref)):.2e}", flush=True)
        c.close()
