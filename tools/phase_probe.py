"""Where k_loglike's time goes at throughput (large B): the same C3 batch with kernel phases skipped (TAMCMC_PROBE_SKIP bit mask:
1 near-field loop, 2 far-field coefficients + their reduction, 4 tile polynomial (Horner), 8 reciprocal/log of the epilogue,
16 no multiplet staged at all).  Results of masked runs are wrong by construction; only the kernel time is read."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
# the phase-skip mask exists in the PROBE build only (`make -C tamcmc-c_amd probe`); the product library has no such code path
pkg.LIB_PATH = os.path.join(ROOT, "tamcmc-c_amd", "libtamcmc_hip_probe.so")
assert os.path.exists(pkg.LIB_PATH), "build it first: make -C tamcmc-c_amd probe"
from tamcmc_c_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 400
star = synth.make_c3_star()
c = pkg.HipContext(0, precision=pkg.PRECISION_STRICT, timing=True)
c.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = c.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
y = star.set_spectrum_from_model(m0[0], 1)
c.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
c.set_spectrum(star.x, y)
rng = np.random.default_rng(0)
P = np.tile(star.params, (B, 1)); idx = star.index_to_relax
P[1:, idx] *= 1 + 0.002 * rng.standard_normal((B - 1, idx.size))
T = 1.3 ** (np.arange(B) % 20)
base = None
masks = [int(v) for v in os.environ.get("MASKS", "0,1,2,4,8,16,3,15,31").split(",")]
for mask in masks:
    os.environ["TAMCMC_PROBE_SKIP"] = str(mask)
    c.loglike_params_batch(star.model_id, P, star.plength, T)
    c.reset_kernel_stats()
    for _ in range(int(os.environ.get("REPS", "5"))):
        c.loglike_params_batch(star.model_id, P, star.plength, T)
    ms, nl, ne = c.kernel_stats()
    us = 1e3 * ms / ne
    base = base or us
    print(f"mask {mask:2d}: {us:6.3f} us per evaluation ({100 * us / base:5.1f} %)", flush=True)
