"""Timing probe of the red-giant path at BASELINE-C5 scale: 2e5 bins, 40 parameter vectors per batch (one per chain)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_package()
from tamcmc_c_amd import synth
import oracle_lib
rng = np.random.default_rng(1)
nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 10
params, pl = synth.make_params_rgb_model(rng, nmax=nmax, dnu=10.0, n_first=6, DPl=80.0, q=0.15, nferr=6, bias_type=1, trunc_c=20.0)
Nx = 200000
fl0 = params[pl[0] + pl[1]:pl[0] + pl[1] + pl[2]]
lo, hi = fl0.min() - 15.0, fl0.max() + 15.0
x = lo + (hi - lo) / Nx * np.arange(Nx)
print("Nparams", params.size, "range", lo, hi, "step", x[1] - x[0], flush=True)
B = 40
P = np.tile(params, (B, 1))
o = np.cumsum([0] + list(pl))
P[1:, o[3] + 1] *= 1 + 0.001 * rng.standard_normal(B - 1)
T = 1.2 ** np.arange(B)
c = pkg.HipContext(0, precision=pkg.PRECISION_FAST, timing=True, bins_per_thread=int(os.environ["RGB_K"]) if os.environ.get("RGB_K") else None)
c.set_spectrum(x, np.ones(Nx))
_, m, st = c.loglike_params_batch(pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4, P[:1], pl, T[:1], want_model=True)
assert st[0] == 0
y = m[0] * np.random.default_rng(2).exponential(1.0, Nx)
c.set_spectrum(x, y)
for _ in range(2):
    logL, _, st = c.loglike_params_batch(pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4, P, pl, T)
assert (st == 0).all()
c.reset_kernel_stats()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    logL, _, st = c.loglike_params_batch(pkg.MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4, P, pl, T)
wall = (time.perf_counter() - t0) / n
ms, nl, ne = c.kernel_stats()
print(f"B={B} Nx={Nx}: wall/call {wall*1e3:.2f} ms (pre-step + tables + likelihood), k_loglike {ms/nl*1e3:.1f} us per launch", flush=True)
if len(sys.argv) > 2:
    orc = oracle_lib.Oracle(fast=True)
    t0 = time.perf_counter()
    ref, _, so = orc.loglike_batch(synth.MODEL_RGB_V4, P[:4], pl, x, y, 1.0, T[:4])
    t1 = time.perf_counter() - t0
    print(f"oracle (CPU, OpenMP): {t1/4*1e3:.1f} ms per evaluation; max |dlogL/logL| {np.abs(logL[:4]/ref-1).max():.2e}")
# host-driven sampler on the same star: 40 tempered chains (BASELINE config C5 asks for 40; the reference caps at 24)
star = synth.make_c5_star(nx=Nx, nmax=nmax, dnu=10.0, bias_type=1, nferr=6)
c.set_spectrum(star.x, np.ones(Nx))
_, m, st = c.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
star.set_spectrum_from_model(m[0], 7)
c.set_spectrum(star.x, star.y)
s = pkg.Sampler(c, star, nchains=40, lambda_temp=1.15, seed=5, engine="host", Nt_learn=(10, 150), periods_learn=(1,))
s.run(150, record=False)   # adaptation phase
t0 = time.perf_counter(); n = 200
s.run(n, record=False)
dt = time.perf_counter() - t0
stt = s.state()
print(f"C5-like MH, 40 chains, host-driven engine: {n/dt:.1f} samples/s ({dt/n*1e3:.2f} ms per iteration), chain-0 acceptance over the run {stt['accepted0']/(150+n):.2f}", flush=True)
