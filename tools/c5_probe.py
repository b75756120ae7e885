"""C5 (red giant, 2e5 bins, 40 chains) on the device-resident engine, alone -- the command tools/profile_bench.sh traces for the
pre-step kernels' share of an iteration.  python tools/c5_probe.py [steps] [engine] [chain_groups] [bins_per_thread]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
from tamcmc_c_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
engine = sys.argv[2] if len(sys.argv) > 2 else "device"
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kbins = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rs = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
rc = pkg.HipContext(0, precision=pkg.PRECISION_FAST, timing=True)
rc.set_spectrum(rs.x, np.ones_like(rs.x))
_, mr, _ = rc.loglike_params_batch(rs.model_id, rs.params, rs.plength, want_model=True)
rs.set_spectrum_from_model(mr[0], 7)
if kbins:
    rc.set_option(pkg.OPT_BINS_PER_THREAD, kbins)
rc.set_spectrum(rs.x, rs.y)
s = pkg.Sampler(rc, rs, nchains=40, lambda_temp=1.15, seed=5, engine=engine, Nt_learn=(10, 200), periods_learn=(1,), chain_groups=groups)
s.run(250, record=False)
s.run(20, record=True)
rc.reset_kernel_stats()
t0 = time.perf_counter()
smp, _ = s.run(steps, record=True)
el = time.perf_counter() - t0
ms, nl, ne = rc.kernel_stats()
acc = np.mean(np.any(smp[1:] != smp[:-1], axis=2), axis=0)
print(f"engine={engine} groups={groups} K={kbins} steps={steps} samples/s={steps / el:.1f} ms/iteration={1e3 * el / steps:.4f} k_loglike us/launch={1e3 * ms / max(nl, 1):.1f} "
      f"accept chain0={acc[0]:.3f} mean={acc.mean():.3f}")
