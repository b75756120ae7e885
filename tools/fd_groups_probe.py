"""Which variables cost what in a finite-difference batch: the headline star's 93 variables in groups (frequencies, heights, widths,
visibilities + inclination + a1, noise), 20 chains, event-timed batches (base launch + moments + far pass + delta launch + sums).
python tools/fd_groups_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
from tamcmc_c_amd import synth

star = synth.make_c3_star()
ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT, timing=True)
ctx.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
star.set_spectrum_from_model(m0[0], seed=20240301)
ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
ctx.set_spectrum(star.x, star.y)
names = np.array(star.names)[star.index_to_relax]
groups = {"frequencies": names == "Frequency_l", "heights": names == "Height_l0", "widths": names == "Width_l0",
          "visibilities+inclination+a1": np.isin(names, ["Visibility_l1", "Visibility_l2", "Visibility_l3", "Inclination", "a1_0"]),
          "noise": np.isin(names, ["Harvey-Noise_H", "Harvey-Noise_tc", "White_Noise_N0"]), "all": np.ones(names.size, bool)}
rng = np.random.default_rng(1)
P = np.tile(star.params, (20, 1))
P[1:, star.index_to_relax] *= 1 + 0.002 * rng.standard_normal((19, star.nvars))
T = 1.3 ** np.arange(20)
for g, sel in groups.items():
    idx = star.index_to_relax[sel]
    h = 1e-7 * np.maximum(np.abs(star.params[idx]), 1e-3)
    ctx.fd_gradient(star.model_id, P, star.plength, idx, h, T, 1.0)
    ctx.reset_kernel_stats()
    for _ in range(5):
        ctx.fd_gradient(star.model_id, P, star.plength, idx, h, T, 1.0)
    ms, nl, ne = ctx.kernel_stats()
    bins, evals = ctx.fd_stats()
    print(f"{g:30s} {idx.size:3d} variables: {1e3 * ms / nl:7.1f} us per batch, {bins / max(evals, 1):9.0f} bins walked per delta evaluation, "
          f"{ctx.fd_full_tables() // 5} full tables", flush=True)
