"""Which kind of workgroup sets the length of a fused step launch: the headline star's acquire-phase iteration time with kinds of
workgroups left out of the launch (PROBE build only, TAMCMC_PROBE_STEP bit mask: 1 candidate roles, 2 commit workgroups, 4 L z blocks,
8 tiles take their slot from memory (no decision), 16 no tiles).  Chains of masked runs are wrong by construction; only the time is read.
python tools/step_probe.py [masks, comma separated]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
pkg.LIB_PATH = os.path.join(ROOT, "tamcmc-c_amd", "libtamcmc_hip_probe.so")
assert os.path.exists(pkg.LIB_PATH), "build it first: make -C tamcmc-c_amd probe"
from tamcmc_c_amd import synth

masks = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,16,16:256,16:512,16:1024,16:768,16:1280,16:1536,24,0:256,0:512,0:1024").split(",")]
star = synth.make_c3_star()
ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
ctx.set_spectrum(star.x, np.ones_like(star.x))
_, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
star.set_spectrum_from_model(m0[0], seed=20240301)
ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
ctx.set_spectrum(star.x, star.y)
for groups in (0,):
    for mask in masks:
        os.environ["TAMCMC_PROBE_STEP"] = mask.split(":")[0]
        os.environ["TAMCMC_PROBE_ADAPT"] = mask.split(":")[1] if ":" in mask else "0"  # 256 / 512 / 1024: no prior / rows / background roles
        s = pkg.Sampler(ctx, star, nchains=20, lambda_temp=1.3, seed=7, engine="device", Nt_learn=(10, 200), periods_learn=(1,), c0=2.0, chain_groups=groups)
        s.run(400, record=False)
        t0 = time.perf_counter()
        s.run(3000, record=False)
        us = (time.perf_counter() - t0) / 3000 * 1e6
        s.close()
        print(f"chain_groups={groups} mask {mask:>8s}: {us:6.2f} us per iteration", flush=True)
