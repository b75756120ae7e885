#!/usr/bin/env python3
"""pack_probe.py -- several independent stars on ONE GPU at once (one context + one device-resident sampler per star, one host
thread each): aggregate MCMC samples/s against the number of co-resident stars.  A single C3 star is latency-bound (two short
dependent kernels per iteration); co-resident stars fill the idle SIMDs.  Usage: python tools/pack_probe.py [S ...]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def make_star(pkg, synth, k, nx, chains, lam, warm):
    star = synth.make_c3_star(seed=20240229 + k, nx=nx, step=2000.0 / nx)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    y = star.set_spectrum_from_model(m0[0], seed=20240301 + k)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, y)
    s = pkg.Sampler(ctx, star, nchains=chains, lambda_temp=lam, seed=7 + k, engine="device", chain_groups=int(os.environ.get("GROUPS", "1")), Nt_learn=(max(warm // 2, 1), max(warm, 2)),
                    periods_learn=(1,))
    s.run(warm, record=False)
    return ctx, s


def main():
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    pkg = entry.load_package()
    from tamcmc_c_amd import synth
    counts = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]
    steps, warm, nx, chains, lam = int(os.environ.get("STEPS", "2000")), 200, 100000, 20, 1.3
    pool = [make_star(pkg, synth, k, nx, chains, lam, warm) for k in range(max(counts))]
    if os.environ.get("PACK_NATIVE"):
        from tamcmc_c_amd import sampler as smod
        for S in counts:
            ss = [pool[k][1] for k in range(S)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            smod.run_packed(ss, steps, record=False)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(f"S={S} (library threads): {S * steps / el:9.0f} samples/s aggregate", flush=True)
        return
    for S in counts:
        bar = threading.Barrier(S + 1)
        per = [0.0] * S

        def work(k):
            bar.wait()
            t0 = time.perf_counter()
            pool[k][1].run(steps, record=False)
            per[k] = time.perf_counter() - t0

        th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        for t in th:
            t.start()
        torch.cuda.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"S={S}: {S * steps / el:9.0f} samples/s aggregate ({1e6 * el / steps:6.1f} us per iteration of all stars; slowest star "
              f"{max(per) * 1e6 / steps:6.1f} us/iteration, fastest {min(per) * 1e6 / steps:6.1f}); "
              f"algorithmic {S * steps / el * chains * 16.0 * nx / 1e9:7.0f} GB/s", flush=True)
    for ctx, s in pool:
        s.close()
        ctx.close()


if __name__ == "__main__":
    main()
