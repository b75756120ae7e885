#!/usr/bin/env python3
"""bench.py -- MCMC samples/s of the MI355X hot path on BASELINE.json's headline configuration.

One "step" = one MCMC iteration of one star = ALL tempered chains advanced once
(propose -> batched Lorentzian-sum model + chi^2(2 d.o.f.) log-likelihood on the GPU -> accept -> PT swap),
i.e. the reference's loop counter i (MALA.cpp:623,743).  Workload at N=1: configs[2] (C3) of BASELINE.json:
global MS fit, model_MS_Global_aj_HarveyLike, 1e5 bins x 111 parameters (93 free) x 20 tempered chains, synthetic star.
N>1 (torchrun, one rank per GPU): one independent star per GPU, no data-path collective -> weak scaling;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

The spectrum is resident in HBM before the timed region; per-step host->device traffic is the chains' mode tables
(~170 KB) -- the boundary hands over host parameter vectors, like Model_def::generate_model does.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the 'packed' leg runs several stars' streams side by side: give every live stream its own hardware queue (ROCclr default: 4; two
# streams that share a queue serialise).  Must be set before the HIP runtime starts; the one-star headline does not depend on it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # vector fp64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chains", type=int, default=20)
    ap.add_argument("--nx", type=int, default=100000)
    ap.add_argument("--precision", choices=["fast", "strict"], default="fast")
    ap.add_argument("--sampler", choices=["mh", "mala"], default="mh",
                    help="mh = adaptive random-walk MH + PT (what the reference runs); mala = Langevin drift, FD gradient")
    ap.add_argument("--engine", choices=["device", "host"], default="device",
                    help="device = whole MCMC iteration resident on the GPU; host = host-driven loop (one device call per step)")
    ap.add_argument("--bins-per-thread", type=int, default=0, choices=[0, 1, 2, 4, 8, 16], help="tile = workgroup*K bins (0 = library default)")
    ap.add_argument("--workgroup", type=int, default=0, choices=[0, 64, 256], help="workgroup size of the likelihood kernel (0 = library default)")
    ap.add_argument("--mala-steps", type=int, default=30, help="extra MALA-FD measurement (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--packed-stars", type=int, default=4,
                    help="extra leg at N=1: this many independent C3 stars co-resident on the GPU (0 = skip); reported under 'packed', "
                         "never as 'value' (the headline stays one star per GPU, BASELINE configs[2])")
    ap.add_argument("--rgb-steps", type=int, default=150,
                    help="extra leg at N=1: BASELINE configs[4] family (red giant, model id 25, 2e5 bins, 40 chains; host-driven engine with the "
                         "mixed-mode solver on the device), this many timed iterations (0 = skip); reported under 'c5_rgb'")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline run (no packed / red-giant / MALA / launch-shape / CPU legs): the command profiled under "
                         "rocprofv3 for profiles/, so that the per-kernel averages are those of the headline launches")
    ap.add_argument("--dn-mixing", type=int, default=1, help="parallel-tempering swap attempt every N iterations (reference default 1, config_default.cfg:28)")
    a = ap.parse_args()
    a.steps = max(a.steps, 1)
    a.warmup = max(a.warmup, 0)
    if a.headline_only:
        a.mala_steps, a.packed_stars, a.rgb_steps, a.no_cpu_baseline = 0, 0, 0, True
    return a


def cpu_baseline(star, y, nchains, lam, budget_s):
    """The CPU restatement (oracle, -O3 build, OpenMP over chains like MALA.cpp:648) timed on the same workload:
    repeated batches of `nchains` model+logL evaluations = the hot-path share of one MCMC iteration."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    cores = min(nchains, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    orc = oracle_lib.Oracle(fast=True)
    rng = np.random.default_rng(1)
    P = np.tile(star.params, (nchains, 1))
    idx = star.index_to_relax
    P[1:, idx] *= 1.0 + 0.002 * rng.standard_normal((nchains - 1, idx.size))
    T = lam ** np.arange(nchains)
    orc.loglike_batch(star.model_id, P, star.plength, star.x, y, 1.0, T)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        orc.loglike_batch(star.model_id, P, star.plength, star.x, y, 1.0, T)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": n / el, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} batches of {nchains} chain evaluations (model_MS_Global_aj_HarveyLike + chi22p, Nx={star.x.size}) "
                      f"in {el:.1f} s; oracle/tamcmc_oracle.c -O3 -march=x86-64-v3, OpenMP over chains as MALA.cpp:648; "
                      "hot path only (no proposal/Cholesky/output cost), so it flatters the CPU"}


def pmc_traffic():
    """HBM bytes per k_loglike launch from the committed PMC passes (None if no profile has been committed)."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        d = json.load(open(p))
        return d["hbm_bytes_per_launch"], d.get("launch", "")
    except Exception:
        return None, ""


def main():
    t_proc = time.perf_counter()
    marks = []  # (label, seconds since the previous mark): where the process's wall time goes besides the timed region

    def mark(label, _t=[t_proc]):
        now = time.perf_counter()
        marks.append((label, round(now - _t[0], 3)))
        _t[0] = now

    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # one rank per GPU; TAMCMC_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks
    backend = os.environ.get("TAMCMC_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    pkg = entry.load_package()
    from tamcmc_c_amd import synth

    lam = 1.3
    star = synth.make_c3_star(seed=20240229 + rank, nx=a.nx, step=2000.0 / a.nx)
    prec = pkg.PRECISION_FAST if a.precision == "fast" else pkg.PRECISION_STRICT
    ctx = pkg.HipContext(device_index, precision=prec, timing=True, bins_per_thread=a.bins_per_thread or None, workgroup=a.workgroup or None)
    # synthetic spectrum y = M(theta_true) * Exp(1): the model row comes from the GPU path itself (STRICT arithmetic)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    y = star.set_spectrum_from_model(m0[0], seed=20240301 + rank)
    ctx.set_option(pkg.OPT_PRECISION, prec)
    ctx.set_spectrum(star.x, y)

    def make_sampler(use_drift, learn_until):
        eng = "host" if use_drift else a.engine   # the Langevin drift runs on the host-driven engine
        return pkg.Sampler(ctx, star, nchains=a.chains, lambda_temp=lam, use_drift=use_drift, seed=7 + rank, engine=eng,
                           Nt_learn=((learn_until // 2, learn_until) if learn_until >= 2 else (10**9, 10**9 + 1)),  # adaptation ends inside the warm-up
                           periods_learn=(1,), dN_mixing=a.dn_mixing)

    from tamcmc_c_amd import shard
    mark("imports, library load, synthetic star, spectrum upload")
    smp = make_sampler(1 if a.sampler == "mala" else 0, a.warmup)
    smp.run(a.warmup, record=False)
    mark("sampler set-up + warm-up steps")
    ctx.reset_kernel_stats()
    acc0 = smp.state()
    # barrier + synchronize on both sides, MAX over ranks (tests/test_multirank_gloo.py covers this on gloo)
    elapsed, _ = shard.timed_region(lambda: smp.run(a.steps, record=False), dist=dist, sync=torch.cuda.synchronize)
    k_ms, k_launches, k_evals = ctx.kernel_stats()
    st = smp.state()
    value = shard.aggregate_rate(a.steps, world, elapsed)
    mark("TIMED REGION (the K steps behind `value`), fences included")

    extra = {}
    if a.mala_steps > 0 and a.sampler == "mh" and world == 1:
        ms = make_sampler(1, 10)
        ms.run(5, record=False)
        ctx.reset_kernel_stats()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ms.run(a.mala_steps, record=False)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        mk_ms, mk_l, mk_e = ctx.kernel_stats()
        extra["mala_fd"] = {"samples_per_s": a.mala_steps / e1, "steps": a.mala_steps,
                            "evals_per_step": mk_e / a.mala_steps, "kernel_us_per_launch": mk_ms / max(mk_l, 1) * 1e3,
                            "alg_GBps": 16.0 * a.nx * mk_e / max(mk_ms * 1e-3, 1e-12) / 1e9}
        ms.close()

    if extra:
        mark("extra leg: mala_fd")
    if a.packed_stars > 1 and a.sampler == "mh" and a.engine == "device" and world == 1:
        # Several independent stars on ONE GPU (one context + one device-resident sampler + one host thread per star,
        # tamcmc_sampler_run_packed): a single star's iteration is two short dependent kernels, co-resident stars fill the idle SIMDs.
        from tamcmc_c_amd import sampler as smod
        pool = []
        for k in range(a.packed_stars):
            sk = synth.make_c3_star(seed=20240229 + 100 + k, nx=a.nx, step=2000.0 / a.nx)
            ck = pkg.HipContext(device_index, precision=pkg.PRECISION_STRICT)
            ck.set_spectrum(sk.x, np.ones_like(sk.x))
            _, mk, _ = ck.loglike_params_batch(sk.model_id, sk.params, sk.plength, want_model=True)
            yk = sk.set_spectrum_from_model(mk[0], seed=20240301 + 100 + k)
            ck.set_option(pkg.OPT_PRECISION, prec)
            ck.set_spectrum(sk.x, yk)
            pool.append((ck, pkg.Sampler(ck, sk, nchains=a.chains, lambda_temp=lam, seed=107 + k, engine="device", chain_groups=1,
                                         Nt_learn=((a.warmup // 2, a.warmup) if a.warmup >= 2 else (10**9, 10**9 + 1)), periods_learn=(1,), dN_mixing=a.dn_mixing)))
        ps = [q[1] for q in pool]
        smod.run_packed(ps, a.warmup, record=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        smod.run_packed(ps, a.steps, record=False)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        rate = a.packed_stars * a.steps / e1
        extra["packed"] = {"stars_per_gpu": a.packed_stars, "samples_per_s": rate, "us_per_star_iteration": 1e6 * e1 / (a.steps * a.packed_stars),
                           "alg_GBps": rate * a.chains * 16.0 * a.nx / 1e9, "frac_of_hbm_peak": rate * a.chains * 16.0 * a.nx / 1e9 / HBM_PEAK_GBS,
                           "note": "aggregate over the co-resident stars (each a full 20-chain C3 fit, one stream group per star); "
                                   "every star's samples are bit-identical to its solo run (tests/test_gpu_sampler.py)"}
        for ck, sk_ in pool:
            sk_.close()
            ck.close()

    if "packed" in extra:
        mark("extra leg: packed (set-up of the co-resident stars + their warm-up + timed steps)")
    if a.rgb_steps > 0 and a.sampler == "mh" and world == 1:
        # BASELINE configs[4] family: red-giant star, mixed modes solved per proposal (csrc/rgb_prestep.hip), 40 tempered chains
        rs = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
        rc = pkg.HipContext(device_index, precision=prec, timing=True)
        rc.set_spectrum(rs.x, np.ones_like(rs.x))
        _, mr, _ = rc.loglike_params_batch(rs.model_id, rs.params, rs.plength, want_model=True)
        rs.set_spectrum_from_model(mr[0], 7)
        rc.set_spectrum(rs.x, rs.y)
        rsmp = pkg.Sampler(rc, rs, nchains=40, lambda_temp=1.15, seed=5, engine="host", Nt_learn=(10, 100), periods_learn=(1,))
        rsmp.run(100, record=False)
        rc.reset_kernel_stats()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rsmp.run(a.rgb_steps, record=False)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        r_ms, r_l, r_e = rc.kernel_stats()
        extra["c5_rgb"] = {"samples_per_s": a.rgb_steps / e1, "ms_per_step": 1e3 * e1 / a.rgb_steps, "steps": a.rgb_steps,
                           "workload": f"model_RGB_asympt_aj_AppWidth_HarveyLike_v4, Nx=200000, {rs.params.size} params ({rs.nvars} free), "
                                       "40 tempered chains, ~150 mixed modes per chain from the device ARMM solver; host-driven engine",
                           "k_loglike_us_per_launch": r_ms / max(r_l, 1) * 1e3,
                           "k_loglike_alg_GBps": 16.0 * 200000 * r_e / max(r_ms * 1e-3, 1e-12) / 1e9}
        rsmp.close()
        rc.close()

    if "c5_rgb" in extra:
        mark("extra leg: c5_rgb")
    shapes = []
    if world == 1 and a.sampler == "mh" and not a.headline_only:
        # the same kernel at other launch sizes (standalone batched calls through the C ABI, live HIP-event timing): the sampler's
        # launches above are small (one chain group), these show where the kernel goes with more evaluations per launch
        rng = np.random.default_rng(3)
        for Bs, reps in ((a.chains, 20), (10 * a.chains, 5)):
            Pm = np.tile(star.params, (Bs, 1))
            Pm[1:, star.index_to_relax] *= 1 + 0.002 * rng.standard_normal((Bs - 1, star.nvars))
            Tm = lam ** (np.arange(Bs) % a.chains)
            ctx.loglike_params_batch(star.model_id, Pm, star.plength, Tm)
            ctx.reset_kernel_stats()
            for _ in range(reps):
                ctx.loglike_params_batch(star.model_id, Pm, star.plength, Tm)
            s_ms, s_l, s_e = ctx.kernel_stats()
            us = s_ms / max(s_l, 1) * 1e3
            shapes.append({"evaluations_per_launch": Bs, "kernel_us_per_launch": us, "achieved_GBps": 16.0 * a.nx * Bs / (us * 1e-6) / 1e9,
                           "frac": 16.0 * a.nx * Bs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})

    if rank == 0:
        st_tab, mults, _, _ = pkg.build_mode_table(star.model_id, star.params, star.plength, star.x)
        W = int(((mults["i1"] - mults["i0"]) * (2 * mults["l"] + 1)).sum())
        evals_per_launch = k_evals / max(k_launches, 1)
        k_s = k_ms * 1e-3 / max(k_launches, 1)
        alg_bytes = 16.0 * a.nx * evals_per_launch          # B_eval = 16*Nx bytes per evaluation
        achieved = alg_bytes / k_s / 1e9
        out = {
            "metric": "MCMC samples/sec (whole node), 1e5 nu-bins x 100 params x 20 tempered chains",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3 global MS fit: model_MS_Global_aj_HarveyLike, Nx={a.nx}, 111 params ({star.nvars} free), "
                                   f"{len(mults)} multiplets, {a.chains} tempered chains (lambda={lam}), one star per GPU",
                       "sampler": "adaptive random-walk MH + parallel tempering (use_drift=0, the reference's sampler)"
                       if a.sampler == "mh" else "Langevin drift, forward-difference gradient (use_drift=1)",
                       "engine": ("device-resident iteration (propose/prior/unpack/accept/swap kernels, no host round trip)"
                                  if (a.engine == "device" and a.sampler == "mh") else "host-driven loop"),
                       "arithmetic": a.precision, "component_bin_evals_per_model": W},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic()[0], "traffic_unit": "bytes per launch",
                         "traffic_source": "profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                           "passes, FETCH_SIZE x2 (gfx950; the guide calibrates the factor on 16 B/lane reads, this kernel reads 8 B/lane, so the "
                                           "absolute is uncertain within that factor -- either way far below the algorithmic bytes: the spectrum is "
                                           "served from L2), launch shape: " + pmc_traffic()[1] + "; not re-measured live",
                         "kernel": "k_loglike",
                         "kernel_us_per_launch": k_s * 1e6, "evaluations_per_launch": evals_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "launch_note": "the device sampler launches the chains as chain groups on separate streams (TAMCMC_CHAIN_GROUPS, "
                                        "default 2): each timed launch carries evaluations_per_launch evaluations and overlaps the other "
                                        "group's proposal kernel",
                         "other_launch_shapes": shapes,
                         "fp64_valu": {"component_evals_per_s": W * evals_per_launch / k_s,
                                       "note": "the path is fp64-VALU-bound (~110 Lorentzian components per 16 B); FAST mode folds far "
                                               "components into one polynomial per tile, so this is an EFFECTIVE rate"}},
            "accept_rate_chain0": (st["accepted0"] - acc0["accepted0"]) / max(a.steps, 1),
            "swap_rate": st["swaps"] / max(st["swap_attempts"], 1),
            "kernel_time_fraction": k_ms * 1e-3 / elapsed,
        }
        out.update(extra)
        mark("other launch shapes, mode table for the report")
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(star, y, a.chains, lam, a.cpu_seconds)
            mark("cpu_baseline leg (CPU restatement, bounded sample)")
        out["wall_seconds"] = {"total_so_far": round(time.perf_counter() - t_proc, 3), "parts": marks,
                               "note": "`value` = steps / TIMED REGION only; everything else is set-up, warm-up and the extra legs reported above"}
        print(json.dumps(out), flush=True)
    smp.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
