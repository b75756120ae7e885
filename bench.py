#!/usr/bin/env python3
"""bench.py -- MCMC samples/s of the MI355X hot path on BASELINE.json's headline configuration.

One "step" = one MCMC iteration of one star = ALL tempered chains advanced once
(propose -> batched Lorentzian-sum model + chi^2(2 d.o.f.) log-likelihood on the GPU -> accept -> PT swap -> sample recorded),
i.e. the reference's loop counter i (MALA.cpp:623,743).  Workload at N=1: configs[2] (C3) of BASELINE.json:
global MS fit, model_MS_Global_aj_HarveyLike, 1e5 bins x 111 parameters (93 free) x 20 tempered chains, synthetic star.
N>1 (torchrun, one rank per GPU): one independent star per GPU, no data-path collective -> weak scaling;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

Phases, like the reference's processing chain Burn-in -> Learning -> Acquire (Config/config_presets.cfg:26-31): the sampler's set-up
runs a burn-in + learning stretch (adaptation of the proposal law on, untimed: it brings the chains and the proposal scales to
the steady state and primes every launch pattern); then W warm-up and K timed iterations of the ACQUIRE phase (no adaptation).
The timed region records the samples and the statistics of every iteration (MALA.cpp:706-716) and copies them back.

The spectrum is resident in HBM before the timed region; the iteration itself runs on the device (no per-step host traffic).
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
# kernel arguments in device memory: the iteration is a chain of short dependent launches, and with the arguments fetched from host memory
# every launch starts later (measured on this image, where 1 is already the default: 47.8 k samples/s against 40.1 k with 0)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_FMA_PEAK = 39.3e12          # vector fp64: 78.6 TFLOP/s spec = 39.3e12 fused multiply-adds per second
SETUP_LEARN = (100, 1100)        # set-up phase: adaptation in iterations [100, 1100)
SETUP_ITERS = 1500               # ... of SETUP_ITERS burn-in + learning iterations


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
    ap.add_argument("--mala-steps", type=int, default=60, help="extra MALA-FD measurement (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--packed-stars", type=int, default=4,
                    help="extra leg at N=1: this many independent C3 stars co-resident on the GPU (0 = skip); reported under 'packed', "
                         "never as 'value' (the headline stays one star per GPU, BASELINE configs[2])")
    ap.add_argument("--rgb-steps", type=int, default=150,
                    help="extra leg at N=1: BASELINE configs[4] family (red giant, model id 25, 2e5 bins, 40 chains; host-driven engine with the "
                         "mixed-mode solver on the device), this many timed iterations (0 = skip); reported under 'c5_rgb'")
    ap.add_argument("--learn-steps", type=int, default=1500,
                    help="extra leg at N=1: the same star in the LEARNING phase -- the proposal law adapted after every test (MALA.cpp:656-667, "
                         "config_default.cfg:17-18), this many timed iterations (0 = skip); reported under 'learning', never as 'value'")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline run (no packed / red-giant / MALA / launch-shape / CPU legs): the command profiled under "
                         "rocprofv3 for profiles/, so that the per-kernel averages are those of the headline launches")
    ap.add_argument("--dn-mixing", type=int, default=1, help="parallel-tempering swap attempt every N iterations (reference default 1, config_default.cfg:28)")
    ap.add_argument("--step-scheme", type=int, default=0, choices=[0, 1, 2, 3],
                    help="device engine: 0 = fused launches (one or two per iteration: automatic), 1 = lockstep kernels only, 2 = fused, one launch per iteration, "
                         "3 = fused, two chain groups")
    ap.add_argument("--dump-samples", default="", help="every rank saves the samples of its timed region to <this>_rank<r>.npy (multi-rank rehearsal test)")
    a = ap.parse_args()
    a.steps = max(a.steps, 1)
    a.warmup = max(a.warmup, 0)
    if a.headline_only:
        a.mala_steps, a.packed_stars, a.rgb_steps, a.learn_steps, a.no_cpu_baseline = 0, 0, 0, 0, True
    return a


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    return {"value": n / el, "unit": "samples/s", "cores": cores, "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{n} batches of {nchains} chain evaluations (model_MS_Global_aj_HarveyLike + chi22p, Nx={star.x.size}) "
                      f"in {el:.1f} s; oracle/tamcmc_oracle.c -O3 -march=x86-64-v3, OpenMP over chains as MALA.cpp:648; "
                      "hot path only (no proposal/Cholesky/output cost), so it flatters the CPU"}


def committed_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


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
    ctx.set_option(pkg.OPT_STEP_SCHEME, a.step_scheme)
    # synthetic spectrum y = M(theta_true) * Exp(1): the model row comes from the GPU path itself (STRICT arithmetic)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    y = star.set_spectrum_from_model(m0[0], seed=20240301 + rank)
    ctx.set_option(pkg.OPT_PRECISION, prec)
    ctx.set_spectrum(star.x, y)
    T = lam ** np.arange(a.chains)

    def make_sampler(use_drift, engine, learn=SETUP_LEARN):
        return pkg.Sampler(ctx, star, nchains=a.chains, lambda_temp=lam, use_drift=use_drift, seed=7 + rank, engine=engine, Nt_learn=learn,
                           periods_learn=(1,), dN_mixing=a.dn_mixing, c0=2.0)

    from tamcmc_c_amd import shard
    mark("imports, library load, synthetic star, spectrum upload")
    use_drift = 1 if a.sampler == "mala" else 0
    smp = make_sampler(use_drift, a.engine, learn=(100, 300) if use_drift else SETUP_LEARN)  # (mala: its set-up phase is 300 iterations)
    # record buffers the run() calls fill: the samples and the statistics of every iteration (page-locked, like a writer's ring buffer;
    # the device engine writes the records straight into them)
    nrec = max(a.steps, a.warmup, 1)
    buf_smp, buf_st = pkg.pinned_empty((nrec, a.chains, smp.nvars)), pkg.pinned_empty((nrec, a.chains, 3))
    # ---- set-up phase (untimed): burn-in + learning; its last stretch already runs the acquire-phase launch pattern, and its last
    # iterations are recorded into the buffers of the timed region (first use of the record path, of the buffers' pages, of the call shape)
    n_setup = SETUP_ITERS if not use_drift else 300
    n_prime = min(nrec, 256, n_setup // 4)
    smp.run(n_setup - n_prime, record=False)
    smp.run(n_prime, out=(buf_smp[:n_prime], buf_st[:n_prime]))
    mark(f"sampler set-up: burn-in + learning phase ({n_setup} iterations, adaptation from iteration {SETUP_LEARN[0]} on)")
    acc0 = smp.state()          # (before the warm-up: reading the state synchronises and would leave the GPU idle in front of the timed region)
    if a.warmup > 0:
        smp.run(a.warmup, out=(buf_smp[:a.warmup], buf_st[:a.warmup]))
    mark("warm-up steps")
    ctx.reset_kernel_stats()
    # barrier + synchronize on both sides, MAX over ranks (tests/test_multirank_gloo.py covers this on gloo)
    elapsed, rec = shard.timed_region(lambda: smp.run(a.steps, out=(buf_smp[:a.steps], buf_st[:a.steps])), dist=dist, sync=torch.cuda.synchronize)
    k_ms, k_launches, k_evals = ctx.kernel_stats()
    st = smp.state()
    value = shard.aggregate_rate(a.steps, world, elapsed)
    mark("TIMED REGION (the K steps behind `value`), fences included")

    # ---- end-to-end check of what the timed region left behind: the chains' tempered logL recomputed with STRICT arithmetic from the
    # final positions (the recorded last sample), and the recorded statistics against the state
    samples, stats = rec
    if a.dump_samples:
        np.save(f"{a.dump_samples}_rank{rank}.npy", np.array(samples))
    Pfin = np.tile(star.params, (a.chains, 1))
    Pfin[:, star.index_to_relax] = st["vars"]
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_STRICT)
    strict_logL, _, strict_status = ctx.loglike_params_batch(star.model_id, Pfin, star.plength, T)
    ctx.set_option(pkg.OPT_PRECISION, prec)
    end_err = float(np.max(np.abs(strict_logL - st["logL"]) / np.abs(strict_logL)))
    assert (strict_status == 0).all() and end_err <= 1e-11, f"end-to-end check failed: {end_err}"
    assert np.array_equal(samples[-1], st["vars"]) and np.array_equal(stats[-1][:, 0], st["logL"]), "recorded samples do not match the final state"
    moved = np.mean(np.any(samples[1:] != samples[:-1], axis=2), axis=0) if a.steps > 1 else np.zeros(a.chains)
    mark("end-to-end check (STRICT re-evaluation of the final states)")

    extra = {}
    # (the red-giant leg runs first among the extra legs: measured right after another leg has released its device memory -- the Langevin
    #  leg's 100 MB of planes and tables -- the same kernels run 18 % slower, 3.9 k instead of 4.8 k iterations/s; the allocator hands the
    #  new context recycled memory.  A fresh process, which is how a red giant is fitted, does not see that.)
    if a.rgb_steps > 0 and a.sampler == "mh" and world == 1:
        # BASELINE configs[4] family: red-giant star, mixed modes solved per proposal (csrc/rgb_prestep.hip), 40 tempered chains
        rs = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
        rc = pkg.HipContext(device_index, precision=prec, timing=True)
        rc.set_spectrum(rs.x, np.ones_like(rs.x))
        _, mr, _ = rc.loglike_params_batch(rs.model_id, rs.params, rs.plength, want_model=True)
        rs.set_spectrum_from_model(mr[0], 7)
        rc.set_spectrum(rs.x, rs.y)
        def c5_run(engine):
            smp_ = pkg.Sampler(rc, rs, nchains=40, lambda_temp=1.15, seed=5, engine=engine, Nt_learn=(10, 200), periods_learn=(1,))
            smp_.run(250, record=False)
            smp_.run(20, record=True)                      # the record buffers exist before the clock starts
            rc.reset_kernel_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            got, _ = smp_.run(a.rgb_steps, record=True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            ks = rc.kernel_stats()
            smp_.close()
            return el, ks, got
        e_host, _, _ = c5_run("host")
        e1, (r_ms, r_l, r_e), rsm = c5_run("device")
        racc = np.mean(np.any(rsm[1:] != rsm[:-1], axis=2), axis=0) if a.rgb_steps > 1 else np.zeros(40)
        r_bytes = 16.0 * 200000 * r_e
        extra["c5_rgb"] = {"samples_per_s": a.rgb_steps / e1, "ms_per_step": 1e3 * e1 / a.rgb_steps, "steps": a.rgb_steps,
                           "workload": f"model_RGB_asympt_aj_AppWidth_HarveyLike_v4, Nx=200000, {rs.params.size} params ({rs.nvars} free), "
                                       "40 tempered chains, ~150 mixed modes per chain from the device ARMM solver; device-resident engine "
                                       "(proposal + prior, scalar unpack, solver, zeta, rows, likelihood, MH test: all enqueued, no host round trip)",
                           "host_driven_engine_samples_per_s": a.rgb_steps / e_host,
                           "accept_rate_chain0": float(racc[0]), "accept_rate_mean": float(racc.mean()),
                           "roofline": {"bound": "hbm", "kernel": "k_loglike (one launch per chain group: 10 evaluations x 2e5 bins, the groups' launches overlap; the pre-step kernels are not in this time)",
                                        "kernel_us_per_launch": r_ms / max(r_l, 1) * 1e3, "evaluations_per_launch": r_e / max(r_l, 1),
                                        "algorithmic_bytes_per_launch": r_bytes / max(r_l, 1), "achieved": r_bytes / max(r_ms * 1e-3, 1e-12) / 1e9,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_bytes / max(r_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS}}
        rc.close()
        mark("extra leg: c5_rgb")

    if a.mala_steps > 0 and a.sampler == "mh" and world == 1:
        # adaptation in [100, 300), as the 300 set-up iterations below: the Langevin proposal needs its step size tuned; the timed steps are
        # acquire-phase steps like the headline's (rounds 1-2 left the learning window open -- (100, 1100) -- so every timed step also
        # paid a covariance update and a Cholesky factor, ~45 us of the step)
        ms = make_sampler(1, a.engine, learn=(100, 300))
        ms.run(300, record=False)
        ctx.reset_kernel_stats()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        msmp, _ = ms.run(a.mala_steps, record=True)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        mk_ms, mk_l, mk_e = ctx.kernel_stats()
        fd_bins, fd_evals = ctx.fd_stats()
        # bytes the FD launches touch: the C base evaluations read x, y and write the model row (24 B x Nx each); a delta evaluation
        # reads x, y and the base model row on its affected range only (24 B per affected bin)
        mala_bytes = 40.0 * a.nx * a.chains * mk_l + 24.0 * fd_bins
        macc = np.mean(np.any(msmp[1:] != msmp[:-1], axis=2), axis=0) if a.mala_steps > 1 else np.zeros(a.chains)
        extra["mala_fd"] = {"samples_per_s": a.mala_steps / e1, "steps": a.mala_steps, "evals_per_step": mk_e / a.mala_steps,
                            "engine": ("device-resident Langevin step (k_mala_settle -> finite-difference batch -> k_mala_test, nothing crosses PCIe)"
                                       if a.engine == "device" else "host-driven loop + device finite-difference batches") + ", windowed delta tables",
                            "accept_rate_chain0": float(macc[0]), "accept_rate_mean": float(macc.mean()),
                            # SURVEY 8(d)'s definition (16 B x Nx per evaluation, every one of the C x (Nvars + 1) evaluations of a step counted in
                            # full) beside the bytes the launches really touch: the windowed differences do not read most of those bytes at all
                            "roofline_8d": {"algorithmic_bytes_per_step": 16.0 * a.nx * mk_e / a.mala_steps,
                                            "achieved_GBps": 16.0 * a.nx * mk_e / e1 / 1e9, "frac_of_hbm_peak": 16.0 * a.nx * mk_e / e1 / 1e9 / HBM_PEAK_GBS,
                                            "note": "an ALGORITHMIC rate (work avoided counts as done): above 1 means the step is faster than streaming "
                                                    "every evaluation's x and y once would allow"},
                            "roofline": {"bound": "hbm", "kernel": "base k_loglike (model rows kept) + k_loglike<DELTA> per FD batch",
                                         "kernel_us_per_batch": mk_ms / max(mk_l, 1) * 1e3,
                                         "bytes_touched_per_batch": mala_bytes / max(mk_l, 1),
                                         "mean_affected_bins_per_delta_evaluation": fd_bins / max(fd_evals, 1),
                                         "achieved": mala_bytes / max(mk_ms * 1e-3, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": mala_bytes / max(mk_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                                         "note": "bytes = 40 B x Nx per base evaluation (x, y read; three planes written) + 24 B per bin the delta "
                                                 "evaluations walk (affected range minus the far-only tiles taken from the base point's moments); "
                                                 "not 16 B x Nx per evaluation: a perturbed mode parameter changes the model inside its window only"}}
        ms.close()
        mark("extra leg: mala_fd")

    if a.learn_steps > 0 and a.sampler == "mh" and world == 1:
        # the Learning phase of the reference's processing chain (config_presets.cfg:26-31; by default 700 000 of a run's samples): every
        # iteration ends with the Robbins-Monro update of mu, Sigma, sigma and a new Cholesky factor (MALA.cpp:296-319, :339-350)
        ls = pkg.Sampler(ctx, star, nchains=a.chains, lambda_temp=lam, seed=11 + rank, engine=a.engine, Nt_learn=(100, 10**9), periods_learn=(1,),
                         dN_mixing=a.dn_mixing, c0=2.0)
        ls.run(400, record=False)
        lb = pkg.pinned_empty((a.learn_steps, a.chains, ls.nvars)), pkg.pinned_empty((a.learn_steps, a.chains, 3))
        ls.run(min(64, a.learn_steps), out=(lb[0][:min(64, a.learn_steps)], lb[1][:min(64, a.learn_steps)]))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ls.run(a.learn_steps, out=lb)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        lacc = np.mean(np.any(lb[0][1:] != lb[0][:-1], axis=2), axis=0)
        extra["learning"] = {"iterations_per_s": a.learn_steps / e1, "us_per_iteration": 1e6 * e1 / a.learn_steps, "steps": a.learn_steps,
                             "what": "C3 star, adaptation of the proposal law after every MH test (Robbins-Monro update + Cholesky factor of the "
                                     f"{ls.nvars} x {ls.nvars} covariance per chain per iteration), samples and statistics recorded",
                             "accept_rate_chain0": float(lacc[0]), "sigma_chain0": float(ls.state()["sigma"][0]),
                             "alg_GBps": a.learn_steps / e1 * a.chains * 16.0 * a.nx / 1e9,
                             "frac_of_hbm_peak": a.learn_steps / e1 * a.chains * 16.0 * a.nx / 1e9 / HBM_PEAK_GBS}
        ls.close()
        mark("extra leg: learning phase")

    if a.packed_stars > 1 and a.sampler == "mh" and a.engine == "device" and world == 1:
        # Several independent stars on ONE GPU (one context + one device-resident sampler + one host thread per star,
        # tamcmc_sampler_run_packed): co-resident stars fill the SIMDs a single star's iteration leaves idle.
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
                                         Nt_learn=SETUP_LEARN, periods_learn=(1,), dN_mixing=a.dn_mixing, c0=2.0)))
        ps = [q[1] for q in pool]
        smod.run_packed(ps, SETUP_ITERS, record=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        smod.run_packed(ps, a.steps, record=True, stats=True)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        rate = a.packed_stars * a.steps / e1
        extra["packed"] = {"stars_per_gpu": a.packed_stars, "samples_per_s": rate, "us_per_star_iteration": 1e6 * e1 / (a.steps * a.packed_stars),
                           "alg_GBps": rate * a.chains * 16.0 * a.nx / 1e9, "frac_of_hbm_peak": rate * a.chains * 16.0 * a.nx / 1e9 / HBM_PEAK_GBS,
                           "note": "aggregate over the co-resident stars (each a full 20-chain C3 fit on its own stream); "
                                   "every star's samples are bit-identical to its solo run (tests/test_gpu_sampler.py)"}
        for ck, sk_ in pool:
            sk_.close()
            ck.close()
        mark("extra leg: packed (set-up of the co-resident stars + their learning phase + timed steps)")

    shapes = []
    if world == 1 and a.sampler == "mh" and not a.headline_only:
        # the likelihood kernel on its own at other launch sizes (standalone batched calls through the C ABI, live HIP-event timing)
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
            shapes.append({"kernel": "k_loglike (standalone)", "evaluations_per_launch": Bs, "kernel_us_per_launch": us,
                           "achieved_GBps": 16.0 * a.nx * Bs / (us * 1e-6) / 1e9, "frac": 16.0 * a.nx * Bs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})

    if rank == 0:
        st_tab, mults, _, _ = pkg.build_mode_table(star.model_id, star.params, star.plength, star.x)
        W = int(((mults["i1"] - mults["i0"]) * (2 * mults["l"] + 1)).sum())
        evals_per_launch = k_evals / max(k_launches, 1)
        k_s = k_ms * 1e-3 / max(k_launches, 1)
        alg_bytes = 16.0 * a.nx * evals_per_launch          # B_eval = 16*Nx bytes per evaluation
        if k_launches <= 0 or k_s <= 0:
            raise SystemExit("bench.py: no launch of the dominant kernel was timed inside the timed region (no roofline without a live measurement)")
        achieved = alg_bytes / k_s / 1e9                     # one launch of the dominant kernel
        agg = value / world * a.chains * 16.0 * a.nx / 1e9   # the whole timed region
        launches_per_iter = k_launches / max(a.steps, 1)
        fused = (a.engine == "device" and a.sampler == "mh" and a.step_scheme != 1 and prec == pkg.PRECISION_FAST)
        pmc = committed_json("r03_pmc_traffic.json") or committed_json("r02_pmc_traffic.json") or {}
        pmc_sq = committed_json("r03_pmc_sq.json") or committed_json("r02_pmc_sq.json") or {}
        valu = None
        if pmc_sq.get("SQ_INSTS_VALU_per_launch"):
            # wave-level VALU instructions x 64 lanes / duration: an upper bound of the fp64 lane-operation rate (integer/address VALU included)
            rate = pmc_sq["SQ_INSTS_VALU_per_launch"] * 64.0 * launches_per_iter / (elapsed / a.steps)
            valu = {"valu_lane_ops_per_s": rate, "frac_of_fp64_fma_peak": rate / FP64_FMA_PEAK, "peak_fma_per_s": FP64_FMA_PEAK,
                    "SQ_INSTS_VALU_per_launch": pmc_sq["SQ_INSTS_VALU_per_launch"], "launches_in_flight": round(launches_per_iter),
                    "source": "profiles/r0N_pmc_sq.json of the newest round (rocprofv3 --pmc, separate pass): VALU instructions of one launch x 64 lanes x launches per iteration / iteration time"}
        out = {
            "metric": "MCMC samples/sec (whole node), 1e5 nu-bins x 100 params x 20 tempered chains",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3 global MS fit: model_MS_Global_aj_HarveyLike, Nx={a.nx}, 111 params ({star.nvars} free), "
                                   f"{len(mults)} multiplets, {a.chains} tempered chains (lambda={lam}), one star per GPU",
                       "sampler": "adaptive random-walk MH + parallel tempering (use_drift=0, the reference's sampler)"
                       if a.sampler == "mh" else "Langevin drift, forward-difference gradient (use_drift=1)",
                       "engine": (("device-resident iteration, fused launches (likelihood tiles of iteration i, each deciding iteration i-1 for its chain first + commit "
                                   "workgroups + iteration i+1's candidates), one per chain group and iteration on two streams"
                                   if fused else "device-resident iteration, lockstep kernels (k_iterate, k_loglike)")
                                  if (a.engine == "device" and a.sampler == "mh") else "host-driven loop"),
                       "phases": f"set-up: {SETUP_ITERS} burn-in + learning iterations with adaptation in {SETUP_LEARN}, the last {min(max(a.steps, a.warmup, 1), 256)} of them recorded like the timed ones (untimed); then {a.warmup} warm-up + "
                                 f"{a.steps} timed iterations of the acquire phase; samples and statistics of every timed iteration recorded and copied back",
                       "arithmetic": a.precision, "component_bin_evals_per_model": W, "dN_mixing": a.dn_mixing},
            # SURVEY 8(d) / VERDICT r1: frac = samples/s x bytes/sample / peak, bytes/sample = Nchains x 16 x Nx (every launch of the timed
            # region counted, over its wall time); the per-launch figures of the dominant kernel -- the ones rocprofv3's average duration
            # must agree with -- are in `kernel`
            "roofline": {"bound": "hbm", "achieved": agg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg / HBM_PEAK_GBS,
                         "formula": "samples/s x Nchains x 16 x Nx / peak (algorithmic bytes of all launches of the timed region / its wall time)",
                         "traffic": (pmc.get("hbm_bytes_per_launch") * launches_per_iter) if pmc.get("hbm_bytes_per_launch") else None,
                         "traffic_unit": "bytes per iteration (PMC bytes per launch x launches per iteration)",
                         "traffic_source": pmc.get("source", "no committed PMC pass for this kernel yet"),
                         "kernel": {"name": "k_step<FAST, K=8> (fused step: one launch per chain group and iteration)" if fused else "k_loglike",
                                    "us_per_launch": k_s * 1e6, "evaluations_per_launch": evals_per_launch,
                                    "algorithmic_bytes_per_launch": alg_bytes, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                                    "launches_per_iteration": launches_per_iter,
                                    "traffic_bytes_per_launch": pmc.get("hbm_bytes_per_launch"),
                                    "how_measured": ("HIP events stamped at the start and at the end of sampled k_step launches of the timed region "
                                                     "(hipExtLaunchKernelGGL; every 97th iteration, second chain group's stream); an iteration is two such launches, one per "
                                                     "chain group, in flight together on two streams (one joint launch when the swap pair straddles "
                                                     "the groups): a launch lasts about one iteration period and the GPU's rate is the sum of the two"
                                                     if fused else "HIP events around sampled k_loglike launches on their stream")},
                         "valu": valu,
                         "other_launch_shapes": shapes,
                         "fp64_valu": {"component_evals_per_s": W * a.chains * value / world,
                                       "note": "the path is fp64-VALU-bound (~110 Lorentzian components per 16 B); FAST mode folds far "
                                               "components into one polynomial per tile, so this is an EFFECTIVE rate"}},
            "accept_rate_chain0": (st["accepted0"] - acc0["accepted0"]) / max(a.steps + a.warmup, 1),   # (warm-up + timed iterations)
            "position_change_rate_mean": float(moved.mean()),
            "swap_rate": (st["swaps"] - acc0["swaps"]) / max(st["swap_attempts"] - acc0["swap_attempts"], 1),
            "end_to_end_check": {"max_rel_err_logL_final_states_vs_STRICT": end_err, "tolerance": 1e-11,
                                 "recorded_last_sample_equals_state": True},
            "kernel_time_fraction": k_ms * 1e-3 / elapsed,   # (> 1 with two launches in flight)
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
