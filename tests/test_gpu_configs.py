"""BASELINE.json configurations that had no -m gpu test of their own: C4 = eight independent C3-size stars, C5 = a
red-giant star at 2e5 bins with a 40-vector batch / 40 tempered chains, and the N > 1 launch path of bench.py rehearsed with two
gloo ranks that run the REAL sampler path on the one GPU of the box (one star per rank, no collective on the data path)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c3_with_spectrum(pkg, synth, seed_off, nx=100000):
    star = synth.make_c3_star(seed=20240229 + seed_off, nx=nx, step=2000.0 / nx)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    star.set_spectrum_from_model(m0[0], seed=20240301 + seed_off)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    return star, ctx


def test_c4_eight_c3_stars_on_one_gpu_equal_their_solo_runs(pkg, synth):
    """C4 workload on the one GPU: 8 C3-size stars (1e5 bins x 111 parameters x 20 chains each, seeds +0..+7) through
    tamcmc_sampler_run_packed -- learning window, fused acquire steps and a second call included -- each star's samples and statistics
    bit-identical to its solo run.  (On an 8-GPU node the same stars run one per rank: bench.py --gpus 8.)"""
    from tamcmc_c_amd import sampler as smod
    S = 8
    kw = dict(nchains=20, lambda_temp=1.3, engine="device", Nt_learn=(8, 20), periods_learn=(1,), dN_mixing=1, c0=2.0, chain_groups=1)
    pool = [_c3_with_spectrum(pkg, synth, k) for k in range(S)]
    solo = []
    for k, (star, ctx) in enumerate(pool):
        s = pkg.Sampler(ctx, star, seed=300 + k, **kw)
        a1, b1 = s.run(45, stats=True)
        a2, b2 = s.run(15, stats=True)
        solo.append((np.concatenate([a1, a2]), np.concatenate([b1, b2]), s.state()))
        s.close()
    packed = [pkg.Sampler(ctx, star, seed=300 + k, **kw) for k, (star, ctx) in enumerate(pool)]
    p1, q1 = smod.run_packed(packed, 45, stats=True)
    p2, q2 = smod.run_packed(packed, 15, stats=True)
    for k in range(S):
        assert np.array_equal(np.concatenate([p1[k], p2[k]]), solo[k][0]), k
        assert np.array_equal(np.concatenate([q1[k], q2[k]]), solo[k][1]), k
        st = packed[k].state()
        assert st["iteration"] == 60 and st["swaps"] == solo[k][2]["swaps"] and np.array_equal(st["vars"], solo[k][2]["vars"])
    assert not np.array_equal(solo[0][0], solo[1][0])      # different stars, different chains
    for s in packed:
        s.close()
    for _, ctx in pool:
        ctx.close()


def test_c5_forty_vector_batch_at_2e5_bins(pkg, oracle, synth):
    """C5 as specified: ONE batch of 40 red-giant parameter vectors at 2e5 bins (model id 25, ~150 mixed modes per vector from the
    device solver).  Three vectors against the oracle (the red-giant tolerance: 1e-8 relative, tests/test_gpu_rgb.py), and on all 40
    the size-independent properties of the path: finite, status OK, the tempered value times its temperature does not depend on the
    temperature nor on the vector's place in the batch (bitwise), equal vectors give equal values."""
    rs = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(rs.x, np.ones_like(rs.x))
    _, mr, _ = ctx.loglike_params_batch(rs.model_id, rs.params, rs.plength, want_model=True)
    rs.set_spectrum_from_model(mr[0], 7)
    ctx.set_spectrum(rs.x, rs.y)
    rng = np.random.default_rng(12)
    B = 40
    P = np.tile(rs.params, (B, 1))
    idx = rs.index_to_relax
    P[1:, idx] *= 1.0 + 2e-4 * rng.standard_normal((B - 1, idx.size))
    P[17] = P[3]                                            # a repeated vector
    T = 1.15 ** np.arange(B)
    L1, _, st1 = ctx.loglike_params_batch(rs.model_id, P, rs.plength, T)
    assert (st1 == 0).all() and np.all(np.isfinite(L1)) and rs.x.size == 200000
    perm = rng.permutation(B)
    T2 = 1.15 ** rng.integers(0, B, B)                      # other temperatures, other places
    L2, _, st2 = ctx.loglike_params_batch(rs.model_id, P[perm], rs.plength, T2)
    assert (st2 == 0).all()
    untempered1, untempered2 = L1 * T, np.empty(B)
    untempered2[perm] = L2 * T2
    assert np.allclose(untempered1, untempered2, rtol=4e-16, atol=0)     # (L/T)*T: one rounding each way
    assert np.isclose(L1[17] * T[17], L1[3] * T[3], rtol=4e-16)
    for b in (0, 9, 31):
        ref, _, so = oracle.loglike_batch(rs.model_id, P[b], rs.plength, rs.x, rs.y, 1.0, T[b:b + 1])
        assert so[0] == 0 and abs(L1[b] - ref[0]) <= 1e-8 * abs(ref[0]), (b, L1[b], ref[0])
    ctx.close()


@pytest.mark.parametrize("engine", ["host", "device"])
def test_c5_forty_chain_sampler_run(pkg, synth, engine):
    """40 tempered chains (the reference caps Nchains at 24, MALA.cpp:580-587) on the 2e5-bin red giant, the solver on the device.
    host: one batched call per iteration; device: the whole iteration enqueued (proposal + unpack, solver, rows, likelihood per chain
    group, four groups).  Finite statistics, chains move, swaps happen, adaptation keeps the covariance matrices symmetric."""
    rs = synth.make_c5_star(nx=200000, nmax=10, dnu=10.0, bias_type=1, nferr=6)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(rs.x, np.ones_like(rs.x))
    _, mr, _ = ctx.loglike_params_batch(rs.model_id, rs.params, rs.plength, want_model=True)
    rs.set_spectrum_from_model(mr[0], 7)
    ctx.set_spectrum(rs.x, rs.y)
    s = pkg.Sampler(ctx, rs, nchains=40, lambda_temp=1.15, seed=5, engine=engine, Nt_learn=(10, 120), periods_learn=(1,))
    smp, st = s.run(200, stats=True)
    state = s.state()
    assert smp.shape == (200, 40, rs.nvars) and np.all(np.isfinite(st[:, :, 0])) and state["iteration"] == 200
    assert state["swap_attempts"] == 199 and 0 < state["swaps"] < 199
    assert np.mean(np.any(smp[1:] != smp[:-1], axis=2)) > 0.05
    mu, cov = s.get_proposal(0)
    assert np.allclose(cov, cov.T, rtol=1e-12, atol=1e-300) and np.all(np.diag(cov) > 0)
    if engine == "device":          # the chain groups are a launch arrangement only: same chains with one group
        s1 = pkg.Sampler(ctx, rs, nchains=40, lambda_temp=1.15, seed=5, engine="device", Nt_learn=(10, 120), periods_learn=(1,), chain_groups=1)
        smp1, st1 = s1.run(200, stats=True)
        assert np.array_equal(smp1, smp) and np.array_equal(st1, st)
        s1.close()
    s.close(); ctx.close()


def test_headline_shape_two_group_fused_steps_equal_the_lockstep_chain(pkg, synth):
    """C3 (1e5 bins, 20 chains, swaps every iteration): 1500 acquire iterations as fused steps -- two launches per iteration, one per
    chain group on its own stream, one joint launch whenever the swap pair straddles the groups -- against the lockstep kernels and
    against the one-launch fused steps: identical samples and statistics (the groups' launches drift several iterations apart between
    two joint iterations; nothing one group writes may be read by the other before they meet)."""
    star, ctx = _c3_with_spectrum(pkg, synth, 0)
    kw = dict(nchains=20, lambda_temp=1.3, seed=11, engine="device", Nt_learn=(10, 60), periods_learn=(1,), dN_mixing=1, c0=2.0)
    runs = []
    for scheme in (1, 0, 2):      # lockstep | automatic (two groups at this size) | one launch per iteration
        ctx.set_option(pkg.OPT_STEP_SCHEME, scheme)
        s = pkg.Sampler(ctx, star, **kw)
        a1, b1 = s.run(100, stats=True)
        a2, b2 = s.run(1500, stats=True)
        runs.append((np.concatenate([a1, a2]), np.concatenate([b1, b2]), s.state()))
        s.close()
    ctx.set_option(pkg.OPT_STEP_SCHEME, 0)
    for k in (1, 2):
        assert np.array_equal(runs[k][0], runs[0][0]) and np.array_equal(runs[k][1], runs[0][1]), k
        assert runs[k][2]["swaps"] == runs[0][2]["swaps"] and np.array_equal(runs[k][2]["vars"], runs[0][2]["vars"])
    assert 100 < runs[0][2]["swaps"] < 1599
    ctx.close()


def test_headline_shape_posterior_brackets_the_truth(pkg, synth):
    """C3 as a sampler, not only as a kernel: 20 tempered chains on the 1e5-bin star, 3000 learning iterations (lockstep kernels with the
    adaptation), then 24 000 acquire iterations as fused steps.  The coldest chain's posterior (mean / sigma per variable, what
    tools/bin2txt_params.cpp:165-168 prints) brackets the frequencies the spectrum was drawn from, its acceptance rate has settled near
    the target (0.234) and the ladder swaps."""
    star, ctx = _c3_with_spectrum(pkg, synth, 0)
    s = pkg.Sampler(ctx, star, nchains=20, lambda_temp=1.3, seed=5, engine="device", Nt_learn=(100, 3100), periods_learn=(1,), dN_mixing=1, c0=2.0)
    s.run(4000, record=False)
    a0 = s.state()
    smp, st = s.run(24000, stats=True)
    a1 = s.state()
    cold = smp[:, 0, :]
    mean, std = cold.mean(0), cold.std(0)
    truth = star.params[star.index_to_relax]
    fidx = np.array([i for i, k in enumerate(star.index_to_relax) if star.names[k] == "Frequency_l"])
    z = np.abs(mean[fidx] - truth[fidx]) / std[fidx]
    assert fidx.size == 56 and np.all(z < 5.5) and np.median(z) < 1.5, z
    assert np.all(std[fidx] > 1e-4) and np.all(std[fidx] < 2.0)
    acc = (a1["accepted0"] - a0["accepted0"]) / 24000.0
    assert 0.08 < acc < 0.5, acc
    assert 0.05 < (a1["swaps"] - a0["swaps"]) / 24000.0 < 0.95
    assert np.isfinite(st).all() and st[-2000:, 0, 2].mean() > st[:2000, 0, 2].mean() - 50.0      # no drift away from the mode
    s.close(); ctx.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_run_the_bench_path_and_match_solo_runs(pkg, synth, tmp_path):
    """bench.py --gpus 2 under torch.distributed.run with the gloo backend (TAMCMC_BENCH_BACKEND=gloo: both ranks share the box's one
    GPU): rank r fits star 20240229 + r with sampler seed 7 + r.  Each rank's recorded samples equal a solo run of the same star in
    this process; the line reports two GPUs' worth of work over the MAX of the two ranks' times."""
    steps, warm = 40, 5
    dump = str(tmp_path / "samples")
    env = dict(os.environ, TAMCMC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm), "--headline-only",
           "--dump-samples", dump]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak" and d["value"] == pytest.approx(2 * steps / (d["ms_per_step"] * 1e-3 * steps))
    sys.path.insert(0, ROOT)
    import bench
    for rank in range(2):
        got = np.load(f"{dump}_rank{rank}.npy")
        star, ctx = _c3_with_spectrum(pkg, synth, rank)
        s = pkg.Sampler(ctx, star, nchains=20, lambda_temp=1.3, seed=7 + rank, engine="device", Nt_learn=bench.SETUP_LEARN, periods_learn=(1,),
                        dN_mixing=1, c0=2.0)
        s.run(bench.SETUP_ITERS, record=False)
        s.run(warm, stats=True)
        ref, _ = s.run(steps, stats=True)
        assert got.shape == ref.shape and np.array_equal(got, ref), rank
        s.close(); ctx.close()


@pytest.mark.parametrize("steps,warm", [(20, 5), (7, 0), (100, 3)])
def test_the_drivers_bench_command_prints_its_line(steps, warm):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` is what the driver records; short windows must still carry a live measurement of
    the dominant kernel (a window whose middle iteration is a joint launch of both chain groups once left none: round 3)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warm), "--headline-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["steps"] == steps and d["warmup"] == warm and d["n_gpus"] == 1
    k = d["roofline"]["kernel"]
    assert k["us_per_launch"] > 1.0 and 0 < k["frac"] < 1 and 0 < d["roofline"]["frac"] < 1
    assert d["value"] == pytest.approx(steps / (d["ms_per_step"] * 1e-3 * steps))
    assert d["end_to_end_check"]["max_rel_err_logL_final_states_vs_STRICT"] <= 1e-11
