"""Diagnostics for the Langevin parity tests and the Monte-Carlo-resolution posterior checks (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import oracle_lib  # noqa: E402
import mc_stats  # noqa: E402

pkg = g.load_package()
from tamcmc_c_amd import synth  # noqa: E402

orc = oracle_lib.Oracle()


def grad_accuracy():
    star = synth.make_c2_star(nx=10000)
    _, m0 = orc.call_model(star.model_id, star.params, star.plength, star.x)
    y = star.set_spectrum_from_model(m0, seed=11)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, y)
    th = star.params.copy()
    idx = star.index_to_relax
    for rel in (1e-7, 1e-6, 1e-5):
        h = rel * np.maximum(np.abs(th[idx]), 1e-3)
        for T in (1.0, 1.7 ** 5):
            l0, gd = ctx.fd_gradient(star.model_id, th[None, :], star.plength, idx, h, np.array([T]))
            st, go, gpo = orc.fd_gradient_posterior(star, y, th, T, h)
            gl = go - gpo
            print("C2 h_rel %.0e T %.2f: max |dg|/max|g| %.2e ; per-comp rel %s" % (rel, T, np.max(np.abs(gd[0] - gl)) / np.max(np.abs(gl)),
                                                                            np.array2string(np.abs(gd[0] - gl) / np.maximum(np.abs(gl), 1e-3 * np.abs(gl).max()), precision=1)))
    ctx.close()


def walk(engine="device"):
    star = synth.make_c2_star(nx=10000)
    _, m0 = orc.call_model(star.model_id, star.params, star.plength, star.x)
    y = star.set_spectrum_from_model(m0, seed=11)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, y)
    nch, lam, c0 = 10, 1.7, 2.0
    T = lam ** np.arange(nch)
    s = pkg.Sampler(ctx, star, nchains=nch, lambda_temp=lam, engine=engine, use_drift=1, seed=5, Nt_learn=(4, 150), periods_learn=(1,), dN_mixing=1, c0=c0)
    init_logL = s.state()["logL"].copy()
    for it in range(12):
        st = s.state()
        params = np.tile(star.params, (nch, 1))
        params[:, star.index_to_relax] = st["vars"]
        before = dict(params=params, vars=st["vars"], logL=st["logL"], logPrior=st["logPrior"], logPost=st["logPost"])
        law = s.proposal_law()
        i = st["iteration"]
        z, u, us, ia = s.draws(i)
        learn = 4 <= i < 150
        exp, law2, rc = orc.sampler_iteration(star, y, T, init_logL, before, law, i=i, z=z, u_mh=u, learn=learn, do_swap=i != 0, ind_A=ia, u_swap=us, c0=c0,
                                              use_drift=True, fd_step_rel=1e-7)
        s.run(1)
        aft = s.state()
        disp = np.linalg.norm(exp["prop_vars"] - before["vars"], axis=1)
        # what the product proposed cannot be read directly; accepted chains show it
        acc = exp["moved"] == 1
        dev = np.linalg.norm(aft["vars"] - exp["vars"], axis=1)
        print("it %2d learn %d moved %s |drift| %s  |x'-x| %s  dev/|x'-x| %s  dPmove %s" % (
            i, learn, exp["moved"], np.array2string(exp["diag"][:, 2], precision=2), np.array2string(disp, precision=2),
            np.array2string(dev / disp, precision=1), np.array2string(np.abs(aft["Pmove"] - exp["Pmove"]), precision=1)), flush=True)
    s.close()
    ctx.close()


def ess():
    star = synth.make_c2_star(nx=4000)
    _, m0 = orc.call_model(star.model_id, star.params, star.plength, star.x)
    star.set_spectrum_from_model(m0, 5)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    for drift, n, learn_to in ((0, 150000, 4100), (0, 150000, 30100), (1, 40000, 4100)):
        s = pkg.Sampler(ctx, star, engine="device", use_drift=drift, nchains=4, lambda_temp=1.6, seed=91 + drift, Nt_learn=(100, learn_to), periods_learn=(1,), c0=5.0)
        s.run(learn_to, record=False)
        smp, _ = s.run(n)
        cold = smp[:, 0, :]
        acc = np.mean(np.any(cold[1:] != cold[:-1], axis=1))
        taus = np.array([mc_stats.tau_int(cold[:, k]) for k in range(cold.shape[1])])
        names = [star.names[i] for i in star.index_to_relax]
        print("drift %d learn_to %d n %d acc %.3f sigma %s" % (drift, learn_to, n, acc, s.state()["sigma"]))
        for k in range(cold.shape[1]):
            half = n // 2
            print("   %-28s tau %8.1f  mean %.5g  sd %.3g   half-means differ by %.2f sd" % (names[k], taus[k], cold[:, k].mean(), cold[:, k].std(),
                                                                                      (cold[:half, k].mean() - cold[half:, k].mean()) / cold[:, k].std()))
        s.close()
    ctx.close()


if __name__ == "__main__":
    for what in sys.argv[1:]:
        {"grad": grad_accuracy, "walk": walk, "ess": ess}[what]()
