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




def hostwalk():
    """Host engine, headline shape: consecutive settled iterations, hot and cold chains against the oracle."""
    star = synth.make_c3_star(seed=20240229, nx=100000, step=0.02)
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_STRICT)
    ctx.set_spectrum(star.x, np.ones_like(star.x))
    _, m0, _ = ctx.loglike_params_batch(star.model_id, star.params, star.plength, want_model=True)
    star.set_spectrum_from_model(m0[0], seed=20240301)
    ctx.set_option(pkg.OPT_PRECISION, pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    nch, lam, c0 = 20, 1.3, 2.0
    T = lam ** np.arange(nch)
    for engine in ("host", "device"):
        s = pkg.Sampler(ctx, star, nchains=nch, lambda_temp=lam, engine=engine, use_drift=1, seed=31, Nt_learn=(20, 120), periods_learn=(1,), dN_mixing=1, c0=c0)
        init_logL = s.state()["logL"].copy()
        s.run(130, record=False)
        idx = star.index_to_relax
        for rep in range(10 if engine == "host" else 0):
            st = s.state()
            if rep == 4:
                s.set_state(st["vars"], iteration=6445)
                st = s.state()
            params = np.tile(star.params, (nch, 1))
            params[:, idx] = st["vars"]
            before = dict(params=params, vars=st["vars"], logL=st["logL"], logPrior=st["logPrior"], logPost=st["logPost"])
            law = s.proposal_law()
            i = st["iteration"]
            z, u, us, ia = s.draws(i)
            mask = np.zeros(nch, dtype=np.int32)
            mask[[0, 9, 17, 18, 19, ia, ia + 1]] = 1
            gprod, gpprod, gvalid = s.gradient()
            hh = 1e-7 * np.maximum(np.abs(law[0][0]), 1e-3)
            gcmp = []
            for m in np.flatnonzero(mask):
                _, go, gpo = orc.fd_gradient_posterior(star, star.y, params[m], T[m], hh)
                e = np.abs(gprod[m] - go) / np.maximum(np.abs(go), 1e-3 * np.abs(go).max())
                gcmp.append((int(m), int(gvalid[m]), float(e.max()), int(e.argmax()), float(gprod[m][e.argmax()]), float(go[e.argmax()])))
            print("   HELD gradient vs oracle at x (chain, valid, max rel dev, variable, held, oracle):", gcmp)
            exp, law2, rc = orc.sampler_iteration(star, star.y, T, init_logL, before, law, i=i, z=z, u_mh=u, learn=False, do_swap=True, ind_A=ia, u_swap=us, c0=c0,
                                                  use_drift=True, fd_step_rel=1e-7, chain_mask=mask)
            s.run(1)
            aft = s.state()
            c = np.flatnonzero(mask)
            h = 1e-7 * np.maximum(np.abs(law[0][0]), 1e-3)
            if engine == "host":
                vp, sp, lqp, gpr_ = s.last_test()
                for m in c:
                    pm = params[m].copy(); pm[idx] = vp[m]
                    _, go, gpo = orc.fd_gradient_posterior(star, star.y, pm, T[m], h)
                    e = np.abs(gpr_[m] - go) / np.maximum(np.abs(go), 1e-3 * np.abs(go).max())
                    w = np.argsort(e)[-3:][::-1]
                    print("   HOST grad at its own x' vs oracle, chain", m, "worst (variable, name, rel dev, prod, oracle, oracle prior share):",
                          [(int(k), star.names[idx[k]], float(e[k]), float(gpr_[m][k]), float(go[k]), float(gpo[k])) for k in w])
                dx = np.linalg.norm(vp[c] - exp["prop_vars"][c], axis=1) / np.linalg.norm(exp["prop_vars"][c] - before["vars"][c], axis=1)
                def chol_ld(M):
                    n = M.shape[0]
                    L = np.zeros((n, n), dtype=np.longdouble)
                    A = M.astype(np.longdouble)
                    for j in range(n):
                        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
                        L[j, j] = np.sqrt(d)
                        for i2 in range(j + 1, n):
                            L[i2, j] = (A[i2, j] - np.dot(L[i2, :j], L[j, :j])) / L[j, j]
                    return L
                for m in (17,):
                    M = (law[1][m] + 1e-12 * np.eye(idx.size)) * law[2][m]
                    ev = np.linalg.eigvalsh((M + M.T) / 2)
                    L = chol_ld(M)
                    def whiten(v):
                        w = np.zeros(idx.size, dtype=np.longdouble)
                        for i2 in range(idx.size):
                            w[i2] = (np.longdouble(v[i2]) - np.dot(L[i2, :i2], w[:i2])) / L[i2, i2]
                        return w
                    x_ = before["vars"][m]
                    d0 = 0.5 * M @ gprod[m]
                    d1 = 0.5 * M @ gpr_[m]
                    qf = float(np.sum(whiten(vp[m] - x_ - d0) ** 2))
                    qr = float(np.sum(whiten(x_ - vp[m] - d1) ** 2))
                    # the well-conditioned route: L^-1 d = (1/2) L^T g
                    zr = -(whiten(vp[m] - x_ - d0)) - 0.5 * (L.T.astype(np.longdouble) @ (gprod[m] + gpr_[m]).astype(np.longdouble))
                    print("   chain", m, "eig(M) min %.3e max %.3e | numpy long double: lq_fwd %.6f lq_rev %.6f | via L^T g: lq_rev %.6f | prod lq_fwd %.6f lq_rev %.6f | oracle diff %.6f" % (
                        ev.min(), ev.max(), -0.5 * qf, -0.5 * qr, -0.5 * float(np.sum(zr ** 2)), lqp[m, 0], lqp[m, 1], exp["diag"][m, 0] - exp["diag"][m, 1]))
                if i == 133:
                    np.savez(os.path.join(ROOT, "gpurun_out", "r3e", "case133.npz"), params17=params[17], vars17=before["vars"][17], logPost17=before["logPost"][17],
                             mu0=law[0][0], cov17=law[1][17], sigma17=law[2][17], z17=z[17], u17=u[17], T17=T[17], vp17=vp[17], gprop17=gpr_[17], gheld17=gprod[17],
                             lq_prod=lqp[17], orc_diag=exp["diag"][17], orc_prop=exp["prop_vars"][17], orc_stats=exp["prop_stats"][17], y=star.y, init17=init_logL[17],
                             orc_Pmove=exp["Pmove"][17], prod_Pmove=aft["Pmove"][17])
                print("   HOST last test: |x'_prod - x'_orc|/|step|", np.array2string(dx, precision=2), "\n      logPost' prod", np.array2string(sp[c, 2], precision=8), "orc",
                      np.array2string(exp["prop_stats"][c, 2], precision=8), "\n      lq_fwd-lq_rev prod", np.array2string(lqp[c, 0] - lqp[c, 1], precision=6), "orc",
                      np.array2string(exp["diag"][c, 0] - exp["diag"][c, 1], precision=6), "\n      lq_fwd prod", np.array2string(lqp[c, 0], precision=6), "lq_rev prod",
                      np.array2string(lqp[c, 1], precision=6))
            outside = []
            for m in c:
                n_out = 0
                for k in range(idx.size):
                    q = exp["prop_vars"][m].copy()
                    pq = params[m].copy(); pq[idx] = q; pq[idx[k]] += h[k]
                    if not np.isfinite(orc.call_prior(star, pq)):
                        n_out += 1
                outside.append(n_out)
            # the finite-difference batch itself at these chains' proposals: device (C ABI) against the oracle, likelihood share
            Pq = params[c].copy()
            Pq[:, idx] = exp["prop_vars"][c]
            l0, gd = ctx.fd_gradient(star.model_id, Pq, star.plength, idx, h, T[c])
            gdev = []
            for j, m in enumerate(c):
                _, go, gpo = orc.fd_gradient_posterior(star, star.y, Pq[j], T[m], h)
                gl = go - gpo
                e = np.abs(gd[j] - gl) / np.maximum(np.abs(gl), 1e-3 * np.abs(gl).max())
                gdev.append((float(e.max()), int(e.argmax())))
            print("   FD batch at x' vs oracle (max rel dev, variable):", gdev)
            lfast, _, stf = ctx.loglike_params_batch(star.model_id, Pq, star.plength, T[c])
            from tamcmc_c_amd import sampler as S_
            import copy
            prh = []
            for j in range(len(c)):
                v, stp = S_.log_prior(star, Pq[j])
                prh.append(v)
            print("   at x': oracle logL", np.array2string(exp["prop_stats"][c, 0], precision=10), "\n          FD-batch L0", np.array2string(l0, precision=10),
                  "\n          FAST logL ", np.array2string(lfast, precision=10), "status", stf, "\n          oracle prior", np.array2string(exp["prop_stats"][c, 1], precision=10),
                  "\n          host prior  ", np.array2string(np.array(prh), precision=10))
            print(engine, "it", i, "A", ia, "swapped", exp["swapped"], "chains", c, "\n   Pmove prod", np.array2string(aft["Pmove"][c], precision=6), "\n   Pmove orc ",
                  np.array2string(exp["Pmove"][c], precision=6), "\n   moved", exp["moved"][c], "lq_fwd-lq_rev", np.array2string(exp["diag"][c, 0] - exp["diag"][c, 1], precision=4),
                  "|drift|", np.array2string(exp["diag"][c, 2], precision=3), "forward prior points outside the support at x':", outside, flush=True)
        s.close()
    ctx.close()


if __name__ == "__main__":
    for what in sys.argv[1:]:
        {"grad": grad_accuracy, "walk": walk, "ess": ess, "hostwalk": hostwalk}[what]()
