// priors_impl.h -- log-priors, written once for host (xreal = long double, like the reference) and device (double).
//   primitives           tamcmc/sources/stats_dictionary.cpp:38-250
//   apply_generic_priors tamcmc/sources/priors_calc.cpp:725-870
//   priors_MS_Global     tamcmc/sources/priors_calc.cpp:27-317   (model_index 9 = aj family, default = Classic)
//   priors_local         tamcmc/sources/priors_calc.cpp:514-629
// Where the reference exits (unsupported prior ids, model classes flagged "needs checks") *status is set to
// TAMCMC_ERR_BAD_MODEL and -inf is returned.
// The priors are a SUM of independent terms; `term range` arguments let the device evaluate the terms in parallel
// (one term per thread, tree-summed) while the host evaluates them all in the reference's order.
#pragma once
#include "mode_tables_impl.h"

namespace tamcmc {
namespace pr {

using mt::xreal;

TM_HD xreal neg_inf() { return -(xreal)INFINITY; }
TM_HD xreal xlog(xreal v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return log(v);
#else
    return logl(v);
#endif
}
TM_HD xreal xsqrt(xreal v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return sqrt(v);
#else
    return sqrtl(v);
#endif
}
TM_HD xreal xpow2(xreal v) {  // pow(v, 2.)
#if defined(__HIP_DEVICE_COMPILE__)
    return v * v;  // a correctly rounded pow(v,2) IS v*v; the device pow() costs ~150 fp64 issue slots
#else
    return powl(v, 2.);
#endif
}
TM_HD xreal xfabs(xreal v) { return v < 0 ? -v : v; }
#define TAMCMC_PIl ((xreal)3.141592653589793238462643383279502884L)

TM_HD xreal logP_uniform(xreal b_min, xreal b_max, xreal x) {
    if ((x <= b_max) && (x >= b_min)) return -xlog(xfabs(b_max - b_min));
    return neg_inf();
}
TM_HD xreal logP_uniform_abs(xreal b_min, xreal b_max, xreal x) {
    if ((xfabs(x) <= b_max) && (xfabs(x) >= b_min)) return -xlog(xfabs(b_max - b_min));
    return neg_inf();
}
TM_HD xreal logP_gaussian(xreal mean, xreal sigma, xreal x) {
    return -xlog(xsqrt(2 * TAMCMC_PIl) * sigma) - 0.5 * xpow2((x - mean) / sigma);
}
TM_HD xreal logP_jeffrey(xreal hmin, xreal hmax, xreal h) {
    if (h < hmax && h > 0) {
        const xreal prior = 1. / (h + hmin), norm = xlog((hmax + hmin) / hmin);
        return xlog(prior / norm);
    }
    return neg_inf();
}
TM_HD xreal logP_jeffrey_abs(xreal hmin, xreal hmax, xreal h) {
    if (xfabs(h) < hmax) {
        const xreal prior = 1. / (xfabs(h) + hmin), norm = xlog((hmax + hmin) / hmin);
        return xlog(prior / norm);
    }
    return neg_inf();
}
TM_HD xreal logP_uniform_gaussian(xreal b_min, xreal b_max, xreal sigma, xreal x) {
    xreal logP = 0;
    if (x < b_min) logP = neg_inf();
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * xpow2((x - b_max) / sigma);
    return logP - xlog(xfabs(b_max - b_min) + 0.5 * xsqrt(2 * TAMCMC_PIl) * sigma);
}
TM_HD xreal logP_gaussian_uniform(xreal b_min, xreal b_max, xreal sigma, xreal x) {
    xreal logP = 0;
    if (x > b_max) logP = neg_inf();
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x < b_min) logP = -0.5 * xpow2((x - b_min) / sigma);
    return logP - xlog(xfabs(b_max - b_min) + 0.5 * xsqrt(2 * TAMCMC_PIl) * sigma);
}
TM_HD xreal logP_gug(xreal b_min, xreal b_max, xreal s1, xreal s2, xreal x) {
    xreal logP = 0;
    if (x < b_min) logP = -0.5 * xpow2((x - b_min) / s1);
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * xpow2((x - b_max) / s2);
    return logP - xlog(xfabs(b_max - b_min) + 0.5 * xsqrt(2 * TAMCMC_PIl) * (s1 + s2));
}

// one term of apply_generic_priors: parameter i (pp = 4 x Np row-major)
TM_HD xreal generic_prior_term(const double *params, long Np, const double *pp, const int *sw, long i, int *status) {
    switch (sw[i]) {
    case 0: case 13: return 0;
    case 1: return logP_uniform(pp[i], pp[Np + i], params[i]);
    case 2: return logP_gaussian(pp[i], pp[Np + i], params[i]);
    case 4: return logP_jeffrey(pp[i], pp[Np + i], params[i]);
    case 5: return logP_uniform_gaussian(pp[i], pp[Np + i], pp[2 * Np + i], params[i]);
    case 6: return logP_gaussian_uniform(pp[i], pp[Np + i], pp[2 * Np + i], params[i]);
    case 7: return logP_gug(pp[i], pp[Np + i], pp[2 * Np + i], pp[3 * Np + i], params[i]);
    case 8: return logP_uniform_abs(pp[i], pp[Np + i], params[i]);
    case 10: return logP_jeffrey_abs(pp[i], pp[Np + i], params[i]);
    default:  // 3 multivariate (fatal in the reference), 9 flagged buggy, 11 unusable, 12 needs GSL tables
        if (status) *status = TAMCMC_ERR_BAD_MODEL;
        return neg_inf();
    }
}

// second difference i of y[0..n) with replicated edges: Scndder_adaptive_reggrid (derivatives_handler.cpp:400-426)
TM_HD double second_difference(const double *y, long n, long i) {
    if (n < 3) return 0.0;
    if (i == 0) return y[2] - 2. * y[1] + y[0];
    if (i == n - 1) return y[n - 1] - 2. * y[n - 2] + y[n - 3];
    return y[i + 1] - 2. * y[i] + y[i - 1];
}

// ---- priors_MS_Global split into: hard constraints (-inf or 0) and a list of additive terms ----
// hard constraints: visibilities >= 0, |aj/a1| limits (model_index 9), Harvey parameters >= 0.
// The checks are independent: check number t0, t0+stride, ... are evaluated (host: t0=0, stride=1 = all of them in the
// reference's order; device: one slice per lane, results OR-ed).  Returns 0 or -inf.
TM_HD xreal ms_global_constraints(const double *params, const int *pl, const int *sw, const double *extra, int *status,
                                  int t0 = 0, int stride = 1) {
    const double *ajova1_limit = &extra[2];
    const int impose_normHnlm = (int)extra[8];
    const int model_index = (int)extra[9];
    const int Nmax = pl[0], lmax = pl[1];
    const int Nfl[4] = {pl[2], pl[3], pl[4], pl[5]};
    const int Nsplit = pl[6], Nwidth = pl[7];
    const int Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
    // check index space: [0, lmax] visibilities, then for model_index 9 the (el, j, n) grid
    const int n_vis = lmax + 1;
    int n_aj = 0;
    if (model_index == 9)
        for (int el = 1; el < lmax + 1 && el < 4; el++) n_aj += 5 * Nfl[el];
#if defined(__HIP_DEVICE_COMPILE__)
    // the same checks dealt out by (degree, order) pair, the five a_j of a pair on one lane: no integer division to find a check's
    // indices, a1 formed once per pair (every check is independent: any one failing rejects)
    for (int t = t0; t < n_vis; t += stride)
        if (params[Nmax + t] < 0) return neg_inf();
    if (n_aj > 0) {
        const int npairs = n_aj / 5;
        for (int pr_ = t0; pr_ < npairs; pr_ += stride) {
            int n = pr_, el = 1, i0 = Nfl[0];
            while (el < 3 && n >= Nfl[el]) { n -= Nfl[el]; i0 += Nfl[el]; el++; }
            const double fl = params[Nmax + lmax + i0 + n];
            const double a1 = params[Nmax + lmax + Nf] + params[Nmax + lmax + Nf + 1] * (fl * 1e-3);
            bool bad = a1 < 0;
#pragma unroll 1
            for (int j = 1; j <= 5; j++) {
                const double aj = params[Nmax + lmax + Nf + 2 * j] + params[Nmax + lmax + Nf + 2 * j + 1] * (fl * 1e-3);
                bad = bad || (fabs(aj / a1) >= ajova1_limit[j]);
            }
            if (bad) return neg_inf();
        }
    }
    for (int t = n_vis + n_aj; t < n_vis + n_aj; t += stride) {
#else
    for (int t = t0; t < n_vis + n_aj; t += stride) {
#endif
        if (t < n_vis) {  // priors_calc.cpp:63-68
            if (params[Nmax + t] < 0) return neg_inf();
            continue;
        }
        int r = t - n_vis, el = 1, i0 = Nfl[0];  // priors_calc.cpp:206-228
        while (el < 3 && r >= 5 * Nfl[el]) { r -= 5 * Nfl[el]; i0 += Nfl[el]; el++; }
        const int j = 1 + r / Nfl[el], n = r % Nfl[el];
        const double fl = params[Nmax + lmax + i0 + n];
        const double a1 = params[Nmax + lmax + Nf] + params[Nmax + lmax + Nf + 1] * (fl * 1e-3);
        const double aj = params[Nmax + lmax + Nf + 2 * j] + params[Nmax + lmax + Nf + 2 * j + 1] * (fl * 1e-3);
        if (fabs(aj / a1) >= ajova1_limit[j]) return neg_inf();
        if (a1 < 0) return neg_inf();
    }
    switch (model_index) {
    case 9: break;
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8:
        if (status) *status = TAMCMC_ERR_BAD_MODEL;  // families without a table builder in this build
        return neg_inf();
    default:  // Classic models (priors_calc.cpp:230-262)
        if (impose_normHnlm != 0) {
            if (status) *status = TAMCMC_ERR_BAD_MODEL;
            return neg_inf();
        }
        break;
    }
    if (t0 != 0) return 0;  // the three noise checks belong to slice 0
    const int on = Nmax + lmax + Nf + Nsplit + Nwidth;  // noise block
    if (sw[on + 3] != 0)
        if ((params[on + 3] < 0) || (params[on + 4] < 0) || (params[on + 5] < 0)) return neg_inf();
    if (sw[on + 6] != 0)
        if ((params[on + 6] < 0) || (params[on + 7] < 0) || (params[on + 8] < 0)) return neg_inf();
    if ((sw[Nmax + lmax + Nf + 9] != 0) && (params[on + 9] < 0)) return neg_inf();  // index as in priors_calc.cpp:272
    return 0;
}

// number of additive terms after the Np generic ones: d02 terms then smoothness terms
TM_HD int ms_global_extra_terms(const int *pl, const double *extra) {
    const int Nfl[4] = {pl[2], pl[3], pl[4], pl[5]};
    int n = 0;
    if (Nfl[0] == Nfl[2]) n += Nfl[0];
    if ((int)extra[0] == 1) {
        for (int el = 0; el < pl[1] + 1 && el < 4; el++) n += Nfl[el];
    }
    return n;
}
// extra term t (0-based) -- Dnu = slope of the l=0 frequencies (priors_calc.cpp:277-313)
TM_HD xreal ms_global_extra_term(const double *params, const int *pl, const double *extra, double Dnu, int t) {
    const int Nmax = pl[0], lmax = pl[1];
    const int Nfl[4] = {pl[2], pl[3], pl[4], pl[5]};
    if (Nfl[0] == Nfl[2]) {
        if (t < Nfl[0]) {
            const double d02 = params[Nmax + lmax + t] - params[Nmax + lmax + Nfl[0] + Nfl[1] + t];
            return logP_gaussian_uniform(0, Dnu / 3., 0.015 * Dnu, d02);
        }
        t -= Nfl[0];
    }
    const double scoef = extra[1];
    int i0 = 0;
    for (int el = 0; el < lmax + 1 && el < 4; el++) {
        if (t < Nfl[el]) return logP_gaussian(0, scoef, second_difference(params + Nmax + lmax + i0, Nfl[el], t));
        t -= Nfl[el];
        i0 += Nfl[el];
    }
    return 0;
}

TM_HD xreal local_constraints(const double *params, const int *pl, const int *sw, const double *extra) {
    const double a3ova1_limit = extra[2];
    const int Nmax = pl[0], Nvis = pl[1];
    const int Nf = pl[2] + pl[3] + pl[4] + pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8];
    const int o = Nmax + Nvis + Nf;
    if (params[o] != 0) {  // an a1 is fitted directly (priors_calc.cpp:541-546)
        if (fabs(params[o + 2] / params[o]) >= a3ova1_limit) return neg_inf();
    } else if ((params[o + 3] != 0) && (params[o + 4] != 0)) {  // sqrt(a1) cos i, sqrt(a1) sin i (:547-554)
        if (fabs(params[o + 2] / (mt::pow_2(params[o + 3]) + mt::pow_2(params[o + 4]))) >= a3ova1_limit) return neg_inf();
    }
    const int oi = o + Nsplit + Nwidth + Nnoise;  // inclination slot (:561-564)
    if ((sw[oi] != 0) && (params[oi] < 0)) return neg_inf();
    return 0;
}

// ---- serial evaluation in the reference's order (host; also usable on the device by one thread) ----
// generic_terms (class 4): the Np generic prior terms already evaluated (the device spreads them over lanes; summed here in the same order)
TM_HD xreal prior_serial(int prior_class, const double *params, const int *pl, long Np, const double *pp, const int *sw,
                         const double *extra, int *status, const xreal *generic_terms = nullptr) {
    xreal f = 0;
    if (prior_class == 2) {
        const xreal c = ms_global_constraints(params, pl, sw, extra, status);
        if (c != 0) return c;
        for (long i = 0; i < Np; i++) f = f + generic_prior_term(params, Np, pp, sw, i, status);
        double fit[2];
        mt::linfit_index(params + pl[0] + pl[1], pl[2], fit);
        const int ne = ms_global_extra_terms(pl, extra);
        for (int t = 0; t < ne; t++) f = f + ms_global_extra_term(params, pl, extra, fit[0], t);
        return f;
    }
    if (prior_class == 3) {
        const xreal c = local_constraints(params, pl, sw, extra);
        if (c != 0) return c;
        for (long i = 0; i < Np; i++) f = f + generic_prior_term(params, Np, pp, sw, i, status);
        return f;
    }
    if (prior_class == 4) {  // io_asymptotic: priors_asymptotic, priors_calc.cpp:319-512 (host engine only: the RGB model's pre-step)
        const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5], Nsplit = pl[6], Nwidth = pl[7];
        const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
        const double scoef = extra[1], a3ova1_limit = extra[2];
        const int model_switch = (int)extra[4];
        const int i0 = Nmax + lmax + Nf + Nsplit, on = i0 + Nwidth;
        const double a3 = params[Nmax + lmax + Nf + 4], rot_env = fabs(params[Nmax + lmax + Nf]);
        for (int i = Nmax; i <= Nmax + lmax; i++)  // (one past the visibilities, as the reference's loop bound)
            if (params[i] < 0) return neg_inf();
        if (fabs(a3 / rot_env) >= a3ova1_limit) return neg_inf();
        if (sw[on + 3] != 0 && ((params[on + 3] < 0) || (params[on + 4] < 0) || (params[on + 5] < 0))) return neg_inf();
        if (sw[on + 6] != 0 && ((params[on + 6] < 0) || (params[on + 7] < 0) || (params[on + 8] < 0))) return neg_inf();
        if ((sw[Nmax + lmax + Nf + 9] != 0) && (params[on + 9] < 0)) return neg_inf();  // index as in priors_calc.cpp:391
        for (int i = i0; i < i0 + Nwidth; i++)
            if (sw[i] == 2 && params[i] < 0) return neg_inf();  // Gaussian priors on the width law: positive support only
        for (long i = 0; i < Np; i++) f = f + (generic_terms ? generic_terms[i] : generic_prior_term(params, Np, pp, sw, i, status));
        if (model_switch == 1) {  // the v3 models (l=1 p-mode list in the parameter vector): not built here
            if (status) *status = TAMCMC_ERR_BAD_MODEL;
            return neg_inf();
        }
        if (model_switch == 3) {
            if (params[Nmax + lmax + Nfl0 + 6] < 0) return neg_inf();  // Wfactor
            if (params[Nmax + lmax + Nfl0 + 7] < 0) return neg_inf();  // Hfactor
        }
        // smoothness of the l=0 and l=3 ladders.  The reference switches it on the prior id of parameter i0+Nwidth -- the loop
        // variable left over from the width check above (priors_calc.cpp:466; most likely meant extra_priors[0]) -- kept as is.
        if (sw[on] == 1) {
            for (int i = 0; i < Nfl0; i++) f = f + logP_gaussian(0, scoef, second_difference(params + Nmax + lmax, Nfl0, i));
            for (int i = 0; i < Nfl3; i++)
                f = f + logP_gaussian(0, scoef, second_difference(params + Nmax + lmax + Nfl0 + Nfl1 + Nfl2, Nfl3, i));
        }
        return f;
    }
    if (status) *status = TAMCMC_ERR_BAD_MODEL;
    return neg_inf();
}

}  // namespace pr
}  // namespace tamcmc
