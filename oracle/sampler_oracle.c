/*
 * sampler_oracle.c -- CPU ORACLE (test infrastructure, NOT product code), part 3: one iteration of the reference's sampler,
 * restated from tamcmc/sources/MALA.cpp with the random numbers passed in EXPLICITLY (the reference draws them from libc rand()
 * seeded by time(NULL), MALA.cpp:62-63, random_JB.cpp:255: unseedable; the product replaces them by counter-based Philox streams
 * and exposes its draws, so the same (z, u) can be fed to both sides):
 *
 *   p1_fct / p2_fct / p3_fct                MALA.cpp:135-176   clips of the Robbins-Monro updates
 *   update_proposal                         MALA.cpp:296-319   mu, covarmat, sigma with gain gamma = c0/(1+i)
 *   new_prop_values                         MALA.cpp:339-369   x' = x + chol((covarmat + epsilon2 I) sigma) z
 *   update_position_MH (accept rule)        MALA.cpp:490-551   incl. the NaN-likelihood and -inf-posterior cases
 *   parallel_tempering                      MALA.cpp:397-461   adjacent pair, tempered likelihoods; the line-444 quirk selectable
 *   the loop body of MALA::execute          MALA.cpp:645-703   propose / generate_model / accept per chain, learning test, swap
 *   Model_def::generate_model               model_def.cpp:466-482  (on top of orc_call_prior / orc_call_model / orc_call_likelihood)
 *   the Langevin step (use_drift = 1)       no reference counterpart (stubs at MALA.cpp:321-337, fatal at :496-500): second half of this file
 *
 * Pinning status: the reference holds no stored numbers for a sampler step (its RNG cannot be seeded); these functions are a
 * statement-by-statement reading of the cited lines and are checked by hand-computed known answers (tests/test_oracle_sampler.py).
 * The Cholesky factor is Eigen's LLT in the reference (blocked, version-dependent rounding): any correct factor agrees to rounding,
 * so positions agree to ~1e-15 relative, not to the bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tamcmc_oracle.h"

#define ORC_PI_L 3.141592653589793238462643383279502884L

/* ---- MALA.cpp:135-151 ---- */
long double orc_p1_fct(long double x, long double epsilon1, long double A1) {
    long double scalar_p = x; /* (the reference leaves it unset for a NaN input) */
    if (x >= epsilon1 && x <= A1) scalar_p = x;
    if (x < epsilon1) scalar_p = epsilon1;
    if (x > A1) scalar_p = A1;
    return scalar_p;
}

/* Eigen's .norm(): Frobenius / Euclidean norm in the scalar type (double) */
static double norm2(const double *v, long n) {
    double s = 0;
    for (long i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

/* ---- MALA.cpp:153-164 ---- */
void orc_p2_fct(double *M, long n, double A1) {
    const double nrm = norm2(M, n * n);
    if (!(nrm <= A1))
        for (long i = 0; i < n * n; i++) M[i] = M[i] * A1 / nrm;
}

/* ---- MALA.cpp:166-176 ---- */
void orc_p3_fct(double *v, long n, double A1) {
    const double nrm = norm2(v, n);
    if (!(nrm <= A1))
        for (long i = 0; i < n; i++) v[i] = v[i] * A1 / nrm;
}

/* ---- MALA.cpp:296-319: Robbins-Monro update of chain m's proposal law (mu: Nvars, covarmat: Nvars x Nvars row-major) ---- */
void orc_update_proposal(double *mu, double *covarmat, double *sigma, const double *vars, long Nvars, long double acceptance,
                         long double gamma, long double target_acceptance, long double epsilon1, long double A1) {
    const double g = (double)gamma; /* Eigen expressions are evaluated in the vectors' scalar type */
    /* var_p3 = mu + gamma (vars - mu); mu = p3(var_p3)                                             :307-308 */
    for (long k = 0; k < Nvars; k++) mu[k] = mu[k] + g * (vars[k] - mu[k]);
    orc_p3_fct(mu, Nvars, (double)A1);
    /* mat = |vars - mu><vars - mu| with the UPDATED mu; var_p2 = covarmat + gamma (mat - covarmat)   :311-313 */
    double *d = (double *)malloc((size_t)Nvars * sizeof(double));
    for (long k = 0; k < Nvars; k++) d[k] = vars[k] - mu[k];
    for (long i = 0; i < Nvars; i++)
        for (long j = 0; j < Nvars; j++) {
            const double c = covarmat[i * Nvars + j];
            covarmat[i * Nvars + j] = c + g * (d[i] * d[j] - c);
        }
    free(d);
    orc_p2_fct(covarmat, Nvars, (double)A1);
    /* var_p1 = sigma + gamma (acceptance - target); sigma = p1(var_p1)   (long double scalars)      :316-317 */
    const long double var_p1 = (long double)*sigma + gamma * (acceptance - target_acceptance);
    *sigma = (double)orc_p1_fct(var_p1, epsilon1, A1);
}

/* lower Cholesky factor of the symmetric matrix A (row-major, in place; upper part zeroed); 0 = ok, 1 = not positive definite */
static int llt_lower(double *A, long n) {
    for (long j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (long k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return 1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (long i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (long k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
        for (long k = j + 1; k < n; k++) A[j * n + k] = 0.0;
    }
    return 0;
}

/* ---- MALA.cpp:339-369: tmpmat = (covarmat + epsilon2) sigma (epsilon2 = epsi2 on the diagonal, :82-83); Lchol = llt(tmpmat).L;
 *      ran = vars + Lchol y.  y = z (the caller's normals).  The redraw on a non-finite result (:356-366) is left to the caller:
 *      the return value says whether `out` is finite.  L_out (may be NULL) receives the factor. */
int orc_new_prop_values(const double *covarmat, double sigma, double epsi2, const double *vars, const double *z, long Nvars, double *out,
                        double *L_out) {
    double *T = (double *)malloc((size_t)(Nvars * Nvars) * sizeof(double));
    for (long i = 0; i < Nvars; i++)
        for (long j = 0; j < Nvars; j++) T[i * Nvars + j] = (covarmat[i * Nvars + j] + (i == j ? epsi2 : 0.0)) * sigma;
    const int bad = llt_lower(T, Nvars);
    int finite = !bad;
    for (long i = 0; i < Nvars; i++) {
        double s = 0;
        for (long k = 0; k <= i; k++) s += T[i * Nvars + k] * z[k];
        out[i] = vars[i] + s;
        if (!isfinite(out[i])) finite = 0;
    }
    if (L_out) memcpy(L_out, T, (size_t)(Nvars * Nvars) * sizeof(double));
    free(T);
    return finite ? 0 : 1;
}

/* ---- MALA.cpp:490-551, use_drift = 0.  Returns 1 = moved, 0 = stays, -1 = the reference would stop ("rejectionrate_isnan", :519-521).
 *      *r_out = the move probability stored in Pmove (:549). ---- */
int orc_mh_accept(double logL_prop, double logPost_prop, double logPost_cur, double u, double *r_out) {
    long double r;
    if (isnan(logL_prop) == 0) {                                   /* :490 */
        if (logPost_prop == -INFINITY) r = 0.;                      /* :491-493 */
        else {
            const double e = exp(logPost_prop - logPost_cur + 0. - 0.);  /* :515, logproba_cur = logproba_prop = 0 */
            if (isnan(e)) { *r_out = NAN; return -1; }              /* minCoeff of (1, NaN) -> NaN -> fatal :519-521 */
            r = e < 1. ? e : 1.;                                    /* ri.minCoeff() :516 */
        }
    } else r = 0.;                                                  /* :522-524 */
    *r_out = (double)r;
    return (u <= r) ? 1 : 0;                                        /* :536 */
}

/* ---- MALA.cpp:397-461.  ind_A and the comparator u are the caller's draws (:400, :406).  Rows of vars / params, the tempered
 *      logLikelihood, logPrior, logPosterior, moved, Pmove of the pair are exchanged exactly as written there.
 *      literal_444 != 0: logPosterior[B] = logL_A_TB + logPrior[A] AFTER logPrior[A] has been overwritten by B's (:433, :444), i.e. with
 *      B's old prior -- what the reference executes.  literal_444 == 0: logL_A_TB + A's prior (the value :444 evidently means).
 *      Returns 1 when swapped; *Pswap = r_T then, 0 otherwise (:417-418, :450). ---- */
int orc_parallel_tempering(double *logL, double *logPrior, double *logPost, double *vars, double *params, int *moved, double *Pmove,
                           const double *Tcoefs, long Nvars, long Nparams, int ind_A, double u, int literal_444, double *Pswap) {
    const int ind_B = ind_A + 1;
    const long double logL_A_TB = (long double)logL[ind_A] * Tcoefs[ind_A] / Tcoefs[ind_B];   /* :409 */
    const long double logL_B_TA = (long double)logL[ind_B] * Tcoefs[ind_B] / Tcoefs[ind_A];   /* :410 */
    const double e = exp((double)(logL_A_TB + logL_B_TA - logL[ind_A] - logL[ind_B]));        /* :412 (VectorXd entry: double) */
    const long double r_T = e < 1. ? e : 1.;                                                   /* :413 */
    *Pswap = 0;
    if (!(u <= r_T)) return 0;
    double *tv = (double *)malloc((size_t)Nvars * sizeof(double)), *tp = (double *)malloc((size_t)Nparams * sizeof(double));
    memcpy(tp, params + (size_t)ind_A * Nparams, (size_t)Nparams * sizeof(double));           /* save A :421-428 */
    memcpy(tv, vars + (size_t)ind_A * Nvars, (size_t)Nvars * sizeof(double));
    const double logPr_A = logPrior[ind_A];
    const int move_A = moved[ind_A];
    const double Pmove_A = Pmove[ind_A];
    memcpy(params + (size_t)ind_A * Nparams, params + (size_t)ind_B * Nparams, (size_t)Nparams * sizeof(double));  /* A <- B :431-438 */
    memcpy(vars + (size_t)ind_A * Nvars, vars + (size_t)ind_B * Nvars, (size_t)Nvars * sizeof(double));
    logL[ind_A] = (double)logL_B_TA;
    logPrior[ind_A] = logPrior[ind_B];
    logPost[ind_A] = (double)(logL_B_TA + logPrior[ind_B]);
    moved[ind_A] = moved[ind_B];
    Pmove[ind_A] = Pmove[ind_B];
    memcpy(params + (size_t)ind_B * Nparams, tp, (size_t)Nparams * sizeof(double));            /* B <- A :441-448 */
    memcpy(vars + (size_t)ind_B * Nvars, tv, (size_t)Nvars * sizeof(double));
    logL[ind_B] = (double)logL_A_TB;
    logPrior[ind_B] = logPr_A;
    logPost[ind_B] = (double)(logL_A_TB + (literal_444 ? logPrior[ind_A] /* already B's old prior */ : logPr_A));
    moved[ind_B] = move_A;
    Pmove[ind_B] = Pmove_A;
    free(tv); free(tp);
    *Pswap = (double)r_T;
    return 1;
}

/* MALA.cpp:656-667: is the proposal law updated after the MH test of iteration i? */
int orc_learn_at(long i, const long *Nt_learn, const long *periods_learn, long n_periods) {
    int logic = 0;
    long which = 0;
    for (long l = 0; l < n_periods; l++) {
        logic = logic || ((i >= Nt_learn[l]) && (i < Nt_learn[l + 1]));
        if ((i >= Nt_learn[l]) && (i < Nt_learn[l + 1])) which = l;
    }
    return (logic == 1 && (i % periods_learn[which]) == 0) ? 1 : 0;
}

/* Model_def::generate_model (model_def.cpp:466-482) for one parameter vector at temperature Tcoef.
 * A table failure (the reference exits: build_lorentzian.cpp:650-665) is reported as a NaN likelihood, which the accept rule rejects. */
void orc_generate_model(const orc_sampler_star *S, const double *params, double Tcoef, double init_logL, double *model_scratch, double *logL,
                        double *logPr, double *logPost) {
    *logPr = orc_call_prior(S->prior_class, params, S->plength, S->priors, S->priors_switch, S->extra_priors);   /* :471 */
    if (*logPr != -INFINITY) {                                                                                      /* :472 */
        const int st = orc_call_model(S->model_id, params, S->plength, S->x, S->Nx, model_scratch);                 /* :473 */
        *logL = (st == ORC_OK) ? orc_call_likelihood(S->y, model_scratch, S->Nx, S->likelihood_params, Tcoef) : NAN; /* :474 */
        *logPost = *logL + *logPr;                                                                                  /* :475 */
    } else {
        *logL = init_logL;                                                                                          /* :478 */
        *logPost = -INFINITY;                                                                                       /* :479 */
    }
}

/* One pass of the loop body of MALA::execute (MALA.cpp:645-703) for iteration i: for every chain propose (:481), scatter the variables
 * into the parameter vector (:486-487, model_def.cpp:484-492), generate_model (:488), accept rule (:490-551), learning test and
 * update_proposal (:656-667); then the parallel-tempering step (:688-699) when do_swap != 0.
 * State arrays are updated in place.  z: [Nchains x Nvars], u_mh: [Nchains].  Optional outputs (may be NULL): prop_vars [Nchains x Nvars],
 * prop_stats [Nchains x 3] = the proposals' logL, logPrior, logPosterior.  Returns 0, or -1 when the reference would have stopped. */
int orc_sampler_iteration(const orc_sampler_star *S, long i, int learn, int do_swap, int ind_A, double u_swap, int literal_444, const double *z,
                          const double *u_mh, double *params, double *vars, double *logL, double *logPrior, double *logPost, int *moved,
                          double *Pmove, double *mu, double *covarmat, double *sigma, int *swapped, double *prop_vars, double *prop_stats) {
    const long C = S->Nchains, Nv = S->Nvars, Np = S->Nparams;
    const long double gamma = (long double)S->c0 / (1. + i);                       /* :645 */
    int fatal = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (long m = 0; m < C; m++) {                                                  /* :648 */
        double *v_new = (double *)malloc((size_t)Nv * sizeof(double)), *p_new = (double *)malloc((size_t)Np * sizeof(double));
        double *model = (double *)malloc((size_t)S->Nx * sizeof(double));
        orc_new_prop_values(covarmat + (size_t)m * Nv * Nv, sigma[m], S->epsi2, vars + (size_t)m * Nv, z + (size_t)m * Nv, Nv, v_new, NULL);
        memcpy(p_new, params + (size_t)m * Np, (size_t)Np * sizeof(double));
        for (long k = 0; k < Nv; k++) p_new[S->index_to_relax[k]] = v_new[k];       /* update_params_with_vars */
        double l, pr, po, r;
        orc_generate_model(S, p_new, S->Tcoefs[m], S->init_logL[m], model, &l, &pr, &po);
        if (prop_vars) memcpy(prop_vars + (size_t)m * Nv, v_new, (size_t)Nv * sizeof(double));
        if (prop_stats) { prop_stats[3 * m] = l; prop_stats[3 * m + 1] = pr; prop_stats[3 * m + 2] = po; }
        const int acc = orc_mh_accept(l, po, logPost[m], u_mh[m], &r);
        if (acc < 0) {
#pragma omp atomic write
            fatal = 1;
        }
        if (acc == 1) {                                                             /* :536-543 */
            memcpy(params + (size_t)m * Np, p_new, (size_t)Np * sizeof(double));
            memcpy(vars + (size_t)m * Nv, v_new, (size_t)Nv * sizeof(double));
            logL[m] = l; logPrior[m] = pr; logPost[m] = po;
            moved[m] = 1;
        } else moved[m] = 0;                                                        /* :545 */
        Pmove[m] = r;                                                               /* :549 */
        if (learn)                                                                  /* :664-667 */
            orc_update_proposal(mu + (size_t)m * Nv, covarmat + (size_t)m * Nv * Nv, sigma + m, vars + (size_t)m * Nv, Nv, Pmove[m], gamma,
                                S->target_acceptance, S->epsilon1, S->A1);
        free(v_new); free(p_new); free(model);
    }
    *swapped = 0;
    if (do_swap) {                                                                  /* :688-689: i % dN_mixing == 0 && i != 0 */
        double Pswap;
        *swapped = orc_parallel_tempering(logL, logPrior, logPost, vars, params, moved, Pmove, S->Tcoefs, Nv, Np, ind_A, u_swap, literal_444, &Pswap);
    }
    return fatal ? -1 : 0;
}

/* =====================================================================================================================================
 * Langevin step (use_drift = 1).  NO REFERENCE COUNTERPART: MALA::D_MALA returns zeros (MALA.cpp:321-328), multinormal_logpdf returns 0
 * (:330-337) and use_drift = 1 is fatal (:496-500); the acceptance ratio they were meant to feed is written at :515 as
 * exp(logPost' - logPost + logproba_prop - logproba_cur).  What follows restates the step the north star names -- finite-difference
 * gradient of the reference log-posterior -> preconditioned drift -> Gaussian proposal -> Metropolis-Hastings ratio with BOTH proposal
 * densities -- from its textbook definition (Roberts & Tweedie 1996; truncated drift: Atchade 2006, the paper MALA.cpp:18 cites),
 * independently of the product: the densities are full multivariate-normal log-densities evaluated by Gaussian elimination with pivoting
 * in long double (no Cholesky factor, no dropped normalisation), the gradient is recomputed from the position at every use (the product
 * carries it along and re-tempers it on swaps), and the likelihood differences are taken term by term in long double.
 *     M_m        = (covarmat_m + epsilon2 I) sigma_m                      the proposal covariance of MALA.cpp:348
 *     grad(x)    = forward differences (steps h) of  logL(x)/T_m + logPrior(x)
 *     drift_m(x) = (1/2) M_m grad(x) min(1, delta/|grad(x)|)              (no truncation for delta <= 0)
 *     x'         = x + drift_m(x) + chol(M_m) z
 *     r          = min(1, exp(logPost(x') - logPost(x) + log N(x; x' + drift_m(x'), M_m) - log N(x'; x + drift_m(x), M_m)))
 * Parity status: unpinned by the reference (it holds no such step); pinned by closed-form known answers in tests/test_oracle_sampler.py
 * (a Gaussian target sampled exactly by the unadjusted step, detailed balance of the kernel on a grid).
 * ===================================================================================================================================== */

/* log N(v; mean, M) of an n-variate normal, M symmetric positive definite (row-major).  Gaussian elimination with partial pivoting on
 * [M | v - mean] in long double: quadratic form from the solution, log det from the pivots.  Returns NaN for a singular matrix. */
long double orc_mvn_logpdf(const double *v, const double *mean, const double *M, long n) {
    long double *A = (long double *)malloc((size_t)(n * (n + 1)) * sizeof(long double));
    long double *d = (long double *)malloc((size_t)n * sizeof(long double));
    for (long i = 0; i < n; i++) {
        for (long j = 0; j < n; j++) A[i * (n + 1) + j] = M[i * n + j];
        d[i] = (long double)v[i] - (long double)mean[i];
        A[i * (n + 1) + n] = d[i];
    }
    long double logdet = 0;
    int bad = 0;
    for (long k = 0; k < n && !bad; k++) {
        long piv = k;
        for (long i = k + 1; i < n; i++)
            if (fabsl(A[i * (n + 1) + k]) > fabsl(A[piv * (n + 1) + k])) piv = i;
        if (piv != k)
            for (long j = 0; j <= n; j++) { const long double t = A[k * (n + 1) + j]; A[k * (n + 1) + j] = A[piv * (n + 1) + j]; A[piv * (n + 1) + j] = t; }
        const long double p = A[k * (n + 1) + k];
        if (!(fabsl(p) > 0)) { bad = 1; break; }
        logdet += logl(fabsl(p));   /* |det| = prod |pivots|; M is positive definite, so det > 0 */
        for (long i = k + 1; i < n; i++) {
            const long double f = A[i * (n + 1) + k] / p;
            for (long j = k; j <= n; j++) A[i * (n + 1) + j] -= f * A[k * (n + 1) + j];
        }
    }
    long double q = 0;
    if (!bad) {
        for (long i = n - 1; i >= 0; i--) {   /* back substitution: A[i][n] <- (M^-1 d)[i] */
            long double s = A[i * (n + 1) + n];
            for (long j = i + 1; j < n; j++) s -= A[i * (n + 1) + j] * A[j * (n + 1) + n];
            A[i * (n + 1) + n] = s / A[i * (n + 1) + i];
        }
        for (long i = 0; i < n; i++) q += d[i] * A[i * (n + 1) + n];
    }
    free(A); free(d);
    if (bad) return NAN;
    return -0.5L * q - 0.5L * logdet - 0.5L * (long double)n * logl(2 * ORC_PI_L);
}

/* Forward-difference gradient of the tempered log-posterior at `params` (steps h[k] on variable k = parameter index_to_relax[k]):
 *   likelihood share  (logL(theta + h e_k) - logL(theta)) / h_applied / T with logL = -p sum_i (y_i / M_i + ln M_i) (likelihoods.cpp:17-28),
 *                     the difference of the two sums taken bin by bin in long double (a difference of two rounded ~Nx-term sums would
 *                     carry ~1e-16 Nx |logL| / h of noise);
 *   prior share       forward difference of call_prior; when theta + h e_k leaves the prior's support, the backward difference; when both
 *                     neighbours do, zero (the convention of the product's gradient, stated in include/tamcmc_sampler.h);
 *   a component that is not finite (table failure at the perturbed point) is zero.
 * grad = both shares, gradP = the prior's share alone.  Returns the model status of the base point. */
int orc_fd_gradient_posterior(const orc_sampler_star *S, const double *params, double Tcoef, const double *h, double *grad, double *gradP) {
    const long Nv = S->Nvars, Np = S->Nparams, Nx = S->Nx;
    const long p = (long)S->likelihood_params;
    double *M0 = (double *)malloc((size_t)Nx * sizeof(double));
    const int st0 = orc_call_model(S->model_id, params, S->plength, S->x, Nx, M0);
    const double pr0 = orc_call_prior(S->prior_class, params, S->plength, S->priors, S->priors_switch, S->extra_priors);
#pragma omp parallel for schedule(dynamic, 1)
    for (long k = 0; k < Nv; k++) {
        double *q = (double *)malloc((size_t)Np * sizeof(double)), *M1 = (double *)malloc((size_t)Nx * sizeof(double));
        memcpy(q, params, (size_t)Np * sizeof(double));
        const long j = S->index_to_relax[k];
        volatile double xp = params[j] + h[k];
        const double happ = xp - params[j];
        q[j] = xp;
        const int st1 = orc_call_model(S->model_id, q, S->plength, S->x, Nx, M1);
        double gl = NAN;
        if (st0 == ORC_OK && st1 == ORC_OK) {
            long double acc = 0;
            for (long i = 0; i < Nx; i++) {
                const long double m0 = M0[i], m1 = M1[i], dm = m1 - m0;
                acc += (long double)S->y[i] * (-dm) / (m0 * m1) + log1pl(dm / m0);   /* y/M1 - y/M0 + ln M1 - ln M0 */
            }
            gl = (double)((-(long double)p * acc) / Tcoef / happ);
        }
        if (!isfinite(gl)) gl = 0.0;
        const double prp = orc_call_prior(S->prior_class, q, S->plength, S->priors, S->priors_switch, S->extra_priors);
        double gp;
        if (isfinite(prp)) gp = (prp - pr0) / happ;
        else {
            q[j] = params[j] - h[k];
            const double prm = orc_call_prior(S->prior_class, q, S->plength, S->priors, S->priors_switch, S->extra_priors);
            gp = isfinite(prm) ? (pr0 - prm) / happ : 0.0;
        }
        const double g = gl + gp;
        grad[k] = isfinite(g) ? g : 0.0;
        gradP[k] = isfinite(g) ? gp : 0.0;
        free(q); free(M1);
    }
    free(M0);
    return st0;
}

/* drift = (1/2) sigma (covarmat + epsilon2 I) grad min(1, delta / |grad|)   (what D_MALA, MALA.cpp:321-328, was to return) */
void orc_langevin_drift(const double *covarmat, double sigma, double epsi2, double delta, const double *grad, long Nvars, double *drift) {
    long double n2 = 0;
    for (long k = 0; k < Nvars; k++) n2 += (long double)grad[k] * grad[k];
    const long double nrm = sqrtl(n2);
    long double scale = 1;
    if (delta > 0 && nrm > delta) scale = delta / nrm;
    for (long i = 0; i < Nvars; i++) {
        long double acc = 0;
        for (long j = 0; j < Nvars; j++) acc += ((long double)covarmat[i * Nvars + j] + (i == j ? (long double)epsi2 : 0.0L)) * grad[j];
        drift[i] = isfinite((double)nrm) ? (double)(0.5L * sigma * scale * acc) : 0.0;
    }
}

/* One pass of the sampler's loop body (the structure of MALA.cpp:645-703) with the Langevin proposal.  Arguments as orc_sampler_iteration, plus:
 * fd_step_rel (steps h_k = fd_step_rel max(|mu_0k|, 1e-3) from chain 0's running mean BEFORE this iteration's adaptation -- the product's
 * rule, include/tamcmc_sampler.h), delta (drift truncation), diag (may be NULL) [Nchains x 4] = log q(x'|x), log q(x|x'), |drift(x)|, |drift(x')|,
 * chain_mask (may be NULL) [Nchains]: only the flagged chains are advanced (a check of a few chains of a large ladder; the caller flags
 * the swap pair, and compares the flagged chains only), prop_given (may be NULL) [Nchains x Nvars]: the test is made AT these proposals
 * instead of the oracle's own (which are still returned in prop_vars).  Why: the model is truncated to windows of whole bins
 * (build_lorentzian.cpp:595-676), so the log-likelihood -- hence its forward-difference gradient -- jumps where a window edge crosses a
 * bin; when the step h_k moves an edge across a bin boundary (probability ~ h_k / bin width per edge) that gradient component carries
 * (one bin's likelihood term) / h_k.  Two proposals that agree to rounding can sit on either side of such a jump and then have different
 * drifts; handing the oracle the product's proposal (checked against its own to rounding) keeps the comparison on one side. */
int orc_langevin_iteration(const orc_sampler_star *S, long i, int learn, int do_swap, int ind_A, double u_swap, int literal_444, const double *z,
                           const double *u_mh, double fd_step_rel, double delta, double *params, double *vars, double *logL, double *logPrior,
                           double *logPost, int *moved, double *Pmove, double *mu, double *covarmat, double *sigma, int *swapped,
                           double *prop_vars, double *prop_stats, double *diag, const int *chain_mask, const double *prop_given) {
    const long C = S->Nchains, Nv = S->Nvars, Np = S->Nparams;
    const long double gamma = (long double)S->c0 / (1. + i);
    double *h = (double *)malloc((size_t)Nv * sizeof(double));
    for (long k = 0; k < Nv; k++) h[k] = fd_step_rel * fmax(fabs(mu[k]), 1e-3);
    int fatal = 0;
    for (long m = 0; m < C; m++) {   /* serial over chains: the gradient is parallel over variables */
        if (chain_mask && !chain_mask[m]) continue;
        double *g = (double *)malloc((size_t)Nv * sizeof(double)), *gp = (double *)malloc((size_t)Nv * sizeof(double));
        double *d0 = (double *)malloc((size_t)Nv * sizeof(double)), *d1 = (double *)malloc((size_t)Nv * sizeof(double));
        double *mean = (double *)malloc((size_t)Nv * sizeof(double)), *v_new = (double *)malloc((size_t)Nv * sizeof(double));
        double *p_new = (double *)malloc((size_t)Np * sizeof(double)), *model = (double *)malloc((size_t)S->Nx * sizeof(double));
        double *Mm = (double *)malloc((size_t)(Nv * Nv) * sizeof(double));
        const double *cov = covarmat + (size_t)m * Nv * Nv, *x = vars + (size_t)m * Nv;
        for (long a = 0; a < Nv; a++)
            for (long b = 0; b < Nv; b++) Mm[a * Nv + b] = (cov[a * Nv + b] + (a == b ? S->epsi2 : 0.0)) * sigma[m];
        /* drift at the current position */
        orc_fd_gradient_posterior(S, params + (size_t)m * Np, S->Tcoefs[m], h, g, gp);
        orc_langevin_drift(cov, sigma[m], S->epsi2, delta, g, Nv, d0);
        /* x' = x + drift + L z */
        for (long k = 0; k < Nv; k++) mean[k] = x[k] + d0[k];
        orc_new_prop_values(cov, sigma[m], S->epsi2, mean, z + (size_t)m * Nv, Nv, v_new, NULL);
        if (prop_vars) memcpy(prop_vars + (size_t)m * Nv, v_new, (size_t)Nv * sizeof(double));   /* the oracle's own proposal, always */
        if (prop_given) memcpy(v_new, prop_given + (size_t)m * Nv, (size_t)Nv * sizeof(double)); /* ... tested at the caller's (see the header) */
        memcpy(p_new, params + (size_t)m * Np, (size_t)Np * sizeof(double));
        for (long k = 0; k < Nv; k++) p_new[S->index_to_relax[k]] = v_new[k];
        double l, pr, po, r;
        orc_generate_model(S, p_new, S->Tcoefs[m], S->init_logL[m], model, &l, &pr, &po);
        if (prop_stats) { prop_stats[3 * m] = l; prop_stats[3 * m + 1] = pr; prop_stats[3 * m + 2] = po; }
        long double lq_fwd = 0, lq_rev = 0;
        double nd1 = 0;
        if (isnan(l) == 0 && po != -INFINITY) {
            orc_fd_gradient_posterior(S, p_new, S->Tcoefs[m], h, g, gp);
            orc_langevin_drift(cov, sigma[m], S->epsi2, delta, g, Nv, d1);
            lq_fwd = orc_mvn_logpdf(v_new, mean, Mm, Nv);                 /* log q(x'|x) = log N(x'; x + drift(x), M) */
            for (long k = 0; k < Nv; k++) mean[k] = v_new[k] + d1[k];
            lq_rev = orc_mvn_logpdf(x, mean, Mm, Nv);                     /* log q(x|x') = log N(x; x' + drift(x'), M) */
            nd1 = norm2(d1, Nv);
            const double e = exp((double)((long double)po - (long double)logPost[m] + lq_rev - lq_fwd));
            r = isnan(e) ? 0.0 : (e < 1. ? e : 1.);   /* a NaN ratio: the reference stops (:519-521), the product rejects */
        } else r = 0.;                                /* :491-493, :522-524 */
        if (diag) { diag[4 * m] = (double)lq_fwd; diag[4 * m + 1] = (double)lq_rev; diag[4 * m + 2] = norm2(d0, Nv); diag[4 * m + 3] = nd1; }
        if (u_mh[m] <= r) {
            memcpy(params + (size_t)m * Np, p_new, (size_t)Np * sizeof(double));
            memcpy(vars + (size_t)m * Nv, v_new, (size_t)Nv * sizeof(double));
            logL[m] = l; logPrior[m] = pr; logPost[m] = po;
            moved[m] = 1;
        } else moved[m] = 0;
        Pmove[m] = r;
        free(g); free(gp); free(d0); free(d1); free(mean); free(v_new); free(p_new); free(model); free(Mm);
    }
    /* adaptation after every chain has been tested: chain 0's mean feeds the steps h of the NEXT iteration only */
    if (learn)
        for (long m = 0; m < C; m++)
            if (!chain_mask || chain_mask[m]) orc_update_proposal(mu + (size_t)m * Nv, covarmat + (size_t)m * Nv * Nv, sigma + m, vars + (size_t)m * Nv, Nv, Pmove[m], gamma,
                                S->target_acceptance, S->epsilon1, S->A1);
    *swapped = 0;
    if (do_swap) {
        double Pswap;
        *swapped = orc_parallel_tempering(logL, logPrior, logPost, vars, params, moved, Pmove, S->Tcoefs, Nv, Np, ind_A, u_swap, literal_444, &Pswap);
    }
    free(h);
    return fatal ? -1 : 0;
}
