/*
 * armm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code), part 2: the red-giant model
 *   model_RGB_asympt_aj_AppWidth_HarveyLike_v4                tamcmc/sources/models.cpp:4684-5079
 * and the host pre-step it needs:
 *   mixed-mode solver (ARMM)                                   external/ARMM/solver_mm.cpp:82-773
 *   zeta function, mixed-mode widths / heights / splittings    external/ARMM/bump_DP.cpp:46-262, :531-547
 *   first derivative on a regular grid                         external/ARMM/derivatives_handler.cpp:30-57, :83-109, :136-163, :425-457
 *   cubic / Hermite spline of the frequency bias               external/spline/src/spline.h:208-505 (ttk592/spline, vendored)
 * Plain C restatement written from reading those files for MEANING; every function cites the lines it follows.
 *
 * Pinning status: PINNED on outputs of the reference's own solver.  external/ARMM/tests/scanner/out/out_{0..10}.res are written by
 * external/ARMM/do_solve.cpp:114-121 (solve_mm_asymptotic_O2from_l0 -> ksi_fct2 "precise" -> h_l_rgb) with every input printed in
 * their `!` header (q = 0 .. 1); copies live in tests/golden/armm_scanner/ and tests/test_armm_scanner_fixtures.py requires
 * orc_armm_solve_O2from_l0 + orc_ksi_fct2_precise + the H1/H0 law to reproduce nu_p, dnu_p, nu_g, DPg, nu_m, zeta_pg and H1/H0 to the
 * printed six significant digits for all 11 files (they do: max relative deviation 5e-6 = the print rounding).
 * Not covered by those files, hence still pinned by analytic known-answer tests only (tests/test_oracle_rgb.py): the O2p driver
 * (model_type 0; it shares solver_mm and the pair loop with the pinned driver), the bias spline (reproduces its nodes / a parabola),
 * and the width / splitting laws of the mixed modes.  (external/ARMM/TEST_EXPECTED_OUTPUTS.txt belongs to test functions that no
 * longer exist in the reference tree: unusable.)
 * The solver's grid is Eigen::LinSpaced in the reference, whose rounding differs between Eigen versions: solutions agree to
 * ~resol*factor, not to the bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tamcmc_oracle.h"

#define PI_L 3.141592653589793238L

/* ---------------------------------------------------------------- small helpers */

static double *dalloc(long n) { return (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double)); }

/* Eigen::VectorXd::LinSpaced(n, lo, hi): lo + i*(hi-lo)/(n-1), the last point exactly hi */
static void linspace(double *v, long n, double lo, double hi) {
    if (n == 1) { v[0] = hi; return; }
    const double step = (hi - lo) / (double)(n - 1);
    for (long i = 0; i < n; i++) v[i] = (i == n - 1) ? hi : lo + (double)i * step;
}

/* solver_mm.cpp:143-148, :179-186: p(nu) = nu - nu_p ; g(nu) = Dnu_p atan(q tan(pi 1e6 (1/nu - 1/nu_g)/DPl)) / pi */
static double pnu_fct(double nu, double nu_p) { return (double)((long double)nu - (long double)nu_p); }
static double gnu_fct(double nu, double nu_g, double Dnu_p, double DPl, double q) {
    const long double X = PI_L * (1.L / nu - 1.L / nu_g) * 1e6L / DPl;
    return (double)(Dnu_p * atanl(q * tanl(X)) / PI_L);
}

/* solver_mm.cpp:82-121: indices i where x[i] -> x[i+1] changes sign (zeros count as a change) */
static long sign_change(const double *x, long n, long *pos) {
    long j = 0;
    for (long i = 0; i < n - 1; i++) {
        if ((x[i + 1] >= 0 && x[i] < 0) || (x[i + 1] > 0 && x[i] <= 0)) pos[j++] = i;
        else if (x[i + 1] <= 0 && x[i] >= 0) pos[j++] = i;
    }
    return j;
}

/* derivatives_handler.cpp:425-457: forward / centred / backward differences on an index grid */
static void frstder_adaptive(const double *y, long n, double *d) {
    if (n < 2) { if (n == 1) d[0] = 0; return; }
    d[0] = y[1] - y[0];
    d[n - 1] = y[n - 1] - y[n - 2];
    for (long i = 1; i < n - 1; i++) d[i] = (y[i + 1] - y[i - 1]) / 2.;
}

/* ---------------------------------------------------------------- the core solver: solver_mm.cpp:340-443 */
/* Solutions of p(nu) = g(nu) for one (p mode, g mode) pair on [numin, numax]; returns their number (<= max_out). */
static long solver_mm(double nu_p, double nu_g, double Dnu_p, double DPl, double q, double numin, double numax, double resol,
                      double factor, double *nu_m, long max_out) {
    long n_sol = 0;
    if (!(nu_g >= numin && nu_g <= numax)) return 0;
    long n;
    double lo;
    if (numin >= 0) { n = (long)((numax - numin) / resol); lo = numin; }
    else { n = (long)(numax / resol); lo = 0; }
    if (n < 2) return 0;
    double *nu = dalloc(n), *diff = dalloc(n);
    long *idx = (long *)malloc((size_t)n * sizeof(long));
    linspace(nu, n, lo, numax);
    for (long i = 0; i < n; i++) diff[i] = pnu_fct(nu[i], nu_p) - gnu_fct(nu[i], nu_g, Dnu_p, DPl, q);
    const long ns = sign_change(diff, n, idx);
    for (long s = 0; s < ns && n_sol < max_out; s++) {
        /* fine local grid around the approximate solution, inverse linear interpolation of (p-g) -> 0 (:378-384) */
        const double rmin = nu[idx[s]] - 2 * resol, rmax = nu[idx[s]] + 2 * resol;
        const long nl = (long)((rmax - rmin) / (resol * factor));
        if (nl < 2) continue;
        double *xl = dalloc(nl), *yl = dalloc(nl);
        linspace(xl, nl, rmin, rmax);
        for (long i = 0; i < nl; i++) yl[i] = pnu_fct(xl[i], nu_p) - gnu_fct(xl[i], nu_g, Dnu_p, DPl, q);
        const double prop = orc_lin_interpol(yl, xl, nl, 0.0);
        free(xl); free(yl);
        /* a pole of tan() also flips the sign: keep true intersections only (:401-406) */
        const double ratio = gnu_fct(prop, nu_g, Dnu_p, DPl, q) / pnu_fct(prop, nu_p);
        if (ratio >= 0.999 && ratio <= 1.001) nu_m[n_sol++] = prop;
    }
    free(nu); free(diff); free(idx);
    return n_sol;
}

static int cmp_dbl(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

void orc_eigensols_free(orc_eigensols *e) {
    if (!e) return;
    free(e->nu_m); free(e->nu_p); free(e->nu_g); free(e->dnup); free(e->dPg);
    memset(e, 0, sizeof *e);
}

/* shared tail of the two drivers (:556-598, :706-748): every (p, g) pair, keep the solutions inside [keep_lo, keep_hi],
 * sort, drop neighbours closer than 2*resol */
static void solve_pairs(const double *nu_p, const double *Dnu_loc, long Lp, const double *nu_g, long Lg, double Dnu_p, double DPl, double q,
                        double zone, double resol, double fact, double keep_lo, double keep_hi, orc_eigensols *out) {
    long cap = 1024, n = 0;
    double *all = dalloc(cap);
    double tmp[64];
    for (long ip = 0; ip < Lp; ip++)
        for (long ig = 0; ig < Lg; ig++) {
            const long k = solver_mm(nu_p[ip], nu_g[ig], Dnu_loc[ip], DPl, q, nu_p[ip] - zone * Dnu_p, nu_p[ip] + zone * Dnu_p, resol, fact, tmp,
                                     64);
            for (long i = 0; i < k; i++)
                if (tmp[i] >= keep_lo && tmp[i] <= keep_hi) {
                    if (n == cap) { cap *= 2; all = (double *)realloc(all, (size_t)cap * sizeof(double)); }
                    all[n++] = tmp[i];
                }
        }
    qsort(all, (size_t)n, sizeof(double), cmp_dbl);
    const double tol = 2 * resol;
    long m = 0;
    for (long i = 0; i < n; i++)  /* std::unique with |a-b| <= tol: compares with the last KEPT element */
        if (m == 0 || !(fabs(all[m - 1] - all[i]) <= tol)) all[m++] = all[i];
    out->nu_m = all;
    out->n_m = m;
}

/* solver_mm.cpp:470-611: p modes from the second-order asymptotic relation */
int orc_armm_solve_O2p(double Dnu_p, double epsilon, int el, double delta0l, double alpha_p, double nmax, double DPl, double alpha, double q,
                       double fmin, double fmax, double resol, orc_eigensols *out) {
    memset(out, 0, sizeof *out);
    double fact = 0.04;
    int np_min = (int)floor(fmin / Dnu_p - epsilon - el / 2 - delta0l);  /* el/2: INTEGER division, as written in the reference */
    int np_max = (int)ceil(fmax / Dnu_p - epsilon - el / 2 - delta0l);
    np_min = (int)floor(np_min - alpha_p * pow(np_min - nmax, 2) / 2.);
    np_max = (int)ceil(np_max + alpha_p * pow(np_max - nmax, 2) / 2.);
    int ng_min = (int)floor(1e6 / (fmax * DPl) - alpha);
    int ng_max = (int)ceil(1e6 / (fmin * DPl) - alpha);
    if (ng_min <= 0 && ng_max < 1) return ORC_OK;  /* "impossible star": the reference returns an EMPTY structure and carries on */
    if (ng_min <= 0 && ng_max >= 1) ng_min = 1;
    const double zone = (ng_max - ng_min < 6) ? (double)np_max : 1.75;
    if (np_min <= 0) np_min = 1;
    if (fmin <= 150) fact = 0.01;
    if (fmin <= 50) fact = 0.005;
    const long Lp = np_max - np_min, Lg = ng_max - ng_min;
    if (Lg < 1) return ORC_OK;  /* no g mode in range: nothing to couple with (empty set) */
    if (Lp < 1) return ORC_ERR_BAD_ARG;
    out->nu_p = dalloc(Lp); out->nu_g = dalloc(Lg); out->dnup = dalloc(Lp); out->dPg = dalloc(Lg);
    out->n_p = Lp; out->n_g = Lg;
    double *loc = dalloc(Lp);
    for (int np = np_min; np < np_max; np++) {  /* asympt_nu_p, :201-211 */
        out->nu_p[np - np_min] = (double)((np + (long double)epsilon + el / 2.L + delta0l + alpha_p * powl(np - nmax, 2) / 2) * Dnu_p);
        loc[np - np_min] = Dnu_p * (1.0 + alpha_p * (np - nmax));  /* local large separation handed to the solver (:566) */
    }
    for (int ng = ng_min; ng < ng_max; ng++) out->nu_g[ng - ng_min] = (double)(1e6L / ((ng + (long double)alpha) * DPl));  /* asympt_nu_g */
    frstder_adaptive(out->nu_p, Lp, out->dnup);
    for (long i = 0; i < Lg; i++) out->dPg[i] = DPl;
    solve_pairs(out->nu_p, loc, Lp, out->nu_g, Lg, Dnu_p, DPl, q, zone, resol, fact, fmin, fmax, out);
    free(loc);
    return ORC_OK;
}

/* solver_mm.cpp:624-760: p modes = the observed l=0 frequencies shifted by l/2 Dnu + delta0l (asympt_nu_p_from_l0_Xd, :261-301) */
int orc_armm_solve_O2from_l0(const double *nu_l0, long n0, int el, double delta0l, double DPl, double alpha, double q, double resol,
                             double freq_min, double freq_max, orc_eigensols *out) {
    memset(out, 0, sizeof *out);
    if (n0 < 2) return ORC_ERR_BAD_ARG;
    double *idx = dalloc(n0), fit[2];
    for (long i = 0; i < n0; i++) idx[i] = (double)i;
    orc_linfit(idx, nu_l0, n0, fit);
    free(idx);
    const double Dnu_p = fit[0];
    double lo0 = nu_l0[0], hi0 = nu_l0[0];
    for (long i = 1; i < n0; i++) { if (nu_l0[i] < lo0) lo0 = nu_l0[i]; if (nu_l0[i] > hi0) hi0 = nu_l0[i]; }
    double fmin = lo0 - Dnu_p, fmax = hi0 + Dnu_p, fact = 0.04;
    if (fmin < 0) fmin = 0;
    int ng_min = (int)floor(1e6 / (fmax * DPl) - alpha);
    int ng_max = (int)ceil(1e6 / (fmin * DPl) - alpha);
    if (ng_min <= 0 && ng_max < 1) return ORC_OK;  /* empty structure, as above */
    if (ng_min <= 0 && ng_max >= 1) ng_min = 1;
    const double zone = (ng_max - ng_min < 6) ? 20. : 1.75;
    if (fmin <= 150) fact = 0.01;
    if (fmin <= 50) fact = 0.005;
    /* l=0 list extended by three orders on each side, shifted, kept inside [fmin, fmax] */
    const long nl = n0 + 6;
    double *ext = dalloc(nl);
    ext[0] = lo0 - 3 * Dnu_p; ext[1] = lo0 - 2 * Dnu_p; ext[2] = lo0 - Dnu_p;
    for (long k = 0; k < n0; k++) ext[k + 3] = nu_l0[k];
    ext[nl - 3] = hi0 + Dnu_p; ext[nl - 2] = hi0 + 2 * Dnu_p; ext[nl - 1] = hi0 + 3 * Dnu_p;
    out->nu_p = dalloc(nl);
    long Lp = 0;
    for (long k = 0; k < nl; k++) {
        const double v = ext[k] + (double)(el / 2.L * Dnu_p + delta0l);
        if (v >= fmin && v <= fmax) out->nu_p[Lp++] = v;
    }
    free(ext);
    const long Lg = ng_max - ng_min;
    if (Lg < 1) { orc_eigensols_free(out); return ORC_OK; }
    if (Lp < 2) { orc_eigensols_free(out); return ORC_ERR_BAD_ARG; }
    out->n_p = Lp; out->n_g = Lg;
    out->nu_g = dalloc(Lg); out->dnup = dalloc(Lp); out->dPg = dalloc(Lg);
    for (int ng = ng_min; ng < ng_max; ng++) out->nu_g[ng - ng_min] = (double)(1e6L / ((ng + (long double)alpha) * DPl));
    frstder_adaptive(out->nu_p, Lp, out->dnup);
    for (long i = 0; i < Lg; i++) out->dPg[i] = DPl;
    solve_pairs(out->nu_p, out->dnup, Lp, out->nu_g, Lg, Dnu_p, DPl, q, zone, resol, fact, freq_min, freq_max, out);
    return ORC_OK;
}

/* ---------------------------------------------------------------- zeta: bump_DP.cpp:46-78, :125-188 */
static double ksi_one(double nu, double nu_p, double nu_g, double Dnu_p, double DPl, double q) {
    const double up = (double)(M_PI * 1e6 * (1. / nu - 1. / nu_g) / DPl);
    const double down = (double)(M_PI * (nu - nu_p) / Dnu_p);
    const double front = 1e-6 * nu * nu * DPl / (q * Dnu_p);
    const double cu = cos(up), cd = cos(down);
    return 1. / (1. + front * ((cu * cu) / (cd * cd)));
}

/* "precise" normalisation: the sum over all (p, g) pairs divided by its maximum over a 4-year-resolution grid, clipped at 1 */
void orc_ksi_fct2_precise(const double *nu, long n, const double *nu_p, const double *Dnu_p, long Lp, const double *nu_g, const double *DPl,
                          long Lg, double q, double *ksi) {
    const long double resol = 1e6L / (4 * 365. * 86400.);
    double pmin = nu_p[0], pmax = nu_p[0], gmin = nu_g[0], gmax = nu_g[0];
    for (long i = 1; i < Lp; i++) { if (nu_p[i] < pmin) pmin = nu_p[i]; if (nu_p[i] > pmax) pmax = nu_p[i]; }
    for (long i = 1; i < Lg; i++) { if (nu_g[i] < gmin) gmin = nu_g[i]; if (nu_g[i] > gmax) gmax = nu_g[i]; }
    const double fmin = pmin >= gmin ? gmin : pmin, fmax = pmax >= gmax ? pmax : gmax;
    const long nh = (long)((fmax - fmin) / resol);
    for (long i = 0; i < n; i++) {
        double s = 0;
        for (long ip = 0; ip < Lp; ip++) {
            double loc = 0;
            for (long ig = 0; ig < Lg; ig++) loc += ksi_one(nu[i], nu_p[ip], nu_g[ig], Dnu_p[ip], DPl[ig], q);
            s += loc;
        }
        ksi[i] = s;
    }
    double norm = 0;
    if (nh >= 2) {
        const double step = (fmax - fmin) / (double)(nh - 1);
#pragma omp parallel for reduction(max : norm) schedule(static)
        for (long i = 0; i < nh; i++) {
            const double x = (i == nh - 1) ? fmax : fmin + (double)i * step;
            double s = 0;
            for (long ip = 0; ip < Lp; ip++) {
                double loc = 0;
                for (long ig = 0; ig < Lg; ig++) loc += ksi_one(x, nu_p[ip], nu_g[ig], Dnu_p[ip], DPl[ig], q);
                s += loc;
            }
            if (s > norm) norm = s;
        }
    }
    for (long i = 0; i < n; i++) {
        ksi[i] = ksi[i] / norm;
        if (ksi[i] > 1) ksi[i] = 1;
    }
}

/* ---------------------------------------------------------------- spline of the bias: spline.h:208-505 */
/* natural boundaries (second derivative 0 at both ends); type 1 = C2 cubic spline, type 2 = C1 cubic Hermite spline */
typedef struct { long n; double *x, *y, *b, *c, *d; double c0; } orc_spline;

static void spline_free(orc_spline *s) { free(s->x); free(s->y); free(s->b); free(s->c); free(s->d); }

static int spline_set(orc_spline *s, const double *x, const double *y, long n, int type) {
    if (n < 3) return ORC_ERR_BAD_ARG;
    for (long i = 0; i < n - 1; i++)
        if (!(x[i] < x[i + 1])) return ORC_ERR_BAD_ARG;
    s->n = n;
    s->x = dalloc(n); s->y = dalloc(n); s->b = dalloc(n); s->c = dalloc(n); s->d = dalloc(n);
    memcpy(s->x, x, (size_t)n * sizeof(double));
    memcpy(s->y, y, (size_t)n * sizeof(double));
    if (type == 1) {
        /* tridiagonal system for the c_i (:300-345), solved by elimination (the reference: banded LU -- same system) */
        double *sub = dalloc(n), *dia = dalloc(n), *sup = dalloc(n), *rhs = dalloc(n);
        for (long i = 1; i < n - 1; i++) {
            sub[i] = 1.0 / 3.0 * (x[i] - x[i - 1]);
            dia[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1]);
            sup[i] = 1.0 / 3.0 * (x[i + 1] - x[i]);
            rhs[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
        }
        dia[0] = 2.0; sup[0] = 0.0; rhs[0] = 0.0; sub[0] = 0.0;
        dia[n - 1] = 2.0; sub[n - 1] = 0.0; rhs[n - 1] = 0.0; sup[n - 1] = 0.0;
        for (long i = 1; i < n; i++) {
            const double w = sub[i] / dia[i - 1];
            dia[i] -= w * sup[i - 1];
            rhs[i] -= w * rhs[i - 1];
        }
        s->c[n - 1] = rhs[n - 1] / dia[n - 1];
        for (long i = n - 2; i >= 0; i--) s->c[i] = (rhs[i] - sup[i] * s->c[i + 1]) / dia[i];
        free(sub); free(dia); free(sup); free(rhs);
        for (long i = 0; i < n - 1; i++) {
            s->d[i] = 1.0 / 3.0 * (s->c[i + 1] - s->c[i]) / (x[i + 1] - x[i]);
            s->b[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - 1.0 / 3.0 * (2.0 * s->c[i] + s->c[i + 1]) * (x[i + 1] - x[i]);
        }
        const double h = x[n - 1] - x[n - 2];
        s->d[n - 1] = 0.0;
        s->b[n - 1] = 3.0 * s->d[n - 2] * h * h + 2.0 * s->c[n - 2] * h + s->b[n - 2];
    } else {
        for (long i = 1; i < n - 1; i++) {  /* three-point slopes (:360-366) */
            const double h = x[i + 1] - x[i], hl = x[i] - x[i - 1];
            s->b[i] = -h / (hl * (hl + h)) * y[i - 1] + (h - hl) / (hl * h) * y[i] + hl / (h * (hl + h)) * y[i + 1];
        }
        {
            const double h = x[1] - x[0];
            s->b[0] = 0.5 * (-s->b[1] - 0.5 * 0.0 * h + 3.0 * (y[1] - y[0]) / h);
        }
        {
            const double h = x[n - 1] - x[n - 2];
            s->b[n - 1] = 0.5 * (-s->b[n - 2] + 0.5 * 0.0 * h + 3.0 * (y[n - 1] - y[n - 2]) / h);
            s->c[n - 1] = 0.5 * 0.0;
        }
        s->d[n - 1] = 0.0;
        for (long i = 0; i < n - 1; i++) {  /* set_coeffs_from_b (:219-239) */
            const double h = x[i + 1] - x[i];
            s->c[i] = (3.0 * (y[i + 1] - y[i]) / h - (2.0 * s->b[i] + s->b[i + 1])) / h;
            s->d[i] = ((s->b[i + 1] - s->b[i]) / (3.0 * h) - 2.0 / 3.0 * s->c[i]) / h;
        }
    }
    s->c0 = s->c[0];
    return ORC_OK;
}

static double spline_eval(const orc_spline *s, double x) {  /* :476-498: quadratic extrapolation outside the nodes */
    const long n = s->n;
    long idx = 0;
    while (idx + 1 < n && s->x[idx + 1] <= x) idx++;  /* last node <= x (0 when x is left of the first) */
    const double h = x - s->x[idx];
    if (x < s->x[0]) return (s->c0 * h + s->b[0]) * h + s->y[0];
    if (x > s->x[n - 1]) return (s->c[n - 1] * h + s->b[n - 1]) * h + s->y[n - 1];
    return ((s->d[idx] * h + s->c[idx]) * h + s->b[idx]) * h + s->y[idx];
}

double orc_spline_eval(const double *xn, const double *yn, long n, int type, double x) {
    orc_spline s;
    if (spline_set(&s, xn, yn, n, type) != ORC_OK) return NAN;
    const double v = spline_eval(&s, x);
    spline_free(&s);
    return v;
}

/* ---------------------------------------------------------------- the model: models.cpp:4684-5079 */
static double app_width(const double g[6], double f) {  /* Appourchaux et al. 2014/2016 width law (:4788-4794) */
    const double lnGamma0 = g[2] * log(f / g[0]) + log(g[3]);
    const double e = 2. * log(f / g[1]) / log(g[4] / g[0]);
    const double lnLorentz = -log(g[5]) / (1. + pow(e, 2));
    return exp(lnGamma0 + lnLorentz);
}

/* cte_width = 0: model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (models.cpp:4684-5079)
 * cte_width = 1: model_RGB_asympt_aj_CteWidth_HarveyLike_v4 (models.cpp:4334-4682): one width parameter, every l=0/2/3 width is that
 *                constant (:4407), and the fmin - Dnu_p >= 0 requirement sits inside the model_type == 0 branch only (:4453-4466). */
static int rgb_v4_modes(const double *params, const int *pl, double step, int cte_width, orc_rgb_modes *out) {
    memset(out, 0, sizeof *out);
    const long double pi = M_PI;
    const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5], Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8],
              Ninc = pl[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const int ocfg = Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise + Ninc;
    const int do_amp = params[ocfg + 1] != 0;
    const double model_type = params[ocfg + 3], bias_type = params[ocfg + 4];
    const int Nferr = (int)params[ocfg + 5];
    if (Nmax < 2 || Nmax != Nfl0 || Nferr < 0 || Nfl1 != 8 + 2 * Nferr || Nwidth < (cte_width ? 1 : 6)) return ORC_ERR_BAD_ARG;
    double g[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < (cte_width ? 1 : 6); k++) g[k] = fabs(params[Nmax + lmax + Nf + Nsplit + k]);
    const double *fl0 = params + Nmax + lmax;
    double *Wl0 = dalloc(Nmax), *Hl0 = dalloc(Nmax);
    for (int n = 0; n < Nmax; n++) Wl0[n] = cte_width ? g[0] : app_width(g, fl0[n]);
    for (int n = 0; n < Nmax; n++) Hl0[n] = do_amp ? (double)fabsl(params[n] * (1. / Wl0[n] / pi)) : fabs(params[n]);
    const int o1 = Nmax + lmax + Nfl0;
    const double delta0l = params[o1], DPl = fabs(params[o1 + 1]), alpha_g = fabs(params[o1 + 2]), q_star = fabs(params[o1 + 3]);
    const double Wfactor = fabs(params[o1 + 6]), Hfactor = fabs(params[o1 + 7]);
    const double *fref = params + o1 + 8, *ferr = params + o1 + 8 + Nferr;
    const int os = Nmax + lmax + Nf;
    const double rot_env = fabs(params[os]), rot_core = fabs(params[os + 1]);
    orc_spline bias;
    int have_bias = 0;
    if (bias_type != 0) {
        if (spline_set(&bias, fref, ferr, Nferr, bias_type == 1 ? 1 : 2) != ORC_OK) { free(Wl0); free(Hl0); return ORC_ERR_BAD_ARG; }
        have_bias = 1;
    }
    double fmin = fl0[0], fmax = fl0[0];
    for (int n = 1; n < Nfl0; n++) { if (fl0[n] < fmin) fmin = fl0[n]; if (fl0[n] > fmax) fmax = fl0[n]; }
    double *xi = dalloc(Nfl0), rfit[2];
    for (int n = 0; n < Nfl0; n++) xi[n] = (double)n;
    orc_linfit(xi, fl0, Nfl0, rfit);
    free(xi);
    const double Dnu_p = rfit[0];
    const int n0 = (int)floor(rfit[1] / Dnu_p);
    const double epsilon_p = rfit[1] / Dnu_p - n0;
    int rc = ORC_OK;
    orc_eigensols sol;
    /* the reference exits: infinite g-mode density (:4851-4857).  The constant-width variant only tests this when model_type == 0;
     * with model_type != 0 its solver would be handed a zero lower bound (1e6/(0*DPl) g modes), which is refused here as well. */
    if (fmin - Dnu_p < 0) rc = ORC_ERR_BAD_ARG;
    else if (model_type == 0) rc = orc_armm_solve_O2p(Dnu_p, epsilon_p, 1, delta0l, 0, 0., DPl, alpha_g, q_star, fmin - Dnu_p, fmax + Dnu_p, step, &sol);
    else rc = orc_armm_solve_O2from_l0(fl0, Nfl0, 1, delta0l, DPl, alpha_g, q_star, step, fmin, fmax, &sol);
    if (rc != ORC_OK) { free(Wl0); free(Hl0); if (have_bias) spline_free(&bias); return rc; }
    const long N1 = sol.n_m;
    double *fl1 = dalloc(N1), *ksi = dalloc(N1), *hr = dalloc(N1), *Hl1 = dalloc(N1), *Wl1 = dalloc(N1), *a1 = dalloc(N1);
    for (long i = 0; i < N1; i++) fl1[i] = sol.nu_m[i] + (have_bias ? spline_eval(&bias, sol.nu_m[i]) : 0.0);
    if (N1 > 0) orc_ksi_fct2_precise(fl1, N1, sol.nu_p, sol.dnup, sol.n_p, sol.nu_g, sol.dPg, sol.n_g, q_star, ksi);
    /* h_l_rgb (bump_DP.cpp:235-254): sqrt(1 - Hfactor*zeta), zeros lifted to 1e-10 */
    for (long i = 0; i < N1; i++) {
        hr[i] = sqrt(1. - Hfactor * ksi[i]);
        if (hr[i] > -1e-5 && hr[i] < 1e-5) hr[i] = 1e-10;
    }
    /* l=0 heights interpolated to the l=1 frequencies on an extended grid that falls to 0 (:4875-4901) */
    const long ni = Nfl0 + 4;
    double *fi = dalloc(ni), *hi = dalloc(ni);
    fi[0] = fmin * 0.6; fi[1] = fmin * 0.8; fi[ni - 2] = fmax * 1.2; fi[ni - 1] = fmax * 1.4;
    hi[0] = 0; hi[1] = Hl0[0] / 4; hi[ni - 2] = Hl0[Nmax - 1] / 4; hi[ni - 1] = 0;
    for (int j = 0; j < Nfl0; j++) { fi[j + 2] = fl0[j]; hi[j + 2] = Hl0[j]; }
    const double Vl1 = lmax >= 1 ? fabs(params[Nmax]) : 0.0;
    for (long i = 0; i < N1; i++) {
        const double t = orc_lin_interpol(fi, hi, ni, fl1[i]);
        const double Hp = t < 0 ? 0.0 : fabs(t);
        Hl1[i] = fabs(hr[i] * (Hp * Vl1));
        /* gamma_l_fct2 (bump_DP.cpp:203-222) */
        Wl1[i] = orc_lin_interpol(fl0, Wl0, Nfl0, fl1[i]) * (1. - Wfactor * ksi[i]) / sqrt(hr[i]);
        /* dnu_rot_2zones (bump_DP.cpp:531-547) */
        a1[i] = fabs(ksi[i] * (rot_core / 2 - rot_env) + rot_env);
    }
    free(fi); free(hi); free(hr);
    if (have_bias) spline_free(&bias);
    out->N0 = Nfl0; out->fl0 = dalloc(Nfl0); memcpy(out->fl0, fl0, (size_t)Nfl0 * sizeof(double));
    out->Wl0 = Wl0; out->Hl0 = Hl0;
    out->N1 = N1; out->fl1 = fl1; out->Wl1 = Wl1; out->Hl1 = Hl1; out->a1_l1 = a1; out->ksi = ksi;
    out->g[0] = g[0]; out->g[1] = g[1]; out->g[2] = g[2]; out->g[3] = g[3]; out->g[4] = g[4]; out->g[5] = g[5];
    orc_eigensols_free(&sol);
    return ORC_OK;
}

int orc_rgb_v4_modes(const double *params, const int *pl, double step, orc_rgb_modes *out) { return rgb_v4_modes(params, pl, step, 0, out); }
int orc_rgb_v4_cte_modes(const double *params, const int *pl, double step, orc_rgb_modes *out) { return rgb_v4_modes(params, pl, step, 1, out); }

void orc_rgb_modes_free(orc_rgb_modes *m) {
    if (!m) return;
    free(m->fl0); free(m->Wl0); free(m->Hl0); free(m->fl1); free(m->Wl1); free(m->Hl1); free(m->a1_l1); free(m->ksi);
    memset(m, 0, sizeof *m);
}

static int rgb_v4_model(const double *params, const int *pl, const double *x, long Nx, double *model, int cte_width) {
    if (Nx < 3) return ORC_ERR_BAD_ARG;
    const long double pi = M_PI;
    const double step = x[2] - x[1];
    const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5], Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8],
              Ninc = pl[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const int ocfg = Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise + Ninc;
    const double trunc_c = params[ocfg];
    const int do_amp = params[ocfg + 1] != 0;
    orc_rgb_modes md;
    int rc = rgb_v4_modes(params, pl, step, cte_width, &md);
    if (rc != ORC_OK) return rc;
    const double inclination = fabs(params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise]);
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double Vl2 = 0, Vl3 = 0;
    if (lmax >= 1) orc_amplitude_ratio(1, inclination, r1);
    if (lmax >= 2) { Vl2 = fabs(params[Nmax + 1]); orc_amplitude_ratio(2, inclination, r2); }
    if (lmax >= 3) { Vl3 = fabs(params[Nmax + 2]); orc_amplitude_ratio(3, inclination, r3); }
    const int os = Nmax + lmax + Nf;
    const double rot_env = fabs(params[os]);
    const double a2_env = params[os + 2], a3_env = params[os + 4], a4_env = params[os + 5], a5_env = params[os + 6], a6_env = params[os + 7];
    const double eta_switch = params[os + 8], asym = params[os + 9];
    const double eta0 = (eta_switch == 1) ? orc_eta0_fct(md.fl0, Nfl0) : 0.0;
    for (long i = 0; i < Nx; i++) model[i] = 0.0;
    for (int n = 0; n < Nfl0 && rc == ORC_OK; n++)
        rc = orc_optimum_lorentzian_calc_aj(x, model, Nx, md.Hl0[n], md.fl0[n], 0, 0, 0, 0, 0, 0, 0, asym, md.Wl0[n], 0, r0, step, trunc_c);
    for (long n = 0; n < md.N1 && rc == ORC_OK; n++)  /* a2 of the mixed modes is NOT used (:4949) */
        rc = orc_optimum_lorentzian_calc_aj(x, model, Nx, md.Hl1[n], md.fl1[n], md.a1_l1[n], 0, 0, 0, 0, 0, eta0, asym, md.Wl1[n], 1, r1, step,
                                            trunc_c);
    for (int n = 0; n < Nfl2 && rc == ORC_OK; n++) {
        const double fl2 = fabs(params[Nmax + lmax + Nfl0 + Nfl1 + n]);
        const double W = cte_width ? md.g[0] : app_width(md.g, fl2);  /* Cte: Wl0_all[n], all equal (:4561) */
        double H = orc_lin_interpol(md.fl0, md.Hl0, Nfl0, fl2);
        H = do_amp ? (double)fabsl(H / (pi * W) * Vl2) : fabs(H * Vl2);
        rc = orc_optimum_lorentzian_calc_aj(x, model, Nx, H, fl2, rot_env, a2_env, a3_env, a4_env, 0, 0, eta0, asym, W, 2, r2, step, trunc_c);
    }
    for (int n = 0; n < Nfl3 && rc == ORC_OK; n++) {
        const double fl3 = fabs(params[Nmax + lmax + Nfl0 + Nfl1 + Nfl2 + n]);
        const double W = cte_width ? md.g[0] : app_width(md.g, fl3);
        double H = orc_lin_interpol(md.fl0, md.Hl0, Nfl0, fl3);
        H = do_amp ? (double)fabsl(H / (pi * W) * Vl3) : fabs(H * Vl3);
        rc = orc_optimum_lorentzian_calc_aj(x, model, Nx, H, fl3, rot_env, a2_env, a3_env, a4_env, a5_env, a6_env, eta0, asym, W, 3, r3, step, trunc_c);
    }
    orc_rgb_modes_free(&md);
    if (rc != ORC_OK) return rc;
    double *np = dalloc(Nnoise);
    for (int k = 0; k < Nnoise; k++) np[k] = fabs(params[Nmax + lmax + Nf + Nsplit + Nwidth + k]);
    orc_harvey_like(np, Nnoise, x, model, Nx, (Nnoise - 1) / 3);
    free(np);
    return ORC_OK;
}

int orc_model_RGB_asympt_aj_AppWidth_HarveyLike_v4(const double *params, const int *pl, const double *x, long Nx, double *model) {
    return rgb_v4_model(params, pl, x, Nx, model, 0);
}

int orc_model_RGB_asympt_aj_CteWidth_HarveyLike_v4(const double *params, const int *pl, const double *x, long Nx, double *model) {
    return rgb_v4_model(params, pl, x, Nx, model, 1);
}
