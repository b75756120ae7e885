/*
 * tamcmc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See tamcmc_oracle.h for the pinning status and the import rules.
 *
 * Plain C restatement, written from the formula sheet of SURVEY.md App. A and
 * from reading the reference for MEANING (operation order, types, branch
 * structure); no reference text is reproduced.  The reference is built with
 * plain -O3 on x86-64 (CMakeLists.txt:38-41: no -march, hence no FMA
 * contraction), so this file must be compiled with -ffp-contract=off.
 * `long double` is used exactly where the reference's expressions are
 * evaluated in long double (Pslm products, the pi used for amplitudes,
 * the likelihood scalars, the priors).
 */
#include "tamcmc_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI_L 3.141592653589793238462643383279502884L

/* ------------------------------------------------------------------ */
/* a-coefficient polynomials                                           */
/* ------------------------------------------------------------------ */

/* acoefs.cpp:19-49 -- Ritzwoller & Lavely (1991) polynomials, s<=6.
 * Integer-valued; the reference evaluates them in double (pow) and stores
 * them in a long double. */
long double orc_Hslm_Ritzoller1991(int s, int l, int m) {
    const int L = l * (l + 1);
    const double dm = (double)m;
    long double H = 0.0L;
    switch (s) {
    case 0: H = 1; break;
    case 1: H = 2 * m; break;
    case 2: H = 6 * pow(dm, 2) - 2 * L; break;
    case 3: H = 20 * pow(dm, 3) - 4 * (3 * L - 1) * m; break;
    case 4: H = 70 * pow(dm, 4) - 10 * (6 * L - 5) * pow(dm, 2) + 6 * L * (L - 2); break;
    case 5: H = 252 * pow(dm, 5) - 140 * (2 * L - 3) * pow(dm, 3) + (20 * L * (3 * L - 10) + 48) * m; break;
    case 6:
        H = 924 * pow(dm, 6) - 420 * pow(dm, 4) * (3 * L - 7) +
            84 * pow(dm, 2) * (5 * pow((double)L, 2) - 25 * L + 14) - 20 * L * (pow((double)L, 2) - 8 * L + 12);
        break;
    default: H = -1; break;
    }
    return H;
}

/* acoefs.cpp:51-110 -- P_s^{(l)}(m) normalised so that P_s(l)=l
 * (Schou, Christensen-Dalsgaard & Thompson 1994).  s=2,3 are double
 * quotients, s=4..6 long double quotients; zero when the normalisation
 * vanishes. */
long double orc_Pslm(int s, int l, int m) {
    const double dm = (double)m, dl = (double)l;
    long double H, c, Ps = 0.0L;
    switch (s) {
    case 0: Ps = l; break;
    case 1: Ps = m; break;
    case 2:
        if (l > 0) {
            double q = (3 * pow(dm, 2) - l * (l + 1)) / (2 * l - 1);
            Ps = q;
        } else
            Ps = 0;
        break;
    case 3:
        if (l > 1) {
            double q = (5 * pow(dm, 3) - (3 * l * (l + 1) - 1) * m) / ((l - 1) * (2 * l - 1));
            Ps = q;
        } else
            Ps = 0;
        break;
    case 4: {
        double h = (35 * pow(dm, 4) - 5 * (6 * l * (l + 1) - 5) * pow(dm, 2)) + 3 * l * (l + 1) * (l * (l + 1) - 2);
        H = h;
        c = 2 * (l - 1) * (2 * l - 1) * (2 * l - 3);
        Ps = (c != 0) ? H / c : 0.0L;
        break;
    }
    case 5: {
        H = orc_Hslm_Ritzoller1991(5, l, m);
        double cd = 8 * (4 * pow(dl, 4) - 20 * pow(dl, 3) + 35 * pow(dl, 2) - 25 * l + 6);
        c = cd;
        Ps = (c != 0) ? H / c : 0.0L;
        break;
    }
    case 6: {
        H = orc_Hslm_Ritzoller1991(6, l, m);
        double cd = 64 * pow(dl, 5) - 480 * pow(dl, 4) + 1360 * pow(dl, 3) - 1800 * pow(dl, 2) + 1096 * l - 240;
        c = cd;
        Ps = (c != 0) ? H / c : 0.0L;
        break;
    }
    default: Ps = 0; break;
    }
    return Ps;
}

/* build_lorentzian.cpp:583-592 -- centrifugal-distortion factor, including
 * the 2/3 (held as a long double copy of the double 2./3). */
double orc_Qlm(int l, int m) {
    const long double Dnl = (long double)(2. / 3);
    double Q = (l * (l + 1) - 3 * pow((double)m, 2)) / ((2 * l - 1) * (2 * l + 3));
    Q = (double)(Q * Dnl);
    return Q;
}

/* ------------------------------------------------------------------ */
/* m-component visibilities                                            */
/* ------------------------------------------------------------------ */

/* function_rot.cpp:94-101 -- int-valued factorial (long inside, int out) */
static int orc_factorial(int n) {
    long f = 1;
    for (long i = 1; i <= n; i++) f = f * i;
    return (int)f;
}
/* function_rot.cpp:90-92 -- integer divisions, then promoted to double */
static double orc_combi(int n, int r) { return (double)(orc_factorial(n) / orc_factorial(n - r) / orc_factorial(r)); }

/* function_rot.cpp:76-88 -- Wigner small-d element d^l_{m1,m2}(beta) */
static double orc_dmm(int l, int m1, int m2, double beta) {
    double sum = 0, var = 0;
    for (long s = 0; s <= l - m1; s++) {
        var = orc_combi(l + m2, (int)(l - m1 - s)) * orc_combi(l - m2, (int)s) * pow(-1, (double)(l - m1 - s));
        var = var * pow(cos(beta / 2.), (double)(2 * s + m1 + m2)) * pow(sin(beta / 2.), (double)(2 * l - 2 * s - m1 - m2));
        sum = sum + var;
    }
    sum = sum * sqrt((double)(orc_factorial(l + m1) * orc_factorial(l - m1)));
    sum = sum / sqrt((double)(orc_factorial(l + m2) * orc_factorial(l - m2)));
    return sum;
}

/* function_rot.cpp:15-41 + :44-74 -- column m'=0 of the rotation matrix,
 * squared, NOT normalised.  Only that column of the four fill loops of
 * function_rot() is needed: rows i>0 come from dmm(l,i,0,beta), rows i<0 are
 * their mirror times (-1)^i, the centre is finally overwritten by
 * dmm(l,0,0,-beta). */
void orc_amplitude_ratio(int l, double beta_deg, double *V) {
    const double PI = 3.141592653589793238462643;
    const double angle = PI * beta_deg / 180.;
    for (int i = 0; i <= l; i++) V[i + l] = orc_dmm(l, i, 0, angle);
    for (int i = -l; i <= 0; i++) V[i + l] = V[-i + l] * pow(-1, (double)i);
    V[l] = orc_dmm(l, 0, 0, -angle);
    V[l] = V[l] * pow(-1, 0.0);
    for (int i = 0; i < 2 * l + 1; i++) V[i] = V[i] * V[i];
}

/* ------------------------------------------------------------------ */
/* small numerical helpers                                             */
/* ------------------------------------------------------------------ */

/* interpol.cpp:13-43 -- piecewise-linear interpolation with linear
 * extrapolation from the first / last segment. */
double orc_lin_interpol(const double *x, const double *y, long n, double x_int) {
    long i = 0;
    double a = 0, b = 0;
    if (x_int >= x[0] && x_int <= x[n - 1]) {
        while ((x_int < x[i] || x_int > x[i + 1]) && i < n - 2) i = i + 1;
        a = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
        b = y[i] - a * x[i];
    }
    if (x_int < x[0]) {
        a = (y[1] - y[0]) / (x[1] - x[0]);
        b = y[0] - a * x[0];
    }
    if (x_int > x[n - 1]) {
        a = (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]);
        b = y[n - 2] - a * x[n - 2];
    }
    return a * x_int + b;
}

/* linfit.cpp:17-35 -- least-squares line; out[0]=slope, out[1]=intercept.
 * (Eigen's vectorised reduction order is unpinned; sequential sums here.) */
void orc_linfit(const double *x, const double *y, long n, double out[2]) {
    double sx = 0, sy = 0;
    for (long i = 0; i < n; i++) sx += x[i];
    for (long i = 0; i < n; i++) sy += y[i];
    const double dn = (double)n;
    const double mean_x = sx / dn;
    double sty = 0, stt = 0;
    for (long i = 0; i < n; i++) sty += (x[i] - mean_x) * y[i];
    for (long i = 0; i < n; i++) stt += (x[i] - mean_x) * (x[i] - mean_x);
    out[0] = sty / stt;
    out[1] = (sy - sx * out[0]) / dn;
}

/* models.cpp:6073-6084 */
double orc_eta0_from_dnu(double Dnu_obs) {
    const double G = 6.667e-8;
    const double Dnu_sun = 135.1;
    const double R_sun = 6.96342e5;
    const double M_sun = 1.98855e30;
    const double rho_sun = M_sun * 1e3 / (4 * M_PI * pow(R_sun * 1e5, 3) / 3);
    double rho = pow(Dnu_obs / Dnu_sun, 2.) * rho_sun;
    return 3. * M_PI / (rho * G);
}

/* models.cpp:6065-6071 -- Dnu = slope of the l=0 frequencies vs 0..n-1 */
double orc_eta0_fct(const double *fl0, long n) {
    double *idx = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double r[2];
    for (long i = 0; i < n; i++) idx[i] = (double)i;
    orc_linfit(idx, fl0, n, r);
    free(idx);
    return orc_eta0_from_dnu(r[0]);
}

/* ------------------------------------------------------------------ */
/* truncation window                                                   */
/* ------------------------------------------------------------------ */

static int orc_d2i(double v) {
    if (v >= (double)INT_MAX) return INT_MAX;
    if (v <= (double)INT_MIN) return INT_MIN;
    return (int)v;
}

/* build_lorentzian.cpp:595-676 -- four (overlapping) regimes of the
 * half-width, edge clamps, floor/ceil on the regular grid, clip to [0,Nx]. */
int orc_set_imin_imax(const double *x, long Nx, int l, double fc_l, double gamma_l, double f_s, double c, double step,
                      int ivals[2]) {
    double p0 = 0, p1 = 0;
    int set = 0;
    if (gamma_l >= 1 && f_s >= 1) {
        if (l != 0) { p0 = fc_l - c * (l * f_s + gamma_l); p1 = fc_l + c * (l * f_s + gamma_l); }
        else { p0 = fc_l - c * gamma_l * 2.2; p1 = fc_l + c * gamma_l * 2.2; }
        set = 1;
    }
    if (gamma_l <= 1 && f_s >= 1) {
        if (l != 0) { p0 = fc_l - c * (l * f_s + 1); p1 = fc_l + c * (l * f_s + 1); }
        else { p0 = fc_l - c * 2.2; p1 = fc_l + c * 2.2; }
        set = 1;
    }
    if (gamma_l >= 1 && f_s <= 1) {
        if (l != 0) { p0 = fc_l - c * (l + gamma_l); p1 = fc_l + c * (l + gamma_l); }
        else { p0 = fc_l - c * 2.2 * gamma_l; p1 = fc_l + c * 2.2 * gamma_l; }
        set = 1;
    }
    if (gamma_l <= 1 && f_s <= 1) {
        if (l != 0) { p0 = fc_l - c * (l + 1); p1 = fc_l + c * (l + 1); }
        else { p0 = fc_l - c * 2.2; p1 = fc_l + c * 2.2; }
        set = 1;
    }
    if (!set) return ORC_ERR_NAN_WINDOW;
    if ((p1 - step) < x[0]) p1 = x[0] + c;
    if ((p0 + step) >= x[Nx - 1]) p0 = x[Nx - 1] - c;
    ivals[0] = orc_d2i(floor((p0 - x[0]) / step));
    ivals[1] = orc_d2i(ceil((p1 - x[0]) / step));
    if (ivals[0] < 0) ivals[0] = 0;
    if (ivals[1] > Nx) ivals[1] = (int)Nx;
    if (ivals[1] - ivals[0] <= 0) return ORC_ERR_EMPTY_WINDOW;
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* multiplet profiles                                                  */
/* ------------------------------------------------------------------ */

/* build_lorentzian.cpp:226-229 (l!=0) / :232-235 (l==0) */
double orc_nu_nlm_aj(double fc_l, double a1, double a2, double a3, double a4, double a5, double a6, double eta0, int l,
                     int m) {
    if (l == 0) return fc_l;
    long double acc = fc_l + a1 * orc_Pslm(1, l, m) + a2 * orc_Pslm(2, l, m) + a3 * orc_Pslm(3, l, m) +
                      a4 * orc_Pslm(4, l, m) + a5 * orc_Pslm(5, l, m) + a6 * orc_Pslm(6, l, m);
    double nu = (double)acc;
    if (eta0 > 0) nu = nu + fc_l * eta0 * orc_Qlm(l, m) * pow(a1 * 1e-6, 2);
    return nu;
}

/* build_lorentzian.cpp:145 (l!=0) / :148 (l==0) */
double orc_nu_nlm_a1etaa3(double fc_l, double f_s, double eta0, double a3, int l, int m) {
    if (l == 0) return fc_l;
    double t = fc_l * (1. + eta0 * pow(f_s * 1e-6, 2) * orc_Qlm(l, m)) + m * f_s;
    long double acc = t + orc_Pslm(3, l, m) * a3;
    return (double)acc;
}

/* shared per-component accumulation: the Eigen statement sequence of
 * build_lorentzian.cpp:230-241 written per bin in the same operation order */
static void orc_add_component(const double *x_l, long N, double nu, double hv, double fc_l, double gamma_l, double asym,
                              double *result) {
    const double g2 = pow(gamma_l, 2);
    if (asym == 0) {
        for (long i = 0; i < N; i++) {
            double d = x_l[i] - nu;
            double prof = d * d;
            prof = 4 * prof / g2;
            double inv = 1.0 / (1.0 + prof);
            result[i] = result[i] + hv * inv;
        }
    } else {
        const double c2 = 0.5 * gamma_l * asym / fc_l;
        const double c2sq = c2 * c2;
        for (long i = 0; i < N; i++) {
            double d = x_l[i] - nu;
            double prof = d * d;
            prof = 4 * prof / g2;
            double inv = 1.0 / (1.0 + prof);
            double t = 1.0 + asym * (x_l[i] / fc_l - 1.0);
            double asy = t * t + c2sq;
            result[i] = result[i] + hv * (asy * inv);
        }
    }
}

/* build_lorentzian.cpp:131-161 */
void orc_build_l_mode_a1etaa3(const double *x_l, long N, double H_l, double fc_l, double f_s, double eta0, double a3,
                              double asym, double gamma_l, int l, const double *V, double *result) {
    for (long i = 0; i < N; i++) result[i] = 0.0;
    for (int m = -l; m <= l; m++) {
        double nu = orc_nu_nlm_a1etaa3(fc_l, f_s, eta0, a3, l, m);
        orc_add_component(x_l, N, nu, H_l * V[m + l], fc_l, gamma_l, asym, result);
    }
}

/* build_lorentzian.cpp:208-246 */
void orc_build_l_mode_aj(const double *x_l, long N, double H_l, double fc_l, double a1, double a2, double a3, double a4,
                         double a5, double a6, double eta0, double asym, double gamma_l, int l, const double *V,
                         double *result) {
    for (long i = 0; i < N; i++) result[i] = 0.0;
    for (int m = -l; m <= l; m++) {
        double nu = orc_nu_nlm_aj(fc_l, a1, a2, a3, a4, a5, a6, eta0, l, m);
        orc_add_component(x_l, N, nu, H_l * V[m + l], fc_l, gamma_l, asym, result);
    }
}

/* build_lorentzian.cpp:441-458 -- window from (l, fc, gamma, f_s), multiplet on
 * the window, added into the model (the reference copies the whole model in
 * and out; only the windowed add is observable). */
int orc_optimum_lorentzian_calc_a1etaa3(const double *x, double *model, long Nx, double H_l, double fc_l, double f_s,
                                        double eta0, double a3, double asym, double gamma_l, int l, const double *V,
                                        double step, double c) {
    int iv[2];
    int st = orc_set_imin_imax(x, Nx, l, fc_l, gamma_l, f_s, c, step, iv);
    if (st) return st;
    long N = iv[1] - iv[0];
    double *m0 = (double *)malloc(sizeof(double) * (size_t)N);
    orc_build_l_mode_a1etaa3(x + iv[0], N, H_l, fc_l, f_s, eta0, a3, asym, gamma_l, l, V, m0);
    for (long i = 0; i < N; i++) model[iv[0] + i] = model[iv[0] + i] + m0[i];
    free(m0);
    return ORC_OK;
}

/* build_lorentzian.cpp:502-522 and the caller's add (models.cpp:1297-1298) */
int orc_optimum_lorentzian_calc_aj(const double *x, double *model, long Nx, double H_l, double fc_l, double a1,
                                   double a2, double a3, double a4, double a5, double a6, double eta0, double asym,
                                   double gamma_l, int l, const double *V, double step, double c) {
    int iv[2];
    int st = orc_set_imin_imax(x, Nx, l, fc_l, gamma_l, a1, c, step, iv);
    if (st) return st;
    long N = iv[1] - iv[0];
    double *m0 = (double *)malloc(sizeof(double) * (size_t)N);
    orc_build_l_mode_aj(x + iv[0], N, H_l, fc_l, a1, a2, a3, a4, a5, a6, eta0, asym, gamma_l, l, V, m0);
    for (long i = 0; i < N; i++) model[iv[0] + i] = model[iv[0] + i] + m0[i];
    free(m0);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* background and likelihood                                           */
/* ------------------------------------------------------------------ */

/* noise_models.cpp:15-39 -- sum of Harvey-like profiles (skipped when the
 * time-scale is exactly 0) plus white noise, added to the model in place. */
void orc_harvey_like(const double *np, long Nnoise, const double *x, double *model, long Nx, int Nharvey) {
    const double white = np[Nnoise - 1];
    int cpt = 0;
    for (int k = 0; k < Nharvey; k++) {
        if (np[cpt + 1] != 0) {
            const double a = (1e-3) * np[cpt + 1];
            const double pw = np[cpt + 2];
            const double H = np[cpt];
            for (long i = 0; i < Nx; i++) {
                double t = pow(a * x[i], pw);
                t = H * (1.0 / (t + 1.0));
                model[i] = model[i] + t;
            }
        }
        cpt = cpt + 3;
    }
    for (long i = 0; i < Nx; i++) model[i] = model[i] + white;
}

/* likelihoods.cpp:17-28 -- chi^2 with 2p d.o.f.: -p (sum y*(1/M) + sum ln M);
 * both sums are double reductions, added in double, then long double. */
long double orc_likelihood_chi22p(const double *y, const double *model, long Nx, long p) {
    double s1 = 0, s2 = 0;
    for (long i = 0; i < Nx; i++) s1 += y[i] * (1.0 / model[i]);
    for (long i = 0; i < Nx; i++) s2 += log(model[i]);
    long double f = s1 + s2;
    f = -p * f;
    return f;
}

long double orc_likelihood_chi22p_ld(const double *y, const double *model, long Nx, long p) {
    long double s1 = 0, s2 = 0;
    for (long i = 0; i < Nx; i++) s1 += (long double)(y[i] * (1.0 / model[i]));
    for (long i = 0; i < Nx; i++) s2 += logl((long double)model[i]);
    return -p * (s1 + s2);
}

/* model_def.cpp:390-403 -- case 0: p = likelihood_params (double -> long), /T */
double orc_call_likelihood(const double *y, const double *model, long Nx, double likelihood_params, double Tcoef) {
    long double logL = orc_likelihood_chi22p(y, model, Nx, (long)likelihood_params);
    return (double)(logL / Tcoef);
}

/* ------------------------------------------------------------------ */
/* model functions                                                     */
/* ------------------------------------------------------------------ */

static void orc_abs_copy(const double *src, long n, double *dst) {
    for (long i = 0; i < n; i++) dst[i] = fabs(src[i]);
}

/* models.cpp:1195-1408 (serial order of the four l-loops) */
int orc_model_MS_Global_aj_HarveyLike(const double *params, const int *pl, const double *x, long Nx, double *model) {
    const double step = x[1] - x[0];
    const long double pi = M_PI;
    const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8], Ninc = pl[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const double trunc_c = params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise + Ninc];
    const int do_amp = (params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise + Ninc + 1] != 0);
    const double inclination = params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise];
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double Vl[4] = {1, 0, 0, 0};
    double *ratios[4] = {r0, r1, r2, r3};
    for (int l = 1; l <= 3; l++)
        if (lmax >= l) {
            Vl[l] = fabs(params[Nmax + l - 1]);
            orc_amplitude_ratio(l, inclination, ratios[l]);
        }
    const double *fl0_all = params + Nmax + lmax;
    const double *Wl0_all = params + Nmax + lmax + Nf + Nsplit;
    const double *Hl0_all = params;
    const double *sp = params + Nmax + lmax + Nf; /* a1_0,a1_1,...,a6_0,a6_1,eta_switch,asym */
    const double asym = sp[13];
    double eta0 = 0;
    if (sp[12] == 1) eta0 = orc_eta0_fct(fl0_all, Nfl0);
    for (long i = 0; i < Nx; i++) model[i] = 0.0;

    for (int n = 0; n < Nfl0; n++) {
        double fl0 = fl0_all[n];
        double Wl0 = fabs(Wl0_all[n]);
        double Hl0 = do_amp ? (double)fabsl(params[n] / (pi * Wl0)) : fabs(params[n]);
        int st = orc_optimum_lorentzian_calc_aj(x, model, Nx, Hl0, fl0, 0, 0, 0, 0, 0, 0, 0, asym, Wl0, 0, r0, step, trunc_c);
        if (st) return st;
    }
    const int Nfl[4] = {Nfl0, Nfl1, Nfl2, Nfl3};
    int off = Nmax + lmax + Nfl0;
    for (int l = 1; l <= 3; l++) {
        for (int n = 0; n < Nfl[l]; n++) {
            double fl = params[off + n];
            double Wl = fabs(orc_lin_interpol(fl0_all, Wl0_all, Nfl0, fl));
            double Hl;
            if (do_amp) Hl = (double)fabsl(orc_lin_interpol(fl0_all, Hl0_all, Nfl0, fl) / (pi * Wl) * Vl[l]);
            else Hl = fabs(orc_lin_interpol(fl0_all, Hl0_all, Nfl0, fl) * Vl[l]);
            double a[7] = {0, 0, 0, 0, 0, 0, 0};
            const int jmax = 2 * l; /* l=1: a1,a2; l=2: a1..a4; l=3: a1..a6 (models.cpp:1314-1367) */
            for (int j = 1; j <= jmax; j++) a[j] = sp[2 * (j - 1)] + sp[2 * (j - 1) + 1] * (fl * 1e-3);
            int st = orc_optimum_lorentzian_calc_aj(x, model, Nx, Hl, fl, a[1], a[2], a[3], a[4], a[5], a[6], eta0, asym,
                                                    Wl, l, ratios[l], step, trunc_c);
            if (st) return st;
        }
        off += Nfl[l];
    }
    {
        const double *np = params + Nmax + lmax + Nf + Nsplit + Nwidth;
        double *npa = (double *)malloc(sizeof(double) * (size_t)(Nnoise > 0 ? Nnoise : 1));
        orc_abs_copy(np, Nnoise, npa);
        orc_harvey_like(npa, Nnoise, x, model, Nx, (Nnoise - 1) / 3);
        free(npa);
    }
    return ORC_OK;
}

/* models.cpp:1943-2121 (n-major order: l=0,1,2,3 for each radial order) */
int orc_model_MS_Global_a1etaa3_HarveyLike_Classic(const double *params, const int *pl, const double *x, long Nx,
                                                   double *model) {
    const double step = x[1] - x[0];
    const long double pi = ORC_PI_L;
    const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8], Ninc = pl[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const int do_amp = (params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise + Ninc + 1] != 0);
    const double trunc_c = params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise + Ninc];
    const double inclination = params[Nmax + lmax + Nf + Nsplit + Nwidth + Nnoise];
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double Vl[4] = {1, 0, 0, 0};
    double *ratios[4] = {r0, r1, r2, r3};
    for (int l = 1; l <= 3; l++)
        if (lmax >= l) {
            Vl[l] = fabs(params[Nmax + l - 1]);
            orc_amplitude_ratio(l, inclination, ratios[l]);
        }
    const double *fl0_all = params + Nmax + lmax;
    const double *Wl0_all = params + Nmax + lmax + Nf + Nsplit;
    const double a1 = fabs(params[Nmax + lmax + Nf]);
    const double eta0 = orc_eta0_fct(fl0_all, Nfl0);
    const double a3 = params[Nmax + lmax + Nf + 2];
    const double asym = params[Nmax + lmax + Nf + 5];
    for (long i = 0; i < Nx; i++) model[i] = 0.0;
    const int foff[4] = {Nmax + lmax, Nmax + lmax + Nfl0, Nmax + lmax + Nfl0 + Nfl1, Nmax + lmax + Nfl0 + Nfl1 + Nfl2};
    for (long n = 0; n < Nmax; n++) {
        double fl0 = fl0_all[n];
        double Wl0 = fabs(Wl0_all[n]);
        double Hl0 = do_amp ? (double)fabsl(params[n] / (pi * Wl0)) : fabs(params[n]);
        int st = orc_optimum_lorentzian_calc_a1etaa3(x, model, Nx, Hl0, fl0, a1, eta0, a3, asym, Wl0, 0, r0, step, trunc_c);
        if (st) return st;
        for (int l = 1; l <= 3; l++) {
            if (lmax < l) continue;
            double fl = params[foff[l] + n];
            double Wl = fabs(orc_lin_interpol(fl0_all, Wl0_all, Nfl0, fl));
            double Hl;
            if (do_amp) Hl = (double)(fabsl(params[n] / (pi * Wl)) * Vl[l]);
            else Hl = fabs(params[n] * Vl[l]);
            st = orc_optimum_lorentzian_calc_a1etaa3(x, model, Nx, Hl, fl, a1, eta0, a3, asym, Wl, l, ratios[l], step, trunc_c);
            if (st) return st;
        }
    }
    {
        const double *np = params + Nmax + lmax + Nf + Nsplit + Nwidth;
        double *npa = (double *)malloc(sizeof(double) * (size_t)(Nnoise > 0 ? Nnoise : 1));
        orc_abs_copy(np, Nnoise, npa);
        orc_harvey_like(npa, Nnoise, x, model, Nx, (Nnoise - 1) / 3);
        free(npa);
    }
    return ORC_OK;
}

/* models.cpp:3012-3195 */
int orc_model_MS_local_basic(const double *params, const int *pl, const double *x, long Nx, double *model) {
    const double step = x[1] - x[0];
    const long double pi = ORC_PI_L;
    const int Nmax = pl[0], Nvis = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8], Ninc = pl[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const double trunc_c = params[Nmax + Nvis + Nf + Nsplit + Nwidth + Nnoise + Ninc];
    const int do_amp = (params[Nmax + Nvis + Nf + Nsplit + Nwidth + Nnoise + Ninc + 1] != 0);
    double inclination = atan(params[Nmax + Nvis + Nf + 4] / params[Nmax + Nvis + Nf + 3]);
    inclination = (double)(inclination * 180. / pi);
    const double a1 = pow(params[Nmax + Nvis + Nf + 3], 2) + pow(params[Nmax + Nvis + Nf + 4], 2);
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double *ratios[4] = {r0, r1, r2, r3};
    const int Nfl[4] = {Nfl0, Nfl1, Nfl2, Nfl3};
    for (int l = 1; l <= 3; l++)
        if (Nfl[l] >= 1) orc_amplitude_ratio(l, inclination, ratios[l]);
    const double eta0 = params[Nmax + Nvis + Nf + 1];
    const double a3 = params[Nmax + Nvis + Nf + 2];
    const double asym = params[Nmax + Nvis + Nf + 5];
    for (long i = 0; i < Nx; i++) model[i] = 0.0;
    int off = 0;
    for (int l = 0; l <= 3; l++) {
        for (long n = 0; n < Nfl[l]; n++) {
            double fl = params[Nmax + Nvis + off + n];
            double Wl = fabs(params[Nmax + Nvis + Nf + Nsplit + off + n]);
            double Hl = do_amp ? (double)fabsl(params[off + n] / (pi * Wl)) : fabs(params[off + n]);
            int st = orc_optimum_lorentzian_calc_a1etaa3(x, model, Nx, Hl, fl, a1, eta0, a3, asym, Wl, l, ratios[l], step, trunc_c);
            if (st) return st;
        }
        off += Nfl[l];
    }
    {
        const double *np = params + Nmax + Nvis + Nf + Nsplit + Nwidth;
        double *npa = (double *)malloc(sizeof(double) * (size_t)(Nnoise > 0 ? Nnoise : 1));
        orc_abs_copy(np, Nnoise, npa);
        orc_harvey_like(npa, Nnoise, x, model, Nx, 0); /* Nharvey forced to 0, models.cpp:3167 */
        free(npa);
    }
    return ORC_OK;
}

/* model_def.cpp:220-388 (ids of Config/default/models_ctrl.list) */
int orc_call_model(int model_id, const double *params, const int *plength, const double *x, long Nx, double *model) {
    switch (model_id) {
    case ORC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC:
        return orc_model_MS_Global_a1etaa3_HarveyLike_Classic(params, plength, x, Nx, model);
    case ORC_MODEL_MS_LOCAL_BASIC: return orc_model_MS_local_basic(params, plength, x, Nx, model);
    case ORC_MODEL_MS_GLOBAL_AJ: return orc_model_MS_Global_aj_HarveyLike(params, plength, x, Nx, model);
    case ORC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4: return orc_model_RGB_asympt_aj_AppWidth_HarveyLike_v4(params, plength, x, Nx, model);
    case ORC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4: return orc_model_RGB_asympt_aj_CteWidth_HarveyLike_v4(params, plength, x, Nx, model);
    default: return ORC_ERR_BAD_MODEL;
    }
}

/* ------------------------------------------------------------------ */
/* batched driver (OpenMP over evaluations, as MALA.cpp:648 over chains) */
/* ------------------------------------------------------------------ */

int orc_loglike_batch(int model_id, int B, const double *params, long Nparams, const int *plength, const double *x,
                      const double *y, long Nx, double likelihood_params, const double *Tcoefs, double *logL,
                      double *model_out, int *status) {
    int first = ORC_OK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        double *m = model_out ? model_out + (size_t)b * (size_t)Nx : (double *)malloc(sizeof(double) * (size_t)Nx);
        int st = orc_call_model(model_id, params + (size_t)b * (size_t)Nparams, plength, x, Nx, m);
        if (st == ORC_OK) logL[b] = orc_call_likelihood(y, m, Nx, likelihood_params, Tcoefs ? Tcoefs[b] : 1.0);
        else logL[b] = NAN;
        if (status) status[b] = st;
        if (st) {
#pragma omp critical
            if (first == ORC_OK) first = st;
        }
        if (!model_out) free(m);
    }
    return first;
}

/* forward differences of the tempered logL over the relaxed parameters */
int orc_fd_gradient(int model_id, const double *params, long Nparams, const int *plength, const int *index_to_relax,
                    int Nvars, const double *hstep, const double *x, const double *y, long Nx, double likelihood_params,
                    double Tcoef, double *logL0, double *grad) {
    const int B = Nvars + 1;
    double *P = (double *)malloc(sizeof(double) * (size_t)B * (size_t)Nparams);
    double *T = (double *)malloc(sizeof(double) * (size_t)B);
    double *L = (double *)malloc(sizeof(double) * (size_t)B);
    for (int b = 0; b < B; b++) {
        memcpy(P + (size_t)b * (size_t)Nparams, params, sizeof(double) * (size_t)Nparams);
        T[b] = Tcoef;
        if (b > 0) P[(size_t)b * (size_t)Nparams + index_to_relax[b - 1]] += hstep[b - 1];
    }
    int st = orc_loglike_batch(model_id, B, P, Nparams, plength, x, y, Nx, likelihood_params, T, L, NULL, NULL);
    *logL0 = L[0];
    for (int k = 0; k < Nvars; k++) {
        /* the actually applied step (params+h)-params, as any FD code should use */
        volatile double xp = params[index_to_relax[k]] + hstep[k];
        double h = xp - params[index_to_relax[k]];
        grad[k] = (L[k + 1] - L[0]) / h;
    }
    free(P); free(T); free(L);
    return st;
}

/* ------------------------------------------------------------------ */
/* priors                                                              */
/* ------------------------------------------------------------------ */

long double orc_logP_uniform(double bmin_, double bmax_, double x_) { /* stats_dictionary.cpp:38-52 */
    long double b_min = bmin_, b_max = bmax_, x = x_;
    if ((x <= b_max) && (x >= b_min)) return -logl(fabsl(b_max - b_min));
    return -INFINITY;
}
long double orc_logP_uniform_abs(double bmin_, double bmax_, double x_) { /* :56-70 */
    long double b_min = bmin_, b_max = bmax_, x = x_;
    if ((fabsl(x) <= b_max) && (fabsl(x) >= b_min)) return -logl(fabsl(b_max - b_min));
    return -INFINITY;
}
long double orc_logP_gaussian(double mean_, double sigma_, double x_) { /* :98-105 */
    long double mean = mean_, sigma = sigma_, x = x_;
    return -logl(sqrtl(2 * ORC_PI_L) * sigma) - 0.5 * powl((x - mean) / sigma, 2.);
}
long double orc_logP_jeffrey(double hmin_, double hmax_, double h_) { /* :127-143 */
    long double hmin = hmin_, hmax = hmax_, h = h_;
    if (h < hmax && h > 0) {
        long double prior = 1. / (h + hmin);
        long double norm = logl((hmax + hmin) / hmin);
        return logl(prior / norm);
    }
    return -INFINITY;
}
long double orc_logP_jeffrey_abs(double hmin_, double hmax_, double h_) { /* :149-165 */
    long double hmin = hmin_, hmax = hmax_, h = h_;
    if (fabsl(h) < hmax) {
        long double prior = 1. / (fabsl(h) + hmin);
        long double norm = logl((hmax + hmin) / hmin);
        return logl(prior / norm);
    }
    return -INFINITY;
}
long double orc_logP_uniform_gaussian(double bmin_, double bmax_, double sigma_, double x_) { /* :173-195 */
    long double b_min = bmin_, b_max = bmax_, sigma = sigma_, x = x_, logP = 0;
    if (x < b_min) logP = -INFINITY;
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * powl((x - b_max) / sigma, 2.);
    long double C = logl(fabsl(b_max - b_min) + 0.5 * sqrtl(2 * ORC_PI_L) * sigma);
    return logP - C;
}
long double orc_logP_gaussian_uniform(double bmin_, double bmax_, double sigma_, double x_) { /* :200-222 */
    long double b_min = bmin_, b_max = bmax_, sigma = sigma_, x = x_, logP = 0;
    if (x > b_max) logP = -INFINITY;
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x < b_min) logP = -0.5 * powl((x - b_min) / sigma, 2.);
    long double C = logl(fabsl(b_max - b_min) + 0.5 * sqrtl(2 * ORC_PI_L) * sigma);
    return logP - C;
}
long double orc_logP_gug(double bmin_, double bmax_, double s1_, double s2_, double x_) { /* :226-250 */
    long double b_min = bmin_, b_max = bmax_, sigma1 = s1_, sigma2 = s2_, x = x_, logP = 0;
    if (x < b_min) logP = -0.5 * powl((x - b_min) / sigma1, 2.);
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * powl((x - b_max) / sigma2, 2.);
    long double C = logl(fabsl(b_max - b_min) + 0.5 * sqrtl(2 * ORC_PI_L) * (sigma1 + sigma2));
    return logP - C;
}

/* priors_calc.cpp:725-870 -- per-parameter primitive selected by id
 * (Config/default/primepriors_ctrl.list); priors is 4 x Nparams row-major.
 * ids 3, 9, 11, 12 are not restated (fatal / flagged buggy / unusable / GSL). */
long double orc_apply_generic_priors(const double *params, long i0, long n, const double *pr, long Np,
                                     const int *sw) {
    long double pena = 0;
    for (long i = i0; i < i0 + n; i++) {
        switch (sw[i]) {
        case 0: break;
        case 1: pena = pena + orc_logP_uniform(pr[i], pr[Np + i], params[i]); break;
        case 2: pena = pena + orc_logP_gaussian(pr[i], pr[Np + i], params[i]); break;
        case 4: pena = pena + orc_logP_jeffrey(pr[i], pr[Np + i], params[i]); break;
        case 5: pena = pena + orc_logP_uniform_gaussian(pr[i], pr[Np + i], pr[2 * Np + i], params[i]); break;
        case 6: pena = pena + orc_logP_gaussian_uniform(pr[i], pr[Np + i], pr[2 * Np + i], params[i]); break;
        case 7: pena = pena + orc_logP_gug(pr[i], pr[Np + i], pr[2 * Np + i], pr[3 * Np + i], params[i]); break;
        case 8: pena = pena + orc_logP_uniform_abs(pr[i], pr[Np + i], params[i]); break;
        case 10: pena = pena + orc_logP_jeffrey_abs(pr[i], pr[Np + i], params[i]); break;
        case 13: break;
        default: return NAN;
        }
    }
    return pena;
}

/* second differences with replicated edges: Scndder_adaptive_reggrid(y) (derivatives_handler.cpp:400-426,
 * forward :248-268 at the first point, backward at the last, centred :294-314 in between; unit spacing) */
static double orc_second_difference(const double *y, long n, long i) {
    if (n < 3) return 0.0;
    if (i == 0) return y[2] - 2. * y[1] + y[0];
    if (i == n - 1) return y[n - 1] - 2. * y[n - 2] + y[n - 3];
    return y[i + 1] - 2. * y[i] + y[i - 1];
}

/* priors_calc.cpp:27-317 -- model class io_MS_Global.  Restated for the model families this build ships:
 * model_index 9 (aj family, :206-228) and the default branch (Classic, :230-262 with impose_normHnlm = 0).
 * Other families return NaN (the reference exits with "needs checks" for most of them). */
long double orc_priors_MS_Global(const double *params, const int *pl, const double *pr, const int *sw, const double *extra) {
    long double f = 0;
    const int smooth_switch = (int)extra[0];
    const double scoef = extra[1];
    const double *ajova1_limit = &extra[2];
    const int impose_normHnlm = (int)extra[8];
    const int model_index = (int)extra[9];
    const int Nmax = pl[0], lmax = pl[1];
    const int Nfl[4] = {pl[2], pl[3], pl[4], pl[5]};
    const int Nsplit = pl[6], Nwidth = pl[7];
    const int Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
    long Np = 0;
    for (int i = 0; i < 11; i++) Np += pl[i];
    for (int i = Nmax; i <= Nmax + lmax; i++)
        if (params[i] < 0) return -INFINITY;
    if (model_index == 9) {
        int i0 = Nfl[0];
        for (int el = 1; el < lmax + 1; el++) {
            for (int j = 1; j < 6; j++)
                for (int n = 0; n < Nfl[el]; n++) {
                    double fl = params[Nmax + lmax + i0 + n];
                    double a1 = params[Nmax + lmax + Nf] + params[Nmax + lmax + Nf + 1] * (fl * 1e-3);
                    double aj = params[Nmax + lmax + Nf + 2 * j] + params[Nmax + lmax + Nf + 2 * j + 1] * (fl * 1e-3);
                    if (fabs(aj / a1) >= ajova1_limit[j]) return -INFINITY;
                    if (a1 < 0) return -INFINITY;
                }
            i0 = i0 + Nfl[el];
        }
    } else if (model_index >= 0 && model_index <= 8) {
        return NAN;
    } else if (impose_normHnlm != 0) {
        return NAN;
    }
    const int on = Nmax + lmax + Nf + Nsplit + Nwidth;
    if (sw[on + 3] != 0)
        if ((params[on + 3] < 0) || (params[on + 4] < 0) || (params[on + 5] < 0)) return -INFINITY;
    if (sw[on + 6] != 0)
        if ((params[on + 6] < 0) || (params[on + 7] < 0) || (params[on + 8] < 0)) return -INFINITY;
    if ((sw[Nmax + lmax + Nf + 9] != 0) && (params[on + 9] < 0)) return -INFINITY; /* index quirk of :272 kept */
    f = f + orc_apply_generic_priors(params, 0, Np, pr, Np, sw);
    double *idx = (double *)malloc(sizeof(double) * (size_t)(Nfl[0] > 0 ? Nfl[0] : 1));
    double fit[2];
    for (int i = 0; i < Nfl[0]; i++) idx[i] = (double)i;
    orc_linfit(idx, params + Nmax + lmax, Nfl[0], fit);
    free(idx);
    const double Dnu = fit[0];
    if (Nfl[0] == Nfl[2])
        for (int i = 0; i < Nfl[0]; i++) {
            double d02 = params[Nmax + lmax + i] - params[Nmax + lmax + Nfl[0] + Nfl[1] + i];
            f = f + orc_logP_gaussian_uniform(0, Dnu / 3., 0.015 * Dnu, d02);
        }
    if (smooth_switch == 1) {
        int i0 = 0;
        for (int el = 0; el < lmax + 1; el++) {
            if (Nfl[el] != 0)
                for (int i = 0; i < Nfl[el]; i++)
                    f = f + orc_logP_gaussian(0, scoef, orc_second_difference(params + Nmax + lmax + i0, Nfl[el], i));
            i0 = i0 + Nfl[el];
        }
    }
    return f;
}

/* priors_calc.cpp:514-629 -- model class io_local */
long double orc_priors_local(const double *params, const int *pl, const double *pr, const int *sw, const double *extra) {
    long double f = 0;
    const double a3ova1_limit = extra[2];
    const int Nmax = pl[0], Nvis = pl[1];
    const int Nf = pl[2] + pl[3] + pl[4] + pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8];
    long Np = 0;
    for (int i = 0; i < 11; i++) Np += pl[i];
    const int o = Nmax + Nvis + Nf;
    if (params[o] != 0) {
        if (fabs(params[o + 2] / params[o]) >= a3ova1_limit) return -INFINITY;
    } else if ((params[o + 3] != 0) && (params[o + 4] != 0)) {
        if (fabs(params[o + 2] / (pow(params[o + 3], 2) + pow(params[o + 4], 2))) >= a3ova1_limit) return -INFINITY;
    }
    const int oi = o + Nsplit + Nwidth + Nnoise;
    if ((sw[oi] != 0) && (params[oi] < 0)) return -INFINITY;
    f = f + orc_apply_generic_priors(params, 0, Np, pr, Np, sw);
    return f;
}

/* priors_calc.cpp:319-512 -- model class io_asymptotic (red giants), model_switch 3 = the v4 models; the v3 branch
 * (model_switch 1, l=1 p modes listed in the parameter vector) is not restated: NAN. */
long double orc_priors_asymptotic(const double *params, const int *pl, const double *pr, const int *sw, const double *extra) {
    long double f = 0;
    const double scoef = extra[1], a3ova1_limit = extra[2];
    const int model_switch = (int)extra[4];
    const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5], Nsplit = pl[6], Nwidth = pl[7];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    long Np = 0;
    for (int k = 0; k < 11; k++) Np += pl[k];
    int i;
    const int i0 = Nmax + lmax + Nf + Nsplit;
    const double a3 = params[Nmax + lmax + Nf + 4], rot_env = fabs(params[Nmax + lmax + Nf]);
    for (i = Nmax; i <= Nmax + lmax; i++)
        if (params[i] < 0) return -INFINITY;
    if (fabs(a3 / rot_env) >= a3ova1_limit) return -INFINITY;
    const int on = Nmax + lmax + Nf + Nsplit + Nwidth;
    if (sw[on + 3] != 0 && ((params[on + 3] < 0) || (params[on + 4] < 0) || (params[on + 5] < 0))) return -INFINITY;
    if (sw[on + 6] != 0 && ((params[on + 6] < 0) || (params[on + 7] < 0) || (params[on + 8] < 0))) return -INFINITY;
    if ((sw[Nmax + lmax + Nf + 9] != 0) && (params[on + 9] < 0)) return -INFINITY;
    for (i = i0; i < i0 + Nwidth; i++)
        if (sw[i] == 2 && params[i] < 0) return -INFINITY;
    f = f + orc_apply_generic_priors(params, 0, Np, pr, Np, sw);
    if (model_switch == 1) return NAN;
    if (model_switch == 3) {
        if (params[Nmax + lmax + Nfl0 + 6] < 0) return -INFINITY;
        if (params[Nmax + lmax + Nfl0 + 7] < 0) return -INFINITY;
    }
    /* :466 -- the switch reads priors_names_switch[i] with i left at i0+Nwidth by the loop above */
    if (sw[i] == 1) {
        for (int k = 0; k < Nfl0; k++) f = f + orc_logP_gaussian(0, scoef, orc_second_difference(params + Nmax + lmax, Nfl0, k));
        for (int k = 0; k < Nfl3; k++)
            f = f + orc_logP_gaussian(0, scoef, orc_second_difference(params + Nmax + lmax + Nfl0 + Nfl1 + Nfl2, Nfl3, k));
    }
    return f;
}

/* call_prior (model_def.cpp:421-464): class 2 = io_MS_Global, 3 = io_local, 4 = io_asymptotic */
double orc_call_prior(int prior_class, const double *params, const int *pl, const double *pr, const int *sw, const double *extra) {
    if (prior_class == 2) return (double)orc_priors_MS_Global(params, pl, pr, sw, extra);
    if (prior_class == 3) return (double)orc_priors_local(params, pl, pr, sw, extra);
    if (prior_class == 4) return (double)orc_priors_asymptotic(params, pl, pr, sw, extra);
    return NAN;
}

/* ---- evidence diagnostic: interpol.cpp:46-101 (quad_interpol, interp1, parabola, interp2), diagnostics.cpp:980-1019 ---- */
static long double orc_interp1(long double x, const double *a, long n) { /* interpol.cpp:64-72 */
    if (x <= 0) return a[0];
    if (x >= n - 1) return a[n - 1];
    int j = (int)x;
    return a[j] + (x - j) * (a[j + 1] - a[j]);
}
static long double orc_parabola(long double x, long double f_1, long double f0, long double f1) { /* :86-92 */
    if (x <= -1) return f_1;
    if (x >= 1) return f1;
    long double l = f0 - x * (f_1 - f0);
    long double r = f0 + x * (f1 - f0);
    return (l + r + x * (r - l)) / 2;
}
static long double orc_interp2(long double x, const double *a, long n) { /* :95-101 */
    if (x <= .5 || x >= n - 1.5) return orc_interp1(x, a, n);
    int j = (int)(x + .5);
    long double t = 2 * (x - j);
    return orc_parabola(t, (a[j - 1] + a[j]) / 2, a[j], (a[j] + a[j + 1]) / 2);
}
void orc_quad_interpol(const double *a, long n, long m, double *b) { /* :46-59 */
    long double step = (double)(n - 1) / (m - 1);
    for (long j = 0; j < m; j++) b[j] = (double)orc_interp2(j * step, a, n);
}
/* Likelihoods: [rows x Nchains] row-major (first Nchains columns of the stat_criteria table, diagnostics.cpp:224) */
double orc_evidence_calc(const double *Tcoefs, long Nchains, const double *Likelihoods, long rows, int interpol_factor, double *beta,
                         double *L_beta, double *beta_interp, double *L_beta_interp) {
    const long Npts = interpol_factor * Nchains;
    for (long i = 0; i < Nchains; i++) {
        beta[i] = 1. / Tcoefs[i];
        double s = 0;
        for (long r = 0; r < rows; r++) s += Likelihoods[r * Nchains + i];
        L_beta[i] = s / rows;
    }
    orc_quad_interpol(beta, Nchains, Npts, beta_interp);
    orc_quad_interpol(L_beta, Nchains, Npts, L_beta_interp);
    double tot = 0;
    for (long j = 0; j < Npts; j++) tot += L_beta_interp[j];
    return tot / Npts;
}
