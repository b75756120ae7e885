// mode_tables_impl.h -- the scalar half of the model functions, written once for host and device.
//
// Host build (mode_tables.cpp, g++/clang x86): `xreal` = long double, i.e. the expression types of the reference
// (Pslm products, pi used for amplitudes) -> tables bit-identical to the CPU restatement.
// Device build (dev_sampler.hip, gfx950): `xreal` = double (no 80-bit type on the GPU) -> nu_nlm / H*V differ from the
// host table by <= 1-2 ulp; window indices are pure double arithmetic and agree bit for bit.
//
// Structure: per parameter vector a small set of SHARED scalars (visibilities, eta0, asymmetry, ...) and then one
// independent multiplet per index -- the host loops over the index, the device gives each index its own thread.
//
//   model_MS_Global_aj_HarveyLike               tamcmc/sources/models.cpp:1195-1408   (id 23)
//   model_MS_Global_a1etaa3_HarveyLike_Classic  tamcmc/sources/models.cpp:1943-2121   (id 3)
//   model_MS_local_basic                        tamcmc/sources/models.cpp:3012-3195   (id 11)
#pragma once
#include <math.h>
#include <limits.h>
#include <stdint.h>

#include "../../include/tamcmc_hip.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TM_HD __host__ __device__ inline
#else
#define TM_HD inline
#endif

namespace tamcmc {
namespace mt {

#if defined(__HIP_DEVICE_COMPILE__)
typedef double xreal;
#else
typedef long double xreal;
#endif

// ---------- polynomial tables (filled once; see mode_tables.cpp / dev_sampler.hip) ----------
struct PolyTab {
    double Q[4][7];       // Qlm(l, m+3)
    xreal P[7][4][7];     // Pslm(s, l, m+3)
};

// acoefs.cpp:19-49
TM_HD xreal ritzwoller_H(int s, int l, int m) {
    const int L = l * (l + 1);
    const double dm = m;
    xreal H = 0;
    if (s == 5) H = 252 * pow(dm, 5) - 140 * (2 * L - 3) * pow(dm, 3) + (20 * L * (3 * L - 10) + 48) * m;
    if (s == 6)
        H = 924 * pow(dm, 6) - 420 * pow(dm, 4) * (3 * L - 7) + 84 * pow(dm, 2) * (5 * pow((double)L, 2) - 25 * L + 14) -
            20 * L * (pow((double)L, 2) - 8 * L + 12);
    return H;
}
// acoefs.cpp:51-110
TM_HD xreal Pslm_compute(int s, int l, int m) {
    const double dm = m, dl = l;
    xreal Ps = 0;
    if (s == 0) Ps = l;
    if (s == 1) Ps = m;
    if (s == 2 && l > 0) {
        const double v = (3 * pow(dm, 2) - l * (l + 1)) / (2 * l - 1);
        Ps = v;
    }
    if (s == 3 && l > 1) {
        const double v = (5 * pow(dm, 3) - (3 * l * (l + 1) - 1) * m) / ((l - 1) * (2 * l - 1));
        Ps = v;
    }
    if (s == 4) {
        const double h = (35 * pow(dm, 4) - 5 * (6 * l * (l + 1) - 5) * pow(dm, 2)) + 3 * l * (l + 1) * (l * (l + 1) - 2);
        const xreal H = h, c = 2 * (l - 1) * (2 * l - 1) * (2 * l - 3);
        if (c != 0) Ps = H / c;
    }
    if (s == 5) {
        const double cd = 8 * (4 * pow(dl, 4) - 20 * pow(dl, 3) + 35 * pow(dl, 2) - 25 * l + 6);
        const xreal H = ritzwoller_H(5, l, m), c = cd;
        if (c != 0) Ps = H / c;
    }
    if (s == 6) {
        const double cd = 64 * pow(dl, 5) - 480 * pow(dl, 4) + 1360 * pow(dl, 3) - 1800 * pow(dl, 2) + 1096 * l - 240;
        const xreal H = ritzwoller_H(6, l, m), c = cd;
        if (c != 0) Ps = H / c;
    }
    return Ps;
}
// build_lorentzian.cpp:583-592
TM_HD double Qlm_compute(int l, int m) {
    const xreal Dnl = 2. / 3;
    double q = (l * (l + 1) - 3 * pow((double)m, 2)) / ((2 * l - 1) * (2 * l + 3));
    q = (double)(q * Dnl);
    return q;
}
TM_HD void fill_poly(PolyTab &t) {
    for (int s = 0; s <= 6; s++)
        for (int l = 0; l <= 3; l++)
            for (int m = -3; m <= 3; m++) t.P[s][l][m + 3] = Pslm_compute(s, l, m);
    for (int l = 0; l <= 3; l++)
        for (int m = -3; m <= 3; m++) t.Q[l][m + 3] = Qlm_compute(l, m);
}

// ---------- m-visibilities: function_rot.cpp:15-101 ----------
TM_HD int ifact(int n) {  // function_rot.cpp:94-101 (n <= 6 on this path: table; same values as the loop)
    if (n <= 1) return 1;
    switch (n) {
    case 2: return 2;
    case 3: return 6;
    case 4: return 24;
    case 5: return 120;
    case 6: return 720;
    default: break;
    }
    long f = 1;
    for (long i = 1; i <= n; i++) f *= i;
    return (int)f;
}
TM_HD double icombi(int n, int r) {
#if defined(__HIP_DEVICE_COMPILE__)
    // n <= 6 on this path: Pascal's triangle packed one row per 64-bit constant (a byte per entry) -- the same integers as the
    // factorial quotients below without three integer divisions per call
    if (n >= 0 && n <= 6 && r >= 0 && r <= n) {
        const unsigned long long row = n == 0 ? 0x01ull : n == 1 ? 0x0101ull : n == 2 ? 0x010201ull : n == 3 ? 0x01030301ull
                                     : n == 4 ? 0x0104060401ull : n == 5 ? 0x01050a0a0501ull : 0x01060f140f0601ull;
        return (double)(int)((row >> (8 * r)) & 0xff);
    }
#endif
    return (double)(ifact(n) / ifact(n - r) / ifact(r));
}
// (-1)^n and x^n for small integer n.  Host: libm pow(), exactly what the reference calls.  Device: sign flip /
// repeated multiplication (pow() on the GPU costs ~150 fp64 issue slots and serialises a lone lane for microseconds);
// (-1)^n is exact either way, x^n agrees with pow() to <= 3 ulp for n <= 6.
TM_HD double pow_m1(int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (n & 1) ? -1.0 : 1.0;
#else
    return pow(-1.0, (double)n);
#endif
}
// x^2.  Host: libm pow(x, 2.), what the reference calls (correctly rounded: the same value as x*x).  Device: the product -- pow() there is
// ~150 fp64 issue slots, half a microsecond for a lone lane, and this sits in the per-m loop of every multiplet.
TM_HD double pow_2(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return x * x;
#else
    return pow(x, 2.);
#endif
}
TM_HD double pow_int(double x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = 1.0;
    for (int i = 0; i < n; i++) r = r * x;
    return r;
#else
    return pow(x, (double)n);
#endif
}
// one term of the sum over s in dmm() and the normalisation applied after the loop (function_rot.cpp:76-88)
TM_HD double wigner_term(int l, int m1, int m2, double beta, long s) {
    double v = icombi(l + m2, (int)(l - m1 - s)) * icombi(l - m2, (int)s) * pow_m1((int)(l - m1 - s));
    v = v * pow_int(cos(beta / 2.), (int)(2 * s + m1 + m2)) * pow_int(sin(beta / 2.), (int)(2 * l - 2 * s - m1 - m2));
    return v;
}
TM_HD double wigner_finish(int l, int m1, int m2, double sum) {
    sum = sum * sqrt((double)(ifact(l + m1) * ifact(l - m1)));
    sum = sum / sqrt((double)(ifact(l + m2) * ifact(l - m2)));
    return sum;
}
TM_HD double wigner_d(int l, int m1, int m2, double beta) {
    double sum = 0;
    for (long s = 0; s <= l - m1; s++) sum = sum + wigner_term(l, m1, m2, beta, s);
    return wigner_finish(l, m1, m2, sum);
}
TM_HD void amplitude_ratio(int l, double beta_deg, double *V) {
    const double PI = 3.141592653589793238462643;
    const double ang = PI * beta_deg / 180.;
    // centre column (m'=0) of the rotation matrix as the four fill passes of function_rot() leave it
    for (int i = 0; i <= l; i++) V[l + i] = wigner_d(l, i, 0, ang);
    for (int i = -l; i <= 0; i++) V[l + i] = V[l - i] * pow_m1(i);
    V[l] = wigner_d(l, 0, 0, -ang);
    V[l] = V[l] * pow_m1(0);
    for (int i = 0; i <= 2 * l; i++) V[i] = V[i] * V[i];
}

// ---------- interpol.cpp:13-43, linfit.cpp:17-35, models.cpp:6065-6084 ----------
// segment of the abscissa grid used by lin_interpol for xi: -1 none (NaN), else the left index of the segment
TM_HD long lin_segment(const double *x, long n, double xi) {
    long seg = -1;
#if defined(__HIP_DEVICE_COMPILE__)
    if (n >= 2 && n <= 16) {  // the same scan on a register copy of the grid: one batch of loads instead of two dependent ones per step
        double g[16];
#pragma unroll
        for (int k = 0; k < 16; k++) g[k] = x[k < n ? k : n - 1];
        const double xl = x[n - 1];
        if (xi >= g[0] && xi <= xl) {
            int i = 0;
            bool go = true;
#pragma unroll
            for (int k = 0; k < 14; k++) {
                const bool adv = go && k < n - 2 && (xi < g[k] || xi > g[k + 1]);
                i = adv ? k + 1 : i;
                go = adv;
            }
            seg = i;
        }
        if (xi < g[0]) seg = 0;
        if (xi > xl) seg = n - 2;
        return seg;
    }
#endif
    if (xi >= x[0] && xi <= x[n - 1]) {
        long i = 0;
        while (i < n - 2 && (xi < x[i] || xi > x[i + 1])) ++i;
        seg = i;
    }
    if (xi < x[0]) seg = 0;
    if (xi > x[n - 1]) seg = n - 2;
    return seg;
}
TM_HD double lin_interpol_seg(const double *x, const double *y, long seg, double xi) {
    double a = 0, b = 0;
    if (seg >= 0) {
        a = (y[seg + 1] - y[seg]) / (x[seg + 1] - x[seg]);  // slope
        b = y[seg] - a * x[seg];                            // ordinate at origin
    }
    return a * xi + b;
}
TM_HD double lin_interpol(const double *x, const double *y, long n, double xi) {
    return lin_interpol_seg(x, y, lin_segment(x, n, xi), xi);
}
// slope/intercept of y against the index 0..n-1 (linfit with x = LinSpaced(n, 0, n-1))
TM_HD void linfit_index(const double *y, long n, double out[2]) {
    double sx = 0, sy = 0, sty = 0, stt = 0;
    for (long i = 0; i < n; i++) sx += (double)i;
    for (long i = 0; i < n; i++) sy += y[i];
    const double dn = (double)n, mx = sx / dn;
    for (long i = 0; i < n; i++) sty += ((double)i - mx) * y[i];
    for (long i = 0; i < n; i++) stt += ((double)i - mx) * ((double)i - mx);
    out[0] = sty / stt;
    out[1] = (sy - sx * out[0]) / dn;
}
TM_HD double eta0_from_dnu(double dnu) {
    const double G = 6.667e-8, Dnu_sun = 135.1, R_sun = 6.96342e5, M_sun = 1.98855e30;
#if defined(__HIP_DEVICE_COMPILE__)
    const double R3 = 0x1.0a5c08c31239dp+108;  // pow(R_sun * 1e5, 3) as libm returns it
#else
    const double R3 = pow(R_sun * 1e5, 3);
#endif
    const double rho_sun = M_sun * 1e3 / (4 * 3.14159265358979323846 * R3 / 3);
    const double rho = pow_2(dnu / Dnu_sun) * rho_sun;
    return 3. * 3.14159265358979323846 / (rho * G);
}
TM_HD double eta0_fct(const double *fl0, long n) {
    double r[2];
    linfit_index(fl0, n, r);
    return eta0_from_dnu(r[0]);
}

// ---------- truncation window: build_lorentzian.cpp:595-676 ----------
TM_HD int to_int_sat(double v) {
    if (v >= (double)INT_MAX) return INT_MAX;
    if (v <= (double)INT_MIN) return INT_MIN;
    return (int)v;
}
TM_HD int set_imin_imax(double x_first, double x_last, int64_t Nx, int l, double fc, double gamma, double f_s, double c,
                        double step, int *i0, int *i1) {
    double lo = 0, hi = 0;
    bool have = false;
    // the four overlapping regimes, later ones overriding earlier ones as in the reference
    if (gamma >= 1 && f_s >= 1) { const double h = (l != 0) ? c * (l * f_s + gamma) : c * gamma * 2.2; lo = fc - h; hi = fc + h; have = true; }
    if (gamma <= 1 && f_s >= 1) { const double h = (l != 0) ? c * (l * f_s + 1) : c * 2.2; lo = fc - h; hi = fc + h; have = true; }
    if (gamma >= 1 && f_s <= 1) { const double h = (l != 0) ? c * (l + gamma) : c * 2.2 * gamma; lo = fc - h; hi = fc + h; have = true; }
    if (gamma <= 1 && f_s <= 1) { const double h = (l != 0) ? c * (l + 1) : c * 2.2; lo = fc - h; hi = fc + h; have = true; }
    if (!have) return TAMCMC_ERR_NAN_WINDOW;
    if ((hi - step) < x_first) hi = x_first + c;
    if ((lo + step) >= x_last) lo = x_last - c;
    int a = to_int_sat(floor((lo - x_first) / step));
    int b = to_int_sat(ceil((hi - x_first) / step));
    if (a < 0) a = 0;
    if (b > Nx) b = (int)Nx;
    if (b - a <= 0) return TAMCMC_ERR_EMPTY_WINDOW;
    *i0 = a;
    *i1 = b;
    return TAMCMC_OK;
}

// ---------- split frequencies ----------
// build_lorentzian.cpp:226-229 (sum evaluated in xreal because Pslm is long double in the reference)
TM_HD double nu_nlm_aj(const PolyTab &T, double fc, const double a[7], double eta0, int l, int m) {
    xreal acc = fc + a[1] * T.P[1][l][m + 3] + a[2] * T.P[2][l][m + 3] + a[3] * T.P[3][l][m + 3] + a[4] * T.P[4][l][m + 3] +
                a[5] * T.P[5][l][m + 3] + a[6] * T.P[6][l][m + 3];
    double nu = (double)acc;
    if (eta0 > 0) nu = nu + fc * eta0 * T.Q[l][m + 3] * pow_2(a[1] * 1e-6);
    return nu;
}
// build_lorentzian.cpp:145
TM_HD double nu_nlm_a1etaa3(const PolyTab &T, double fc, double f_s, double eta0, double a3, int l, int m) {
    const double t = fc * (1. + eta0 * pow_2(f_s * 1e-6) * T.Q[l][m + 3]) + m * f_s;
    const xreal acc = t + T.P[3][l][m + 3] * a3;
    return (double)acc;
}

// ---------- params_length decoding shared by the three models (models.cpp:1207-1219) ----------
struct Layout {
    int Nmax, lmax, Nfl[4], Nsplit, Nwidth, Nnoise, Ninc, Nf;
    int o_vis, o_f[4], o_split, o_width, o_noise, o_inc, o_cfg;
};
TM_HD Layout make_layout(const int32_t *pl) {
    Layout L;
    L.Nmax = pl[0]; L.lmax = pl[1];
    for (int i = 0; i < 4; i++) L.Nfl[i] = pl[2 + i];
    L.Nsplit = pl[6]; L.Nwidth = pl[7]; L.Nnoise = pl[8]; L.Ninc = pl[9];
    L.Nf = L.Nfl[0] + L.Nfl[1] + L.Nfl[2] + L.Nfl[3];
    L.o_vis = L.Nmax;
    L.o_f[0] = L.Nmax + L.lmax;
    for (int i = 1; i < 4; i++) L.o_f[i] = L.o_f[i - 1] + L.Nfl[i - 1];
    L.o_split = L.Nmax + L.lmax + L.Nf;
    L.o_width = L.o_split + L.Nsplit;
    L.o_noise = L.o_width + L.Nwidth;
    L.o_inc = L.o_noise + L.Nnoise;
    L.o_cfg = L.o_inc + L.Ninc;
    return L;
}

// number of multiplets a parameter vector of this layout produces (-1: unknown model)
TM_HD int count_multiplets(int model_id, const int32_t *pl) {
    switch (model_id) {
    case TAMCMC_MODEL_MS_GLOBAL_AJ:
    case TAMCMC_MODEL_MS_LOCAL_BASIC: return pl[2] + pl[3] + pl[4] + pl[5];
    case TAMCMC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC: {
        const int lm = pl[1] < 3 ? pl[1] : 3;
        return pl[0] * (1 + (lm > 0 ? lm : 0));
    }
    default: return -1;
    }
}

// ---------- shared scalars of one parameter vector ----------
struct Shared {
    Layout L;
    double ratios[4][7];  // m-visibilities per degree
    double Vl[4];         // |visibility| per degree
    double eta0, asym, trunc_c, a1, a3, inc;
    double centre[4];     // device: d^l_{0,0}(-beta) before the final overwrite of the centre element
    int need_ratio[4];    // which degrees need amplitude_ratio()
    int do_amp;
    int nharvey;
};

// everything but the m-visibilities (the device computes those in parallel, one Wigner element per lane)
TM_HD void shared_scalars_base(int model_id, const double *p, const int32_t *pl, Shared &S) {
    S.L = make_layout(pl);
    const Layout &L = S.L;
    S.trunc_c = p[L.o_cfg];
    S.do_amp = (p[L.o_cfg + 1] != 0) ? 1 : 0;
    for (int l = 0; l < 4; l++) {
        S.Vl[l] = (l == 0) ? 1.0 : 0.0;
        for (int m = 0; m < 7; m++) S.ratios[l][m] = 0.0;
    }
    S.ratios[0][0] = 1.0;
    S.a1 = 0; S.a3 = 0;
    for (int l = 0; l < 4; l++) { S.need_ratio[l] = 0; S.centre[l] = 0.0; }
    if (model_id == TAMCMC_MODEL_MS_LOCAL_BASIC) {  // models.cpp:3059-3088
        const xreal pi = 3.141592653589793238462643383279502884L;
        double inc = atan(p[L.o_split + 4] / p[L.o_split + 3]);
        inc = (double)(inc * 180. / pi);
        S.inc = inc;
        S.a1 = pow_2(p[L.o_split + 3]) + pow_2(p[L.o_split + 4]);
        for (int l = 1; l <= 3; l++)
            if (L.Nfl[l] >= 1) S.need_ratio[l] = 1;
        S.eta0 = p[L.o_split + 1];
        S.a3 = p[L.o_split + 2];
        S.asym = p[L.o_split + 5];
        S.nharvey = 0;  // models.cpp:3167
        return;
    }
    S.inc = p[L.o_inc];
    for (int l = 1; l <= 3; l++)
        if (L.lmax >= l) {
            S.Vl[l] = fabs(p[L.o_vis + l - 1]);
            S.need_ratio[l] = 1;
        }
    S.nharvey = (L.Nnoise - 1) / 3;
    if (model_id == TAMCMC_MODEL_MS_GLOBAL_AJ) {  // models.cpp:1264-1276
        const double *sp = p + L.o_split;
        S.asym = sp[13];
        S.eta0 = (sp[12] == 1) ? eta0_fct(p + L.o_f[0], L.Nfl[0]) : 0.0;
    } else {  // Classic, models.cpp:2013-2017
        S.a1 = fabs(p[L.o_split]);
        S.eta0 = eta0_fct(p + L.o_f[0], L.Nfl[0]);
        S.a3 = p[L.o_split + 2];
        S.asym = p[L.o_split + 5];
    }
}

TM_HD void shared_scalars(int model_id, const double *p, const int32_t *pl, Shared &S) {
    shared_scalars_base(model_id, p, pl, S);
    for (int l = 1; l <= 3; l++)
        if (S.need_ratio[l]) amplitude_ratio(l, S.inc, S.ratios[l]);
}

TM_HD xreal xabs(xreal v) { return v < 0 ? -v : v; }

// ---------- one multiplet by index (reference accumulation order) ----------
// aj: l-major (all l=0, then l=1, ...) models.cpp:1288-1376; Classic: n-major (l=0..lmax per order) :2026-2085;
// local: l-major :3096-3159.
// defer_ratio: leave hv[m] = H (the caller multiplies by the m-visibilities once they are known: same product, same bits)
#if defined(TAMCMC_PROBE) && defined(__HIP_DEVICE_COMPILE__)
#define BMSTAMP(k) do { if (pst && index == 20) pst[k] = (long)wall_clock64(); } while (0)
#else
#define BMSTAMP(k)
#endif
TM_HD int build_multiplet(int model_id, const PolyTab &T, const double *p, const Shared &S, int index, double x_first,
                          double x_last, int64_t Nx, double step, tamcmc_multiplet *r, bool defer_ratio = false, long *pst = nullptr) {
    const Layout &L = S.L;
    BMSTAMP(4);
    int l = 0, n = 0;
    if (model_id == TAMCMC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC) {
        const int per_n = 1 + (L.lmax < 3 ? (L.lmax > 0 ? L.lmax : 0) : 3);
        n = index / per_n;
        l = index % per_n;
    } else {
        int rem = index;
        while (l < 3 && rem >= L.Nfl[l]) { rem -= L.Nfl[l]; l++; }
        n = rem;
    }
    double H, W, f;
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    double f_s;   // what set_imin_imax receives as splitting
    double eta0 = S.eta0;
    if (model_id == TAMCMC_MODEL_MS_GLOBAL_AJ) {
        const xreal pi = 3.14159265358979323846;  // M_PI widened, models.cpp:1205
        const double *fl0 = p + L.o_f[0], *Wl0 = p + L.o_width, *Hl0 = p;
        f = p[L.o_f[l] + n];
        if (l == 0) {
            W = fabs(Wl0[n]);
            if (S.do_amp) H = (double)xabs(p[n] / (pi * W));
            else H = fabs(p[n]);
            eta0 = 0.0;  // the l=0 call passes eta0 = 0 (models.cpp:1296)
        } else {
            const double *sp = p + L.o_split;
            const long seg = lin_segment(fl0, L.Nfl[0], f);  // widths and heights share the abscissa search
            W = fabs(lin_interpol_seg(fl0, Wl0, seg, f));
            if (S.do_amp) H = (double)xabs(lin_interpol_seg(fl0, Hl0, seg, f) / (pi * W) * S.Vl[l]);
            else H = fabs(lin_interpol_seg(fl0, Hl0, seg, f) * S.Vl[l]);
            // a_j(nu) = aj0 + aj1 nu[mHz] for j <= 2l (models.cpp:1314-1367); constant trip count keeps a[] in registers
            for (int j = 1; j <= 6; j++) a[j] = (j <= 2 * l) ? sp[2 * (j - 1)] + sp[2 * (j - 1) + 1] * (f * 1e-3) : 0.0;
        }
        f_s = a[1];
    } else if (model_id == TAMCMC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC) {
        const xreal pi = 3.141592653589793238462643383279502884L;
        const double *fl0 = p + L.o_f[0], *Wl0 = p + L.o_width;
        if (l == 0) {
            f = fl0[n];
            W = fabs(Wl0[n]);
            if (S.do_amp) H = (double)xabs(p[n] / (pi * W));
            else H = fabs(p[n]);
        } else {
            f = p[L.o_f[l] + n];
            W = fabs(lin_interpol(fl0, Wl0, L.Nfl[0], f));
            if (S.do_amp) H = (double)(xabs(p[n] / (pi * W)) * S.Vl[l]);
            else H = fabs(p[n] * S.Vl[l]);
        }
        f_s = S.a1;
    } else {  // local basic
        const xreal pi = 3.141592653589793238462643383279502884L;
        int off = 0;
        for (int k = 0; k < l; k++) off += L.Nfl[k];
        f = p[L.o_f[0] + off + n];
        W = fabs(p[L.o_width + off + n]);
        if (S.do_amp) H = (double)xabs(p[off + n] / (pi * W));
        else H = fabs(p[off + n]);
        f_s = S.a1;
    }
    int i0 = 0, i1 = 0;
    BMSTAMP(5);
    const int st = set_imin_imax(x_first, x_last, Nx, l, f, W, f_s, S.trunc_c, step, &i0, &i1);
    BMSTAMP(6);
    if (st) return st;
    r->l = l; r->i0 = i0; r->i1 = i1; r->flags = 0;
    r->fc = f; r->gamma = W; r->asym = S.asym;
    // the 2l+1 components, unused entries zero (a constant trip count: the table reads of all seven go out together)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 7; k++) {
        const int m = k - l;
        double nu = 0.0, hv = 0.0;
        if (k <= 2 * l) {
            nu = f;
            if (l != 0) {
                if (model_id == TAMCMC_MODEL_MS_GLOBAL_AJ) nu = nu_nlm_aj(T, f, a, eta0, l, m);
                else nu = nu_nlm_a1etaa3(T, f, f_s, eta0, S.a3, l, m);
            }
            hv = defer_ratio ? H : H * S.ratios[l][k];
        }
        r->nu[k] = nu;
        r->hv[k] = hv;
    }
    BMSTAMP(7);
    return TAMCMC_OK;
}

}  // namespace mt
}  // namespace tamcmc
