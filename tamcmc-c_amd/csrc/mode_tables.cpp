// mode_tables.cpp -- host-side scalar half of the model functions (product code, C++).
//
// Every Lorentzian model of the reference's dispatch table (model_def.cpp:220-388) unpacks `params`
// by `params_length` into per-mode scalars and then calls optimum_lorentzian_calc_* per (n,l) multiplet
// (SURVEY App. D).  This file performs exactly that scalar part -- parameter unpack, m-visibilities,
// width/height interpolation, a-coefficients -> nu_nlm, eta0, truncation window -- and emits the flat
// multiplet table the device kernel consumes.  The per-bin work is NOT done here.
//
//   model_MS_Global_aj_HarveyLike               tamcmc/sources/models.cpp:1195-1408   (id 23)
//   model_MS_Global_a1etaa3_HarveyLike_Classic  tamcmc/sources/models.cpp:1943-2121   (id 3)
//   model_MS_local_basic                        tamcmc/sources/models.cpp:3012-3195   (id 11)
//
// Expression types follow the reference (long double where its expressions are long double) so that
// the table is bit-identical to what the reference would feed its own per-bin loops.
// Must be compiled without FMA contraction (-ffp-contract=off).
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/tamcmc_hip.h"
#include "mode_tables.h"

namespace tamcmc {

// ---------- a-coefficient polynomials: acoefs.cpp:19-110 ----------
static long double ritzwoller_H(int s, int l, int m) {
    const int L = l * (l + 1);
    const double dm = m;
    long double H = 0;
    if (s == 5) H = 252 * std::pow(dm, 5) - 140 * (2 * L - 3) * std::pow(dm, 3) + (20 * L * (3 * L - 10) + 48) * m;
    if (s == 6)
        H = 924 * std::pow(dm, 6) - 420 * std::pow(dm, 4) * (3 * L - 7) +
            84 * std::pow(dm, 2) * (5 * std::pow((double)L, 2) - 25 * L + 14) -
            20 * L * (std::pow((double)L, 2) - 8 * L + 12);
    return H;
}

static long double Pslm_compute(int s, int l, int m) {
    const double dm = m, dl = l;
    long double Ps = 0;
    if (s == 1) Ps = m;
    if (s == 2 && l > 0) {
        const double v = (3 * std::pow(dm, 2) - l * (l + 1)) / (2 * l - 1);
        Ps = v;
    }
    if (s == 3 && l > 1) {
        const double v = (5 * std::pow(dm, 3) - (3 * l * (l + 1) - 1) * m) / ((l - 1) * (2 * l - 1));
        Ps = v;
    }
    if (s == 4) {
        const double h = (35 * std::pow(dm, 4) - 5 * (6 * l * (l + 1) - 5) * std::pow(dm, 2)) +
                         3 * l * (l + 1) * (l * (l + 1) - 2);
        const long double H = h, c = 2 * (l - 1) * (2 * l - 1) * (2 * l - 3);
        if (c != 0) Ps = H / c;
    }
    if (s == 5) {
        const double cd = 8 * (4 * std::pow(dl, 4) - 20 * std::pow(dl, 3) + 35 * std::pow(dl, 2) - 25 * l + 6);
        const long double H = ritzwoller_H(5, l, m), c = cd;
        if (c != 0) Ps = H / c;
    }
    if (s == 6) {
        const double cd = 64 * std::pow(dl, 5) - 480 * std::pow(dl, 4) + 1360 * std::pow(dl, 3) -
                          1800 * std::pow(dl, 2) + 1096 * l - 240;
        const long double H = ritzwoller_H(6, l, m), c = cd;
        if (c != 0) Ps = H / c;
    }
    return Ps;
}

// build_lorentzian.cpp:583-592
static double Qlm_compute(int l, int m) {
    const long double Dnl = 2. / 3;
    double q = (l * (l + 1) - 3 * std::pow((double)m, 2)) / ((2 * l - 1) * (2 * l + 3));
    q = q * Dnl;
    return q;
}

// The polynomials depend on (s,l,m) only: evaluate them once (s<=6, l<=3, |m|<=3) and serve every later call from
// the table -- same values, bit for bit, as recomputing them per multiplet like the reference does.
namespace {
struct PolyTables {
    long double P[7][4][7];
    double Q[4][7];
    PolyTables() {
        for (int s = 0; s <= 6; s++)
            for (int l = 0; l <= 3; l++)
                for (int m = -3; m <= 3; m++) P[s][l][m + 3] = Pslm_compute(s, l, m);
        for (int l = 0; l <= 3; l++)
            for (int m = -3; m <= 3; m++) Q[l][m + 3] = Qlm_compute(l, m);
    }
};
const PolyTables &poly() {
    static const PolyTables t;
    return t;
}
}  // namespace

long double Pslm(int s, int l, int m) {
    if (s >= 0 && s <= 6 && l >= 0 && l <= 3 && m >= -3 && m <= 3) return poly().P[s][l][m + 3];
    return Pslm_compute(s, l, m);
}
double Qlm(int l, int m) {
    if (l >= 0 && l <= 3 && m >= -3 && m <= 3) return poly().Q[l][m + 3];
    return Qlm_compute(l, m);
}

// ---------- m-visibilities: function_rot.cpp:15-101 ----------
static int ifact(int n) {
    long f = 1;
    for (long i = 1; i <= n; i++) f *= i;
    return (int)f;
}
static double icombi(int n, int r) { return ifact(n) / ifact(n - r) / ifact(r); }
static double wigner_d(int l, int m1, int m2, double beta) {
    double sum = 0;
    for (long s = 0; s <= l - m1; s++) {
        double v = icombi(l + m2, (int)(l - m1 - s)) * icombi(l - m2, (int)s) * std::pow(-1, (double)(l - m1 - s));
        v = v * std::pow(std::cos(beta / 2.), (double)(2 * s + m1 + m2)) *
            std::pow(std::sin(beta / 2.), (double)(2 * l - 2 * s - m1 - m2));
        sum = sum + v;
    }
    sum = sum * std::sqrt((double)(ifact(l + m1) * ifact(l - m1)));
    sum = sum / std::sqrt((double)(ifact(l + m2) * ifact(l - m2)));
    return sum;
}
void amplitude_ratio(int l, double beta_deg, double *V) {
    const double PI = 3.141592653589793238462643;
    const double ang = PI * beta_deg / 180.;
    // centre column (m'=0) of the rotation matrix as the four fill passes of function_rot() leave it
    for (int i = 0; i <= l; i++) V[l + i] = wigner_d(l, i, 0, ang);
    for (int i = -l; i <= 0; i++) V[l + i] = V[l - i] * std::pow(-1, (double)i);
    V[l] = wigner_d(l, 0, 0, -ang);
    V[l] = V[l] * std::pow(-1, 0.);
    for (int i = 0; i <= 2 * l; i++) V[i] = V[i] * V[i];
}

// ---------- interpol.cpp:13-43, linfit.cpp:17-35, models.cpp:6065-6084 ----------
double lin_interpol(const double *x, const double *y, long n, double xi) {
    double a = 0, b = 0;
    if (xi >= x[0] && xi <= x[n - 1]) {
        long i = 0;
        while (i < n - 2 && (xi < x[i] || xi > x[i + 1])) ++i;
        a = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
        b = y[i] - a * x[i];
    }
    if (xi < x[0]) {
        a = (y[1] - y[0]) / (x[1] - x[0]);
        b = y[0] - a * x[0];
    }
    if (xi > x[n - 1]) {
        a = (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]);
        b = y[n - 2] - a * x[n - 2];
    }
    return a * xi + b;
}

void linfit(const double *x, const double *y, long n, double out[2]) {
    double sx = 0, sy = 0, sty = 0, stt = 0;
    for (long i = 0; i < n; i++) sx += x[i];
    for (long i = 0; i < n; i++) sy += y[i];
    const double dn = (double)n, mx = sx / dn;
    for (long i = 0; i < n; i++) sty += (x[i] - mx) * y[i];
    for (long i = 0; i < n; i++) stt += (x[i] - mx) * (x[i] - mx);
    out[0] = sty / stt;
    out[1] = (sy - sx * out[0]) / dn;
}

double eta0_from_dnu(double dnu) {
    const double G = 6.667e-8, Dnu_sun = 135.1, R_sun = 6.96342e5, M_sun = 1.98855e30;
    const double rho_sun = M_sun * 1e3 / (4 * M_PI * std::pow(R_sun * 1e5, 3) / 3);
    const double rho = std::pow(dnu / Dnu_sun, 2.) * rho_sun;
    return 3. * M_PI / (rho * G);
}

double eta0_fct(const double *fl0, long n) {
    std::vector<double> idx((size_t)(n > 0 ? n : 1));
    for (long i = 0; i < n; i++) idx[(size_t)i] = (double)i;
    double r[2];
    linfit(idx.data(), fl0, n, r);
    return eta0_from_dnu(r[0]);
}

// ---------- truncation window: build_lorentzian.cpp:595-676 ----------
static int to_int_sat(double v) {
    if (v >= (double)INT_MAX) return INT_MAX;
    if (v <= (double)INT_MIN) return INT_MIN;
    return (int)v;
}

int set_imin_imax(double x_first, double x_last, int64_t Nx, int l, double fc, double gamma, double f_s, double c,
                  double step, int *i0, int *i1) {
    double lo = 0, hi = 0;
    bool have = false;
    auto span = [&](double half_l, double half_0) {
        const double h = (l != 0) ? half_l : half_0;
        lo = fc - h;
        hi = fc + h;
        have = true;
    };
    // the four overlapping regimes, later ones overriding earlier ones as in the reference
    if (gamma >= 1 && f_s >= 1) span(c * (l * f_s + gamma), c * gamma * 2.2);
    if (gamma <= 1 && f_s >= 1) span(c * (l * f_s + 1), c * 2.2);
    if (gamma >= 1 && f_s <= 1) span(c * (l + gamma), c * 2.2 * gamma);
    if (gamma <= 1 && f_s <= 1) span(c * (l + 1), c * 2.2);
    if (!have) return TAMCMC_ERR_NAN_WINDOW;
    if ((hi - step) < x_first) hi = x_first + c;
    if ((lo + step) >= x_last) lo = x_last - c;
    int a = to_int_sat(std::floor((lo - x_first) / step));
    int b = to_int_sat(std::ceil((hi - x_first) / step));
    if (a < 0) a = 0;
    if (b > Nx) b = (int)Nx;
    if (b - a <= 0) return TAMCMC_ERR_EMPTY_WINDOW;
    *i0 = a;
    *i1 = b;
    return TAMCMC_OK;
}

// ---------- split frequencies ----------
// build_lorentzian.cpp:226-229 (sum evaluated in long double because Pslm is long double)
static double nu_nlm_aj(double fc, const double a[7], double eta0, int l, int m) {
    long double acc = fc + a[1] * Pslm(1, l, m) + a[2] * Pslm(2, l, m) + a[3] * Pslm(3, l, m) +
                      a[4] * Pslm(4, l, m) + a[5] * Pslm(5, l, m) + a[6] * Pslm(6, l, m);
    double nu = (double)acc;
    if (eta0 > 0) nu = nu + fc * eta0 * Qlm(l, m) * std::pow(a[1] * 1e-6, 2);
    return nu;
}
// build_lorentzian.cpp:145
static double nu_nlm_a1etaa3(double fc, double f_s, double eta0, double a3, int l, int m) {
    const double t = fc * (1. + eta0 * std::pow(f_s * 1e-6, 2) * Qlm(l, m)) + m * f_s;
    const long double acc = t + Pslm(3, l, m) * a3;
    return (double)acc;
}

namespace {

struct Builder {
    const double *x;
    int64_t Nx;
    double step;
    tamcmc_multiplet *out;
    int cap;
    int n = 0;
    tamcmc_multiplet scratch;  // rows beyond `cap` land here: the caller only learns the needed count

    tamcmc_multiplet *next() {
        tamcmc_multiplet *r = (n < cap) ? &out[n] : &scratch;
        ++n;
        std::memset(r, 0, sizeof(*r));
        return r;
    }
    // optimum_lorentzian_calc_aj (build_lorentzian.cpp:502-522)
    int add_aj(double H, double fc, const double a[7], double eta0, double asym, double gamma, int l, const double *V,
               double c) {
        int i0, i1;
        const int st = set_imin_imax(x[0], x[Nx - 1], Nx, l, fc, gamma, a[1], c, step, &i0, &i1);
        if (st) return st;
        tamcmc_multiplet *r = next();
        r->l = l; r->i0 = i0; r->i1 = i1; r->fc = fc; r->gamma = gamma; r->asym = asym;
        for (int m = -l; m <= l; m++) {
            r->nu[m + l] = (l != 0) ? nu_nlm_aj(fc, a, eta0, l, m) : fc;
            r->hv[m + l] = H * V[m + l];
        }
        return TAMCMC_OK;
    }
    // optimum_lorentzian_calc_a1etaa3 (build_lorentzian.cpp:441-458)
    int add_a1etaa3(double H, double fc, double f_s, double eta0, double a3, double asym, double gamma, int l,
                    const double *V, double c) {
        int i0, i1;
        const int st = set_imin_imax(x[0], x[Nx - 1], Nx, l, fc, gamma, f_s, c, step, &i0, &i1);
        if (st) return st;
        tamcmc_multiplet *r = next();
        r->l = l; r->i0 = i0; r->i1 = i1; r->fc = fc; r->gamma = gamma; r->asym = asym;
        for (int m = -l; m <= l; m++) {
            r->nu[m + l] = (l != 0) ? nu_nlm_a1etaa3(fc, f_s, eta0, a3, l, m) : fc;
            r->hv[m + l] = H * V[m + l];
        }
        return TAMCMC_OK;
    }
};

struct Layout {  // params_length decoding shared by the three models (models.cpp:1207-1219)
    int Nmax, lmax, Nfl[4], Nsplit, Nwidth, Nnoise, Ninc, Nf;
    int o_vis, o_f[4], o_split, o_width, o_noise, o_inc, o_cfg;
    explicit Layout(const int32_t *pl) {
        Nmax = pl[0]; lmax = pl[1];
        for (int i = 0; i < 4; i++) Nfl[i] = pl[2 + i];
        Nsplit = pl[6]; Nwidth = pl[7]; Nnoise = pl[8]; Ninc = pl[9];
        Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
        o_vis = Nmax;
        o_f[0] = Nmax + lmax;
        for (int i = 1; i < 4; i++) o_f[i] = o_f[i - 1] + Nfl[i - 1];
        o_split = Nmax + lmax + Nf;
        o_width = o_split + Nsplit;
        o_noise = o_width + Nwidth;
        o_inc = o_noise + Nnoise;
        o_cfg = o_inc + Ninc;
    }
};

void emit_noise(const double *p, const Layout &L, int nharvey_used, double *noise_abs, int *nharvey, int *nnoise) {
    for (int i = 0; i < L.Nnoise; i++) noise_abs[i] = std::abs(p[L.o_noise + i]);
    *nharvey = nharvey_used;
    *nnoise = L.Nnoise;
}

// models.cpp:1195-1408
int table_aj(const double *p, const int32_t *pl, Builder &B, double *noise_abs, int *nharvey, int *nnoise) {
    const Layout L(pl);
    const long double pi = M_PI;
    const double trunc_c = p[L.o_cfg];
    const bool do_amp = p[L.o_cfg + 1];
    const double inc = p[L.o_inc];
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double *ratios[4] = {r0, r1, r2, r3};
    double Vl[4] = {1, 0, 0, 0};
    for (int l = 1; l <= 3; l++)
        if (L.lmax >= l) {
            Vl[l] = std::abs(p[L.o_vis + l - 1]);
            amplitude_ratio(l, inc, ratios[l]);
        }
    const double *fl0 = p + L.o_f[0];
    const double *Wl0 = p + L.o_width;
    const double *Hl0 = p;
    const double *sp = p + L.o_split;  // a1_0,a1_1,..,a6_0,a6_1, eta switch (@12), asym (@13)
    const double asym = sp[13];
    const double eta0 = (sp[12] == 1) ? eta0_fct(fl0, L.Nfl[0]) : 0.0;
    const double zero[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int n = 0; n < L.Nfl[0]; n++) {
        const double W = std::abs(Wl0[n]);
        double H;
        if (do_amp) H = std::abs(p[n] / (pi * W));
        else H = std::abs(p[n]);
        if (int st = B.add_aj(H, fl0[n], zero, 0.0, asym, W, 0, r0, trunc_c)) return st;
    }
    for (int l = 1; l <= 3; l++) {
        for (int n = 0; n < L.Nfl[l]; n++) {
            const double f = p[L.o_f[l] + n];
            const double W = std::abs(lin_interpol(fl0, Wl0, L.Nfl[0], f));
            double H;
            if (do_amp) H = std::abs(lin_interpol(fl0, Hl0, L.Nfl[0], f) / (pi * W) * Vl[l]);
            else H = std::abs(lin_interpol(fl0, Hl0, L.Nfl[0], f) * Vl[l]);
            double a[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int j = 1; j <= 2 * l; j++) a[j] = sp[2 * (j - 1)] + sp[2 * (j - 1) + 1] * (f * 1e-3);
            if (int st = B.add_aj(H, f, a, eta0, asym, W, l, ratios[l], trunc_c)) return st;
        }
    }
    emit_noise(p, L, (L.Nnoise - 1) / 3, noise_abs, nharvey, nnoise);
    return TAMCMC_OK;
}

// models.cpp:1943-2121
int table_classic(const double *p, const int32_t *pl, Builder &B, double *noise_abs, int *nharvey, int *nnoise) {
    const Layout L(pl);
    const long double pi = 3.141592653589793238462643383279502884L;
    const double trunc_c = p[L.o_cfg];
    const bool do_amp = p[L.o_cfg + 1];
    const double inc = p[L.o_inc];
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double *ratios[4] = {r0, r1, r2, r3};
    double Vl[4] = {1, 0, 0, 0};
    for (int l = 1; l <= 3; l++)
        if (L.lmax >= l) {
            Vl[l] = std::abs(p[L.o_vis + l - 1]);
            amplitude_ratio(l, inc, ratios[l]);
        }
    const double *fl0 = p + L.o_f[0];
    const double *Wl0 = p + L.o_width;
    const double a1 = std::abs(p[L.o_split]);
    const double eta0 = eta0_fct(fl0, L.Nfl[0]);
    const double a3 = p[L.o_split + 2];
    const double asym = p[L.o_split + 5];
    for (long n = 0; n < L.Nmax; n++) {
        const double W0 = std::abs(Wl0[n]);
        double H0;
        if (do_amp) H0 = std::abs(p[n] / (pi * W0));
        else H0 = std::abs(p[n]);
        if (int st = B.add_a1etaa3(H0, fl0[n], a1, eta0, a3, asym, W0, 0, r0, trunc_c)) return st;
        for (int l = 1; l <= 3; l++) {
            if (L.lmax < l) continue;
            const double f = p[L.o_f[l] + n];
            const double W = std::abs(lin_interpol(fl0, Wl0, L.Nfl[0], f));
            double H;
            if (do_amp) H = std::abs(p[n] / (pi * W)) * Vl[l];
            else H = std::abs(p[n] * Vl[l]);
            if (int st = B.add_a1etaa3(H, f, a1, eta0, a3, asym, W, l, ratios[l], trunc_c)) return st;
        }
    }
    emit_noise(p, L, (L.Nnoise - 1) / 3, noise_abs, nharvey, nnoise);
    return TAMCMC_OK;
}

// models.cpp:3012-3195
int table_local_basic(const double *p, const int32_t *pl, Builder &B, double *noise_abs, int *nharvey, int *nnoise) {
    const Layout L(pl);  // here plength[1] = Nvis
    const long double pi = 3.141592653589793238462643383279502884L;
    const double trunc_c = p[L.o_cfg];
    const bool do_amp = p[L.o_cfg + 1];
    double inc = std::atan(p[L.o_split + 4] / p[L.o_split + 3]);
    inc = inc * 180. / pi;
    const double a1 = std::pow(p[L.o_split + 3], 2) + std::pow(p[L.o_split + 4], 2);
    double r0[1] = {1.0}, r1[3], r2[5], r3[7];
    double *ratios[4] = {r0, r1, r2, r3};
    for (int l = 1; l <= 3; l++)
        if (L.Nfl[l] >= 1) amplitude_ratio(l, inc, ratios[l]);
    const double eta0 = p[L.o_split + 1];
    const double a3 = p[L.o_split + 2];
    const double asym = p[L.o_split + 5];
    int off = 0;
    for (int l = 0; l <= 3; l++) {
        for (long n = 0; n < L.Nfl[l]; n++) {
            const double f = p[L.o_f[0] + off + n];
            const double W = std::abs(p[L.o_width + off + n]);
            double H;
            if (do_amp) H = std::abs(p[off + n] / (pi * W));
            else H = std::abs(p[off + n]);
            if (int st = B.add_a1etaa3(H, f, a1, eta0, a3, asym, W, l, ratios[l], trunc_c)) return st;
        }
        off += L.Nfl[l];
    }
    emit_noise(p, L, 0, noise_abs, nharvey, nnoise);  // Nharvey forced to 0 (models.cpp:3167)
    return TAMCMC_OK;
}

}  // namespace

int build_mode_table(int model_id, const double *params, const int32_t *plength, const double *x, int64_t Nx,
                     tamcmc_multiplet *mults, int max_mults, int *n_mults, double *noise_abs, int *nharvey,
                     int *nnoise) {
    if (!params || !plength || !x || Nx < 2 || !n_mults || !noise_abs || !nharvey || !nnoise) return TAMCMC_ERR_BAD_ARG;
    Builder B{x, Nx, x[1] - x[0], mults, mults ? max_mults : 0, 0, {}};
    int st;
    switch (model_id) {
    case TAMCMC_MODEL_MS_GLOBAL_AJ: st = table_aj(params, plength, B, noise_abs, nharvey, nnoise); break;
    case TAMCMC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC: st = table_classic(params, plength, B, noise_abs, nharvey, nnoise); break;
    case TAMCMC_MODEL_MS_LOCAL_BASIC: st = table_local_basic(params, plength, B, noise_abs, nharvey, nnoise); break;
    default: return TAMCMC_ERR_BAD_MODEL;
    }
    *n_mults = B.n;
    return st;
}

int count_multiplets(int model_id, const int32_t *pl) {
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

}  // namespace tamcmc

extern "C" int tamcmc_build_mode_table(int model_id, const double *params, const int32_t *plength, const double *x,
                                       int64_t Nx, tamcmc_multiplet *mults, int max_mults, int *n_mults,
                                       double *noise_abs, int *nharvey, int *nnoise) {
    return tamcmc::build_mode_table(model_id, params, plength, x, Nx, mults, max_mults, n_mults, noise_abs, nharvey,
                                    nnoise);
}
