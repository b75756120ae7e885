// mode_tables.cpp -- host-side scalar half of the model functions (product code, C++).
//
// Every Lorentzian model of the reference's dispatch table (model_def.cpp:220-388) unpacks `params`
// by `params_length` into per-mode scalars and then calls optimum_lorentzian_calc_* per (n,l) multiplet
// (SURVEY App. D).  This file performs exactly that scalar part -- parameter unpack, m-visibilities,
// width/height interpolation, a-coefficients -> nu_nlm, eta0, truncation window -- and emits the flat
// multiplet table the device kernel consumes.  The per-bin work is NOT done here.
// The arithmetic lives in mode_tables_impl.h (shared with the device-resident sampler); on the host the
// extended type is long double, so the table is bit-identical to what the reference would feed its own
// per-bin loops.  Must be compiled without FMA contraction (-ffp-contract=off).
#include <cmath>
#include <cstring>

#include "mode_tables.h"
#include "mode_tables_impl.h"

namespace tamcmc {

// The polynomials depend on (s,l,m) only: evaluate them once (s<=6, l<=3, |m|<=3) and serve every later call from
// the table -- same values, bit for bit, as recomputing them per multiplet like the reference does.
const mt::PolyTab &poly_table() {
    static const mt::PolyTab t = [] {
        mt::PolyTab p;
        mt::fill_poly(p);
        return p;
    }();
    return t;
}

long double Pslm(int s, int l, int m) {
    if (s >= 0 && s <= 6 && l >= 0 && l <= 3 && m >= -3 && m <= 3) return poly_table().P[s][l][m + 3];
    return mt::Pslm_compute(s, l, m);
}
double Qlm(int l, int m) {
    if (l >= 0 && l <= 3 && m >= -3 && m <= 3) return poly_table().Q[l][m + 3];
    return mt::Qlm_compute(l, m);
}
void amplitude_ratio(int l, double beta_deg, double *V) { mt::amplitude_ratio(l, beta_deg, V); }
double lin_interpol(const double *x, const double *y, long n, double xi) { return mt::lin_interpol(x, y, n, xi); }
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
double eta0_from_dnu(double dnu) { return mt::eta0_from_dnu(dnu); }
double eta0_fct(const double *fl0, long n) { return mt::eta0_fct(fl0, n); }
int set_imin_imax(double x_first, double x_last, int64_t Nx, int l, double fc, double gamma, double f_s, double c,
                  double step, int *i0, int *i1) {
    return mt::set_imin_imax(x_first, x_last, Nx, l, fc, gamma, f_s, c, step, i0, i1);
}
int count_multiplets(int model_id, const int32_t *pl) { return mt::count_multiplets(model_id, pl); }

int build_mode_table(int model_id, const double *params, const int32_t *plength, const double *x, int64_t Nx,
                     tamcmc_multiplet *mults, int max_mults, int *n_mults, double *noise_abs, int *nharvey,
                     int *nnoise) {
    if (!params || !plength || !x || Nx < 2 || !n_mults || !noise_abs || !nharvey || !nnoise) return TAMCMC_ERR_BAD_ARG;
    const int total = mt::count_multiplets(model_id, plength);
    if (total < 0) return TAMCMC_ERR_BAD_MODEL;
    mt::Shared S;
    mt::shared_scalars(model_id, params, plength, S);
    const mt::PolyTab &T = poly_table();
    const double step = x[1] - x[0];
    tamcmc_multiplet scratch;  // rows beyond max_mults land here: the caller only learns the needed count
    int n = 0;
    for (int i = 0; i < total; i++) {
        tamcmc_multiplet *r = (mults && n < max_mults) ? &mults[n] : &scratch;
        const int st = mt::build_multiplet(model_id, T, params, S, i, x[0], x[Nx - 1], Nx, step, r);
        if (st) { *n_mults = n; return st; }
        n++;
    }
    *n_mults = n;
    for (int i = 0; i < S.L.Nnoise; i++) noise_abs[i] = std::fabs(params[S.L.o_noise + i]);
    *nharvey = S.nharvey;
    *nnoise = S.L.Nnoise;
    return TAMCMC_OK;
}

}  // namespace tamcmc

extern "C" int tamcmc_build_mode_table(int model_id, const double *params, const int32_t *plength, const double *x,
                                       int64_t Nx, tamcmc_multiplet *mults, int max_mults, int *n_mults,
                                       double *noise_abs, int *nharvey, int *nnoise) {
    return tamcmc::build_mode_table(model_id, params, plength, x, Nx, mults, max_mults, n_mults, noise_abs, nharvey,
                                    nnoise);
}
