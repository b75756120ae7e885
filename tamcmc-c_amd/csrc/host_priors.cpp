// host_priors.cpp -- log-priors (host, O(Nparams) per chain per step; never on the device in the host-driven sampler).
//   primitives           tamcmc/sources/stats_dictionary.cpp:38-250
//   apply_generic_priors tamcmc/sources/priors_calc.cpp:725-870
//   priors_MS_Global     tamcmc/sources/priors_calc.cpp:27-317   (model_index 9 = aj family, default = Classic)
//   priors_local         tamcmc/sources/priors_calc.cpp:514-629
// Arithmetic in long double like the reference.  Where the reference exits (unsupported prior ids, model classes
// flagged "needs checks") *status is set to TAMCMC_ERR_BAD_MODEL and -inf is returned.
#include <cmath>
#include <limits>

#include "host_sampler.h"
#include "mode_tables.h"

namespace tamcmc {

static const long double PIl = 3.141592653589793238462643383279502884L;
static const long double NEG_INF = -std::numeric_limits<long double>::infinity();

long double logP_uniform(long double b_min, long double b_max, long double x) {
    if ((x <= b_max) && (x >= b_min)) return -std::log(std::abs(b_max - b_min));
    return NEG_INF;
}
long double logP_uniform_abs(long double b_min, long double b_max, long double x) {
    if ((std::abs(x) <= b_max) && (std::abs(x) >= b_min)) return -std::log(std::abs(b_max - b_min));
    return NEG_INF;
}
long double logP_gaussian(long double mean, long double sigma, long double x) {
    return -std::log(std::sqrt(2 * PIl) * sigma) - 0.5 * std::pow((x - mean) / sigma, 2.);
}
long double logP_jeffrey(long double hmin, long double hmax, long double h) {
    if (h < hmax && h > 0) {
        const long double prior = 1. / (h + hmin), norm = std::log((hmax + hmin) / hmin);
        return std::log(prior / norm);
    }
    return NEG_INF;
}
long double logP_jeffrey_abs(long double hmin, long double hmax, long double h) {
    if (std::abs(h) < hmax) {
        const long double prior = 1. / (std::abs(h) + hmin), norm = std::log((hmax + hmin) / hmin);
        return std::log(prior / norm);
    }
    return NEG_INF;
}
long double logP_uniform_gaussian(long double b_min, long double b_max, long double sigma, long double x) {
    long double logP = 0;
    if (x < b_min) logP = NEG_INF;
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * std::pow((x - b_max) / sigma, 2.);
    return logP - std::log(std::abs(b_max - b_min) + 0.5 * std::sqrt(2 * PIl) * sigma);
}
long double logP_gaussian_uniform(long double b_min, long double b_max, long double sigma, long double x) {
    long double logP = 0;
    if (x > b_max) logP = NEG_INF;
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x < b_min) logP = -0.5 * std::pow((x - b_min) / sigma, 2.);
    return logP - std::log(std::abs(b_max - b_min) + 0.5 * std::sqrt(2 * PIl) * sigma);
}
long double logP_gaussian_uniform_gaussian(long double b_min, long double b_max, long double s1, long double s2,
                                           long double x) {
    long double logP = 0;
    if (x < b_min) logP = -0.5 * std::pow((x - b_min) / s1, 2.);
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * std::pow((x - b_max) / s2, 2.);
    return logP - std::log(std::abs(b_max - b_min) + 0.5 * std::sqrt(2 * PIl) * (s1 + s2));
}

long double apply_generic_priors(const double *params, long Nparams, const Matrix &pp, const std::vector<int> &sw,
                                 int *status) {
    long double pena = 0;
    for (long i = 0; i < Nparams; i++) {
        switch (sw[(size_t)i]) {
        case 0: case 13: break;
        case 1: pena = pena + logP_uniform(pp(0, i), pp(1, i), params[i]); break;
        case 2: pena = pena + logP_gaussian(pp(0, i), pp(1, i), params[i]); break;
        case 4: pena = pena + logP_jeffrey(pp(0, i), pp(1, i), params[i]); break;
        case 5: pena = pena + logP_uniform_gaussian(pp(0, i), pp(1, i), pp(2, i), params[i]); break;
        case 6: pena = pena + logP_gaussian_uniform(pp(0, i), pp(1, i), pp(2, i), params[i]); break;
        case 7: pena = pena + logP_gaussian_uniform_gaussian(pp(0, i), pp(1, i), pp(2, i), pp(3, i), params[i]); break;
        case 8: pena = pena + logP_uniform_abs(pp(0, i), pp(1, i), params[i]); break;
        case 10: pena = pena + logP_jeffrey_abs(pp(0, i), pp(1, i), params[i]); break;
        default:  // 3 multivariate (fatal in the reference), 9 flagged buggy, 11 unusable, 12 needs GSL tables
            if (status) *status = TAMCMC_ERR_BAD_MODEL;
            return NEG_INF;
        }
    }
    return pena;
}

// second differences with replicated edges: Scndder_adaptive_reggrid(y) (derivatives_handler.cpp:400-426)
static void second_differences(const double *y, long n, std::vector<double> &d) {
    d.assign((size_t)n, 0.0);
    if (n < 3) return;
    d[0] = y[2] - 2. * y[1] + y[0];
    d[(size_t)n - 1] = y[n - 1] - 2. * y[n - 2] + y[n - 3];
    for (long i = 0; i < n - 2; i++) d[(size_t)i + 1] = y[i + 2] - 2. * y[i + 1] + y[i];
}

long double priors_MS_Global(const double *params, const std::vector<int> &pl, const Matrix &pp,
                             const std::vector<int> &sw, const std::vector<double> &extra, int *status) {
    long double f = 0;
    const int smooth_switch = (int)extra[0];
    const double scoef = extra[1];
    const double *ajova1_limit = &extra[2];
    const int impose_normHnlm = (int)extra[8];
    const int model_index = (int)extra[9];
    const int Nmax = pl[0], lmax = pl[1];
    const int Nfl[4] = {pl[2], pl[3], pl[4], pl[5]};
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8];
    const int Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
    long Nparams = 0;
    for (int v : pl) Nparams += v;
    (void)Nnoise;

    for (int i = Nmax; i <= Nmax + lmax; i++)  // positivity of the visibilities (priors_calc.cpp:63-68)
        if (params[i] < 0) return NEG_INF;

    switch (model_index) {
    case 9: {  // aj family: |aj/a1| limits at every l>0 frequency, a1 >= 0 (priors_calc.cpp:206-228)
        int i0 = Nfl[0];
        for (int el = 1; el < lmax + 1; el++) {
            for (int j = 1; j < 6; j++) {
                for (int n = 0; n < Nfl[el]; n++) {
                    const double fl = params[Nmax + lmax + i0 + n];
                    const double a1 = params[Nmax + lmax + Nf] + params[Nmax + lmax + Nf + 1] * (fl * 1e-3);
                    const double aj = params[Nmax + lmax + Nf + 2 * j] + params[Nmax + lmax + Nf + 2 * j + 1] * (fl * 1e-3);
                    if (std::abs(aj / a1) >= ajova1_limit[j]) return NEG_INF;
                    if (a1 < 0) return NEG_INF;
                }
            }
            i0 = i0 + Nfl[el];
        }
        break;
    }
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8:
        // families this build does not ship a table builder for
        if (status) *status = TAMCMC_ERR_BAD_MODEL;
        return NEG_INF;
    default:  // Classic models (priors_calc.cpp:230-262)
        if (impose_normHnlm != 0) {
            if (status) *status = TAMCMC_ERR_BAD_MODEL;
            return NEG_INF;
        }
        break;
    }
    const int on = Nmax + lmax + Nf + Nsplit + Nwidth;  // noise block
    if (sw[(size_t)on + 3] != 0)
        if ((params[on + 3] < 0) || (params[on + 4] < 0) || (params[on + 5] < 0)) return NEG_INF;
    if (sw[(size_t)on + 6] != 0)
        if ((params[on + 6] < 0) || (params[on + 7] < 0) || (params[on + 8] < 0)) return NEG_INF;
    if ((sw[(size_t)(Nmax + lmax + Nf + 9)] != 0) && (params[on + 9] < 0)) return NEG_INF;  // index as in :272

    f = f + apply_generic_priors(params, Nparams, pp, sw, status);

    std::vector<double> idx((size_t)Nfl[0]);
    for (int i = 0; i < Nfl[0]; i++) idx[(size_t)i] = i;
    double fit[2];
    linfit(idx.data(), params + Nmax + lmax, Nfl[0], fit);
    const double Dnu = fit[0];
    if (Nfl[0] == Nfl[2]) {  // d02 ~ GU(0, Dnu/3, 0.015 Dnu) (priors_calc.cpp:289-294)
        for (int i = 0; i < Nfl[0]; i++) {
            const double d02 = params[Nmax + lmax + i] - params[Nmax + lmax + Nfl[0] + Nfl[1] + i];
            f = f + logP_gaussian_uniform(0, Dnu / 3., 0.015 * Dnu, d02);
        }
    }
    if (smooth_switch == 1) {  // smoothness of each l's frequency list (priors_calc.cpp:299-313)
        int i0 = 0;
        std::vector<double> d2;
        for (int el = 0; el < lmax + 1; el++) {
            if (Nfl[el] != 0) {
                second_differences(params + Nmax + lmax + i0, Nfl[el], d2);
                for (int i = 0; i < Nfl[el]; i++) f = f + logP_gaussian(0, scoef, d2[(size_t)i]);
            }
            i0 = i0 + Nfl[el];
        }
    }
    return f;
}

long double priors_local(const double *params, const std::vector<int> &pl, const Matrix &pp, const std::vector<int> &sw,
                         const std::vector<double> &extra, int *status) {
    long double f = 0;
    const double a3ova1_limit = extra[2];
    const int Nmax = pl[0], Nvis = pl[1];
    const int Nf = pl[2] + pl[3] + pl[4] + pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8];
    long Nparams = 0;
    for (int v : pl) Nparams += v;
    const int o = Nmax + Nvis + Nf;
    if (params[o] != 0) {  // an a1 is fitted directly
        if (std::abs(params[o + 2] / params[o]) >= a3ova1_limit) return NEG_INF;
    } else if ((params[o + 3] != 0) && (params[o + 4] != 0)) {  // sqrt(a1) cos i, sqrt(a1) sin i
        if (std::abs(params[o + 2] / (std::pow(params[o + 3], 2) + std::pow(params[o + 4], 2))) >= a3ova1_limit)
            return NEG_INF;
    }
    const int oi = o + Nsplit + Nwidth + Nnoise;  // inclination slot
    if ((sw[(size_t)oi] != 0) && (params[oi] < 0)) return NEG_INF;
    f = f + apply_generic_priors(params, Nparams, pp, sw, status);
    return f;
}

}  // namespace tamcmc
