// host_priors.cpp -- log-priors on the host (O(Nparams) per chain per step).  The arithmetic is in priors_impl.h
// (shared with the device-resident sampler); here it runs in long double, in the reference's term order.
#include "host_sampler.h"
#include "priors_impl.h"

namespace tamcmc {

long double logP_uniform(long double a, long double b, long double x) { return pr::logP_uniform(a, b, x); }
long double logP_uniform_abs(long double a, long double b, long double x) { return pr::logP_uniform_abs(a, b, x); }
long double logP_gaussian(long double m, long double s, long double x) { return pr::logP_gaussian(m, s, x); }
long double logP_jeffrey(long double a, long double b, long double h) { return pr::logP_jeffrey(a, b, h); }
long double logP_jeffrey_abs(long double a, long double b, long double h) { return pr::logP_jeffrey_abs(a, b, h); }
long double logP_uniform_gaussian(long double a, long double b, long double s, long double x) { return pr::logP_uniform_gaussian(a, b, s, x); }
long double logP_gaussian_uniform(long double a, long double b, long double s, long double x) { return pr::logP_gaussian_uniform(a, b, s, x); }
long double logP_gaussian_uniform_gaussian(long double a, long double b, long double s1, long double s2, long double x) {
    return pr::logP_gug(a, b, s1, s2, x);
}

long double apply_generic_priors(const double *params, long Nparams, const Matrix &pp, const std::vector<int> &sw, int *status) {
    long double pena = 0;
    for (long i = 0; i < Nparams; i++) pena = pena + pr::generic_prior_term(params, Nparams, pp.a.data(), sw.data(), i, status);
    return pena;
}

static long nparams_of(const std::vector<int> &pl) {
    long n = 0;
    for (int v : pl) n += v;
    return n;
}

long double priors_MS_Global(const double *params, const std::vector<int> &pl, const Matrix &pp, const std::vector<int> &sw,
                             const std::vector<double> &extra, int *status) {
    return pr::prior_serial(2, params, pl.data(), nparams_of(pl), pp.a.data(), sw.data(), extra.data(), status);
}

long double priors_local(const double *params, const std::vector<int> &pl, const Matrix &pp, const std::vector<int> &sw,
                         const std::vector<double> &extra, int *status) {
    return pr::prior_serial(3, params, pl.data(), nparams_of(pl), pp.a.data(), sw.data(), extra.data(), status);
}

long double priors_asymptotic(const double *params, const std::vector<int> &pl, const Matrix &pp, const std::vector<int> &sw,
                              const std::vector<double> &extra, int *status) {
    return pr::prior_serial(4, params, pl.data(), nparams_of(pl), pp.a.data(), sw.data(), extra.data(), status);
}

}  // namespace tamcmc
