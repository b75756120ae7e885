// bg_series.h -- Taylor series of the background (Harvey terms + white noise) on one tile of the likelihood kernel.
// H/(1+(a x)^p) with a = 1e-3 tau is analytic on a tile whose centre x_c is far from 0 (h/x_c <= EPS_MAX): with
// s = (x-x_c)/h,  u(s) = (a x_c)^p (1+eps s)^p  (binomial series, eps = h/x_c), then the reciprocal series of 1+u.
// The NH coefficients join the tile polynomial of the FAST far field (kernels.hip).  They depend on (evaluation, tile)
// only, so whoever builds an evaluation's table can build them once (dev_unpack.h, k_bg_poly) instead of every tile's
// workgroup; the arithmetic is this one function either way.   (noise_models.cpp:15-39 is the per-bin definition.)
#pragma once
#include <hip/hip_runtime.h>

namespace tamcmc {
namespace bg {

constexpr int NH = 8;             // Taylor coefficients of a Harvey term on a tile (x_c >> h)
constexpr double EPS_MAX = 0.02;  // ... used when h/x_c <= EPS_MAX: truncation ~ C(p,8) 0.02^8 = 2.6e-14

// v_rcp_f64 seed + two Newton-Raphson steps: ~1 ulp
__device__ __forceinline__ double rcp2(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

// centre and half-width of tile `tile` (tile_bins bins) on the regular grid x[i] = x0 + i*step
__device__ __forceinline__ void tile_geometry(int tile, int tile_bins, double x0, double step, double &xc, double &h) {
    const int t0 = tile * tile_bins;
    h = 0.5 * (double)tile_bins * step;
    xc = x0 + ((double)t0 + 0.5 * (double)tile_bins - 0.5) * step;
}
__device__ __forceinline__ bool series_valid(double xc, double h) { return (fabs(h) <= EPS_MAX * fabs(xc)) && (xc > 0.0); }

// series of ONE term Hh/(1+(1e-3 tau x)^pw) about x_c; f[] = 0 when tau == 0 (noise_models.cpp:29)
__device__ __forceinline__ void harvey_term_series(double Hh, double tau, double pw, double xc, double h, double (&f)[NH]) {
#pragma unroll
    for (int k = 0; k < NH; k++) f[k] = 0.0;
    if (tau != 0.0) {
        const double eps = h / xc;
        double u[NH];
        u[0] = exp(pw * log(1e-3 * tau * xc));
#pragma unroll
        for (int k = 0; k < NH - 1; k++) u[k + 1] = u[k] * eps * (pw - (double)k) * (1.0 / (double)(k + 1));
        const double iv0 = rcp2(1.0 + u[0]);
        f[0] = Hh * iv0;
#pragma unroll
        for (int k = 1; k < NH; k++) {
            double acc2 = 0.0;
#pragma unroll
            for (int jj = 1; jj <= k; jj++) acc2 = fma(u[jj], f[k - jj], acc2);
            f[k] = -acc2 * iv0;
        }
    }
}

// all terms of one evaluation, summed in term order, + white noise (the LAST of the nn noise values)
template <class NoiseAt>
__device__ __forceinline__ void tile_series(NoiseAt nz, int nh, int nn, double xc, double h, double (&out)[NH]) {
#pragma unroll
    for (int k = 0; k < NH; k++) out[k] = 0.0;
    for (int t = 0; t < nh; t++) {
        double f[NH];
        harvey_term_series(nz(3 * t), nz(3 * t + 1), nz(3 * t + 2), xc, h, f);
#pragma unroll
        for (int k = 0; k < NH; k++) out[k] = out[k] + f[k];
    }
    out[0] = out[0] + nz(nn - 1);
}

}  // namespace bg
}  // namespace tamcmc
