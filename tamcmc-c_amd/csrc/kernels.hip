// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the TAMCMC hot path.
//
// k_loglike : fused  model row (sum of truncated Lorentzian multiplets + Harvey background)
//             -> chi^2(2 d.o.f.) terms y/M + ln M -> per-workgroup partial sums
//             (replaces build_l_mode_* build_lorentzian.cpp:131-161,208-246, the windowed adds of
//              optimum_lorentzian_calc_* :441-458,502-522, harvey_like noise_models.cpp:15-39 and
//              likelihood_chi22p likelihoods.cpp:17-28 of the reference, for a whole batch of chains)
// k_finalize: deterministic second-level reduction of the partials, one workgroup per evaluation.
//
// Mapping: one workgroup = 256 threads = 4 wave64 = one tile of 256*K consecutive bins of ONE evaluation.
// x/y are read coalesced (lane i -> bin base+i); the multiplets whose window intersects the tile are
// compacted IN ORDER into LDS by wave 0 (ballot + popcount prefix), with the per-multiplet scalars hoisted
// (gamma^2 or 2/gamma, asymmetry constants, "covers the whole tile" / "product cannot overflow" flags), and are
// then broadcast-read by every lane as 16-byte {nu_m, H*V_m} pairs into registers; the per-degree bodies are
// fully unrolled (2l+1 = 1,3,5,7).  The log-likelihood terms are reduced with wave64 shuffles, then across the
// 4 waves through LDS.  No MFMA: the path is elementwise + reduction, bounded by fp64 VALU throughput.
// Block index -> (tile, evaluation) is XCD-aware: blocks b, b+8, b+16.. share an XCD (round-robin dispatch),
// so all evaluations of one tile are placed on the same XCD and re-read x/y from that XCD's L2.
//
// Two arithmetic modes (see include/tamcmc_hip.h): STRICT keeps the reference's per-bin operation order
// (this file is compiled with -ffp-contract=off; every fused multiply-add below is an explicit fma()).
// FAST measured costs on MI355X (tools/ubench.hip): fma/add/mul ~1 issue slot, v_rcp_f64 ~3.3 slots with
// 2^-24.4 relative accuracy, IEEE divide ~12.6, exp ~21, log ~72, pow ~150 slots.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "bg_series.h"
#include "loglike_tile.h"

namespace tamcmc {

namespace {

using namespace tile;

constexpr int WG = 256;  // k_finalize / k_bg_poly; k_loglike's workgroup size is a template parameter (256 = 4 waves, or 64 = one wave per tile)

template <int MODE, int WGS, int K, bool WRITE_MODEL, bool DELTA = false>
__global__ void __launch_bounds__(WGS) k_loglike(const LoglikeArgs a) {
    __shared__ TileLds<MODE, WGS> lds;
    loglike_tile<MODE, WGS, K, WRITE_MODEL, DELTA>(a, (int)blockIdx.x, lds, NoTail());
}

// Background series of every (evaluation, tile) of a launch: one thread each (bg_series.h).
__global__ void __launch_bounds__(WG) k_bg_poly(const double *noise, int noise_stride, const int32_t *nharvey, const int32_t *nnoise,
                                               int B, int ntiles, int tile_bins, double x0, double step, double *out) {
    const long id = (long)blockIdx.x * WG + threadIdx.x;
    if (id >= (long)B * ntiles) return;
    const int b = (int)(id / ntiles), tile = (int)(id - (long)b * ntiles);
    const int nn = nnoise[b];
    if (nn <= 0) return;
    double xc, h;
    bg::tile_geometry(tile, tile_bins, x0, step, xc, h);
    if (!bg::series_valid(xc, h)) return;
    const double *nz = noise + (size_t)b * noise_stride;
    double o[NH];
    bg::tile_series([nz](int i) { return nz[i]; }, nharvey[b], nn, xc, h, o);
#pragma unroll
    for (int k = 0; k < NH; k++) out[(size_t)id * NH + k] = o[k];
}

// One workgroup per evaluation: fixed-order sum of the per-tile partials -> S[b] = sum1 + sum2.
__global__ void __launch_bounds__(WG) k_finalize(const double *partials, int ntiles, double *S) {
    __shared__ double s_red[2 * (WG / 64)];
    const int b = blockIdx.x;
    double s[2] = {0.0, 0.0};
    for (int t = threadIdx.x; t < ntiles; t += WG) {
        const double *p = partials + ((size_t)b * ntiles + t) * 2;
        s[0] = s[0] + p[0];
        s[1] = s[1] + p[1];
    }
    double out[2];
    block_reduce<2>(s, s_red, out);
    if (threadIdx.x == 0) S[b] = out[0] + out[1];
}

template <int MODE, int WGS, int K>
void launch_k(const LoglikeArgs &a, bool write_model, int grid, hipStream_t st) {
    if (write_model) hipLaunchKernelGGL((k_loglike<MODE, WGS, K, true>), dim3(grid), dim3(WGS), 0, st, a);
    else hipLaunchKernelGGL((k_loglike<MODE, WGS, K, false>), dim3(grid), dim3(WGS), 0, st, a);
}

template <int MODE>
bool launch_geom(const LoglikeArgs &a, int wgs, int K, bool write_model, int grid, hipStream_t st) {
    if (wgs == 256) {
        if (K == 1) launch_k<MODE, 256, 1>(a, write_model, grid, st);
        else if (K == 2) launch_k<MODE, 256, 2>(a, write_model, grid, st);
        else if (K == 4) launch_k<MODE, 256, 4>(a, write_model, grid, st);
        else return false;
    } else if (wgs == 64) {  // one wave per tile: no cross-wave barrier, every wave stages / expands / reduces its own tile
        if (K == 4) launch_k<MODE, 64, 4>(a, write_model, grid, st);
        else if (K == 8) launch_k<MODE, 64, 8>(a, write_model, grid, st);
        else if (K == 16) launch_k<MODE, 64, 16>(a, write_model, grid, st);
        else return false;
    } else return false;
    return true;
}

}  // namespace

int tile_bins(int wgs, int K) { return wgs * K; }
bool valid_geometry(int wgs, int K) {
    return (wgs == 256 && (K == 1 || K == 2 || K == 4)) || (wgs == 64 && (K == 4 || K == 8 || K == 16));
}

hipError_t launch_loglike(LoglikeArgs a, int mode, int wgs, int K, bool write_model, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    if (!valid_geometry(wgs, K)) return hipErrorInvalidValue;
    const int tb = wgs * K;
    a.ntiles = (a.Nx + tb - 1) / tb;
    const int ntiles_pad = ((a.ntiles + 7) / 8) * 8;
    const long long grid = (long long)ntiles_pad * a.B;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.tile_rot < 0 || a.tile_rot >= a.ntiles) a.tile_rot = 0;
    bool ok;
    if (mode == M_FAST) ok = launch_geom<M_FAST>(a, wgs, K, write_model, (int)grid, st);
    else if (mode == M_FAST_DIRECT) ok = launch_geom<M_FAST_DIRECT>(a, wgs, K, write_model, (int)grid, st);
    else ok = launch_geom<M_STRICT>(a, wgs, K, write_model, (int)grid, st);
    if (!ok) return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int MODE>
bool launch_geom_delta(const LoglikeArgs &a, int wgs, int K, int grid, hipStream_t st) {
    if (wgs == 256 && K == 4) hipLaunchKernelGGL((k_loglike<MODE, 256, 4, false, true>), dim3(grid), dim3(256), 0, st, a);
    else if (wgs == 64 && K == 8) hipLaunchKernelGGL((k_loglike<MODE, 64, 8, false, true>), dim3(grid), dim3(64), 0, st, a);
    else if (wgs == 64 && K == 4) hipLaunchKernelGGL((k_loglike<MODE, 64, 4, false, true>), dim3(grid), dim3(64), 0, st, a);
    else return false;
    return true;
}

bool delta_geometry(int wgs, int K) { return (wgs == 256 && K == 4) || (wgs == 64 && (K == 8 || K == 4)); }

hipError_t launch_loglike_delta(LoglikeArgs a, int mode, int wgs, int K, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    if (!delta_geometry(wgs, K) || mode == M_STRICT || !a.d_range || !a.d_flags || !a.d_noise_old || !a.d_row || !a.model0) return hipErrorInvalidValue;
    const int tb = wgs * K;
    a.ntiles = (a.Nx + tb - 1) / tb;
    const int ntiles_pad = ((a.ntiles + 7) / 8) * 8;
    const long long grid = (long long)ntiles_pad * a.B;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.tile_rot < 0 || a.tile_rot >= a.ntiles) a.tile_rot = 0;
    const bool ok = (mode == M_FAST) ? launch_geom_delta<M_FAST>(a, wgs, K, (int)grid, st) : launch_geom_delta<M_FAST_DIRECT>(a, wgs, K, (int)grid, st);
    if (!ok) return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_bg_poly(const LoglikeArgs &a, int wgs, int K, double *bg, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    if (!valid_geometry(wgs, K) || !bg) return hipErrorInvalidValue;
    const int tb = wgs * K;
    const int ntiles = (a.Nx + tb - 1) / tb;
    const long n = (long)a.B * ntiles;
    hipLaunchKernelGGL(k_bg_poly, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, st, a.noise, a.noise_stride, a.nharvey, a.nnoise, a.B, ntiles, tb,
                       a.x0, a.step, bg);
    return hipGetLastError();
}

hipError_t launch_finalize(const double *partials, int B, int ntiles, double *S, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize, dim3(B), dim3(WG), 0, st, partials, ntiles, S);
    return hipGetLastError();
}

}  // namespace tamcmc
