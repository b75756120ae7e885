// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the TAMCMC hot path.
//
// k_loglike : fused  model row (sum of truncated Lorentzian multiplets + Harvey background)
//             -> chi^2(2 d.o.f.) terms y/M + ln M -> per-workgroup partial sums
//             (replaces build_l_mode_* build_lorentzian.cpp:131-161,208-246, the windowed adds of
//              optimum_lorentzian_calc_* :441-458,502-522, harvey_like noise_models.cpp:15-39 and
//              likelihood_chi22p likelihoods.cpp:17-28 of the reference, for a whole batch of chains)
// k_finalize: deterministic second-level reduction of the partials, one workgroup per evaluation.
// k_fd_moments / k_fd_far: finite-difference batches (fd_batch.hip) -- moments of the base points per tile, and the far-only tiles of the
//             light delta evaluations taken from them, one lane per tile (no bin walked).
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

// Moments of a base point on one tile, for the DELTA launch's far-only tiles (loglike_tile.h): with u = dM/M0 = P(s)/M0 and P the tile
// polynomial of the changed multiplets, the change of the tile's likelihood terms sum_b [(1 - y/M0) u - (1/2 - y/M0) u^2 + O(u^3)] is
//   sum_k c_k W1_k + sum_m (c*c)_m W2_m,   W1_k = sum_b (1 - y_b/M0_b)/M0_b s_b^k,   W2_m = sum_b (y_b/M0_b - 1/2)/M0_b^2 s_b^m
// (k < 16, m < 31; s = (x - x_c)/h on the nominal tile, as the tile polynomial is evaluated).  One wave per (tile, row); slot 47: max 1/M0.
template <int NJ>  // bins per lane (tile_bins / 64)
__global__ void __launch_bounds__(64) k_fd_moments(const double *x, const double *planes, size_t plane, int Nx, int ntiles, int tile_bins_, double x0,
                                                   double step, double *mom, double *momT) {
    __shared__ double s_m[64][FD_MOM + 1];
    const int tile = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int t0 = tile * tile_bins_;
    const double h = 0.5 * (double)tile_bins_ * step, xc = x0 + ((double)t0 + 0.5 * (double)tile_bins_ - 0.5) * step, inv_h = 1.0 / h;
    double m1[NC], m2[2 * NC - 1], rmax = 0.0;
#pragma unroll
    for (int k = 0; k < NC; k++) m1[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 2 * NC - 1; k++) m2[k] = 0.0;
    // every bin's three values are requested before the first is used (one memory round trip per tile, not one per bin)
    double r0v[NJ], yrv[NJ], xv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int bin = min(t0 + j * 64 + lane, Nx - 1);
        r0v[j] = planes[(size_t)b * Nx + bin];
        yrv[j] = planes[plane + (size_t)b * Nx + bin];
        xv[j] = x[bin];
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        if (t0 + j * 64 + lane >= Nx) continue;
        const double r0 = r0v[j], yr = yrv[j];
        const double sx = (xv[j] - xc) * inv_h;
        const double w1 = (1.0 - yr) * r0, w2 = (yr - 0.5) * r0 * r0;
        rmax = fmax(rmax, fabs(r0));
        double pw = 1.0;
#pragma unroll
        for (int k = 0; k < 2 * NC - 1; k++) {
            if (k < NC) m1[k] = fma(w1, pw, m1[k]);
            m2[k] = fma(w2, pw, m2[k]);
            pw = pw * sx;
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) s_m[lane][k] = m1[k];
#pragma unroll
    for (int k = 0; k < 2 * NC - 1; k++) s_m[lane][NC + k] = m2[k];
    s_m[lane][FD_MOM - 1] = rmax;
    __syncthreads();
    if (lane < FD_MOM) {  // lanes in order: deterministic
        double v = s_m[0][lane];
        if (lane == FD_MOM - 1) {
            for (int r = 1; r < 64; r++) v = fmax(v, s_m[r][lane]);
        } else
            for (int r = 1; r < 64; r++) v = v + s_m[r][lane];
        mom[((size_t)b * ntiles + tile) * FD_MOM + lane] = v;
        momT[((size_t)b * FD_MOM + lane) * ntiles + tile] = v;
    }
}

// The far-only tiles of the light evaluations of a DELTA launch, one LANE per tile (see launch_fd_far, kernels.h).  A tile is taken when
// every row of the evaluation's delta table that overlaps it covers it and lies in its far field -- the staging pass's own criteria
// (loglike_tile.h) -- and its polynomial is small enough for the moment form (sum|c_k| max(1/M0) <= 1e-5); its 16 coefficients are the
// far-field recurrences of tile_compute, summed over the rows in table order, then the dot products with the tile's moments.
constexpr int FAR_ROWS = 16, FAR_WAVES = 4, FAR_WG = 64 * FAR_WAVES;
// grid (evaluations, chunks of 64 tiles); the four waves of a workgroup share the rows of the delta table (row r on wave r mod 4: the
// longest evaluation's chain of 16 rows x 7 components was the kernel's duration), wave 0 adds their coefficient vectors in wave order
__global__ void __launch_bounds__(FAR_WG) k_fd_far(const LoglikeArgs a, const int tile_bins_, unsigned char *done) {
    __shared__ tamcmc_multiplet s_rows[FAR_ROWS];
    __shared__ double s_fc[FAR_WAVES][NC][64];
    __shared__ int s_take[FAR_WAVES][64];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = (int)blockIdx.y * 64 + lane;
    const int mbeg = a.offsets[2 * b], mend = a.offsets[2 * b + 1], nrows = mend - mbeg;
    const int flags = a.d_flags[b], lo = a.d_range[2 * b], hi = a.d_range[2 * b + 1];
    const bool light = nrows > 0 && nrows <= FAR_ROWS && flags == 0 && a.nnoise[b] > 0;
    const int c0 = (int)blockIdx.y * 64 * tile_bins_;  // first bin of the chunk
    if (!light || c0 >= hi || c0 + 64 * tile_bins_ <= lo) {  // (workgroup-uniform) nothing of this chunk is taken here
        if (wave == 0 && tile < a.ntiles) done[(size_t)b * a.ntiles + tile] = 0;
        return;
    }
    {
        static_assert(sizeof(tamcmc_multiplet) % 8 == 0, "copied as doubles");
        const double *src = (const double *)(a.mults + mbeg);
        double *dst = (double *)s_rows;
        for (int i = tid; i < nrows * (int)(sizeof(tamcmc_multiplet) / 8); i += FAR_WG) dst[i] = src[i];
    }
    __syncthreads();
    const double h = 0.5 * (double)tile_bins_ * a.step;
    const int t0 = tile * tile_bins_, t1 = min(t0 + tile_bins_, a.Nx);
    bool take = tile < a.ntiles && !(t1 <= lo || t0 >= hi);
    double fc[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) fc[k] = 0.0;
    const double xc = a.x0 + ((double)t0 + 0.5 * (double)tile_bins_ - 0.5) * a.step;
    for (int r = wave; r < nrows; r += FAR_WAVES) {
        if (!__any(take)) break;  // (wave-uniform)
        const tamcmc_multiplet &g = s_rows[r];  // the same row in every lane (LDS broadcast): degree and asymmetry are wave-uniform
        const int nm = __builtin_amdgcn_readfirstlane(2 * g.l + 1);
        const bool asym = __builtin_amdgcn_readfirstlane(g.asym != 0.0 ? 1 : 0) != 0;
        bool act = take && (g.i0 < t1) && (g.i1 > t0);        // no overlap: contributes nothing to this tile
        if (act && !(g.i0 <= t0 && g.i1 >= t1)) { take = false; act = false; }  // a window edge inside the tile: near field
        const double ig = 2.0 * rcp_nr2(g.gamma), beta = ig * h, beta2 = beta * beta;
        const double r2 = asym ? RHO_MAX2_ASYM : RHO_MAX2;
        double Am[7];
#pragma unroll
        for (int m = 0; m < 7; m++) {
            Am[m] = ig * (g.nu[m] - xc);
            if (act && m < nm && !(beta2 <= r2 * fma(Am[m], Am[m], 1.0))) { take = false; act = false; }
        }
        if (!__any(act)) continue;
        const double ifc = rcp_nr2(g.fc), c2 = 0.5 * g.gamma * g.asym * ifc, c2sq = c2 * c2, fcx = g.asym * ifc;
        const double p0 = fma(fcx, xc, 1.0 - g.asym), p1 = fcx * h;
        const double A0 = fma(p0, p0, c2sq), A1 = 2.0 * p0 * p1, A2 = p1 * p1;
#pragma unroll
        for (int m = 0; m < 7; m++) {
            if (m >= nm) break;  // (uniform)
            const double A = Am[m];
            const double inv = rcp_nr2(fma(A, A, 1.0));
            const double two_req = 2.0 * beta * A * inv, q2 = beta * beta * inv;
            double cm = act ? g.hv[m] * inv : 0.0, cc = cm * two_req;  // (a lane that does not take this row adds zeros)
            if (!asym) {
                fc[0] = fc[0] + cm;
                fc[1] = fc[1] + cc;
#pragma unroll
                for (int k = 2; k < NC; k++) {
                    const double cn = fma(two_req, cc, -q2 * cm);
                    fc[k] = fc[k] + cn;
                    cm = cc;
                    cc = cn;
                }
            } else {
                double c2k = 0.0, c1 = 0.0, c0k = cm;
                const double nxt = cc;
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    fc[k] = fc[k] + fma(A0, c0k, fma(A1, c1, A2 * c2k));
                    const double cn = (k == 0) ? nxt : fma(two_req, c0k, -q2 * c1);
                    c2k = c1;
                    c1 = c0k;
                    c0k = cn;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) s_fc[wave][k][lane] = fc[k];
    s_take[wave][lane] = take ? 1 : 0;
    __syncthreads();
    if (wave != 0 || tile >= a.ntiles) return;
#pragma unroll
    for (int w = 1; w < FAR_WAVES; w++) {
        take = take && s_take[w][lane];
#pragma unroll
        for (int k = 0; k < NC; k++) fc[k] = fc[k] + s_fc[w][k][lane];
    }
    double tot = 0.0;
    if (take) {
        // (moments along the tiles: the lanes of a wave hold consecutive tiles and read consecutive addresses)
        const double *mm = a.fd_momT + (size_t)a.d_row[b] * FD_MOM * a.ntiles + tile;
        const size_t ms = (size_t)a.ntiles;
        double ab = 0.0;
#pragma unroll
        for (int k = 0; k < NC; k++) ab = ab + fabs(fc[k]);
        if (ab * mm[(FD_MOM - 1) * ms] <= 1e-5) {
#pragma unroll
            for (int k = 0; k < NC; k++) tot = fma(fc[k], mm[k * ms], tot);
#pragma unroll
            for (int m = 0; m < 2 * NC - 1; m++) {
                double cv = 0.0;
#pragma unroll
                for (int j = 0; j < NC; j++)
                    if (m - j >= 0 && m - j < NC) cv = fma(fc[j], fc[m - j], cv);
                tot = fma(cv, mm[(NC + m) * ms], tot);
            }
        } else take = false;
    }
    if (take) {
        double *p = a.partials + ((size_t)b * a.ntiles + tile) * 2;
        p[0] = tot;
        p[1] = 0.0;
    }
    done[(size_t)b * a.ntiles + tile] = take ? 1 : 0;
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

hipError_t launch_fd_moments(const LoglikeArgs &a, int wgs, int K, double *mom, double *momT, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    if (!valid_geometry(wgs, K) || !mom || !momT || !a.fd_rows) return hipErrorInvalidValue;
    const int tb = wgs * K;
    const int ntiles = (a.Nx + tb - 1) / tb;
    static_assert(FD_MOM == NC + (2 * NC - 1) + 1, "moment layout");
    if (tb == 256) hipLaunchKernelGGL(k_fd_moments<4>, dim3(ntiles, a.B), dim3(64), 0, st, a.x, a.fd_rows, a.fd_plane, a.Nx, ntiles, tb, a.x0, a.step, mom, momT);
    else if (tb == 512) hipLaunchKernelGGL(k_fd_moments<8>, dim3(ntiles, a.B), dim3(64), 0, st, a.x, a.fd_rows, a.fd_plane, a.Nx, ntiles, tb, a.x0, a.step, mom, momT);
    else if (tb == 1024) hipLaunchKernelGGL(k_fd_moments<16>, dim3(ntiles, a.B), dim3(64), 0, st, a.x, a.fd_rows, a.fd_plane, a.Nx, ntiles, tb, a.x0, a.step, mom, momT);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fd_far(const LoglikeArgs &d, int wgs, int K, unsigned char *done, hipStream_t st) {
    if (d.B <= 0) return hipSuccess;
    if (!delta_geometry(wgs, K) || !done || !d.fd_momT || !d.d_range || !d.d_flags || !d.d_row) return hipErrorInvalidValue;
    LoglikeArgs a = d;
    const int tb = wgs * K;
    a.ntiles = (a.Nx + tb - 1) / tb;
    hipLaunchKernelGGL(k_fd_far, dim3(a.B, (a.ntiles + 63) / 64), dim3(FAR_WG), 0, st, a, tb, done);
    return hipGetLastError();
}

hipError_t launch_finalize(const double *partials, int B, int ntiles, double *S, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize, dim3(B), dim3(WG), 0, st, partials, ntiles, S);
    return hipGetLastError();
}

}  // namespace tamcmc
