// loglike_tile.h -- the tile body of the likelihood kernel (see kernels.hip for the mapping and the arithmetic modes), as a device
// function shared by k_loglike (kernels.hip) and the fused sampler step (dev_sampler.hip): one workgroup = one tile of WGS*K
// consecutive bins of ONE evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "bg_series.h"

namespace tamcmc {
namespace tile {

constexpr int CHUNK = 64;  // multiplets staged per pass: one wave compacts one chunk

constexpr int F_FULL = 1;  // window covers every bin of the tile
constexpr int F_SAFE = 2;  // FAST: product of the 2l+1 denominators stays far below DBL_MAX on this tile
constexpr int F_ASYM = 4;  // asymmetry coefficient != 0
constexpr int F_FAR = 8;   // FAST far field: the multiplet joins the tile polynomial instead of the per-bin loop

// arithmetic modes of k_loglike (see include/tamcmc_hip.h)
constexpr int M_STRICT = 0, M_FAST_DIRECT = 2, M_FAST = 1;

// FAST far field: a multiplet whose window covers the whole tile and whose components all sit at least 1/RHO_MAX tile
// half-widths away from the tile centre is an analytic function of x on the tile; the sum of ALL such multiplets is ONE
// polynomial of degree NC-1 in s = (x - x_c)/h, evaluated per bin by Horner (NC-1 fma) instead of ~6 ops per component.
// Component: hv/(1+(beta s - A)^2) = sum_k c_k s^k, c_k = -hv Im(u q^k), u = 1/(A+i), q = beta u, |q| = rho, with the
// three-term recurrence c_{k+1} = 2 Re(q) c_k - |q|^2 c_{k-1}.  Truncation error <= rho^NC/(1-rho) of the (small) far term.
constexpr int NC = 16;
constexpr double RHO_MAX2 = 1.0 / 36.0;  // rho <= 1/6 -> truncation <= 6^-16/(1-1/6) = 4e-13 of the far term (itself <~ 0.3 M)
constexpr double RHO_MAX2_ASYM = 1.0 / 64.0;  // asymmetric profiles: the far wing can dominate M and the quadratic
                                              // factor feeds degree >= NC terms back -> rho <= 1/8 (3.6e-15)
constexpr int NH = bg::NH;               // Taylor coefficients of the background on a tile (bg_series.h)
constexpr int ROW = NC + 2;  // LDS row stride of the coefficient reduction (16-byte aligned, conflict-free b128 writes)

// LDS image of a multiplet (160 B, every field group 16-byte aligned for ds_read_b128).
struct __attribute__((aligned(16))) LdsMult {
    int i0, i1, l, flags;
    double g;     // STRICT: gamma^2          FAST: 2/gamma
    double asym;  // asymmetry coefficient
    double c2sq;  // (0.5*gamma*asym/fc)^2
    double fcx;   // STRICT: nu_c             FAST: asym/nu_c
    double2 nh[7];  // STRICT: {nu_nlm, H*V_m}   FAST: {A_m = (nu_nlm - x_c) 2/gamma, H*V_m}  (x_c = centre of the nominal tile)
};

// v_rcp_f64 seed (2^-24.4) + ONE Newton-Raphson step: relative error <= 2.1e-15
__device__ __forceinline__ double rcp_nr1(double d) {
    double r = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, r, 1.0), r, r);
}
// two steps: ~1 ulp
__device__ __forceinline__ double rcp_nr2(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

template <int NV, int WGS = 256>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double *s_red, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] = v[i] + __shfl_down(v[i], off, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) s_red[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            double s = s_red[i];
            for (int w = 1; w < WGS / 64; w++) s = s + s_red[w * NV + i];
            out[i] = s;
        }
    }
}

// ---- STRICT: the reference's statement sequence per bin, one IEEE operation per statement ----
template <int NM, int K, bool FULL>
__device__ __forceinline__ void strict_mult(const LdsMult &M, const double (&xv)[K], const int (&bin)[K], double (&acc)[K]) {
    double nu[NM], hv[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) {
        const double2 p = M.nh[m];
        nu[m] = p.x;
        hv[m] = p.y;
    }
    const double g2 = M.g;
    const int i0 = M.i0, i1 = M.i1;
    if (!(M.flags & F_ASYM)) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (FULL || (bin[k] >= i0 && bin[k] < i1)) {
                double res = 0.0;
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    const double d = xv[k] - nu[m];
                    double p = d * d;
                    p = 4.0 * p / g2;
                    const double inv = 1.0 / (1.0 + p);
                    res = res + hv[m] * inv;
                }
                acc[k] = acc[k] + res;
            }
        }
    } else {
        const double asym = M.asym, fc = M.fcx, c2sq = M.c2sq;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (FULL || (bin[k] >= i0 && bin[k] < i1)) {
                const double t = 1.0 + asym * (xv[k] / fc - 1.0);
                const double asy = t * t + c2sq;
                double res = 0.0;
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    const double d = xv[k] - nu[m];
                    double p = d * d;
                    p = 4.0 * p / g2;
                    const double inv = 1.0 / (1.0 + p);
                    res = res + hv[m] * (asy * inv);
                }
                acc[k] = acc[k] + res;
            }
        }
    }
}

// ---- FAST: sum_m hv_m/q_m over a common denominator: 3 ops per component for q_m, 3 for (N,D),
//      ONE reciprocal (+1 Newton step) per multiplet per bin ----
//      The argument of component m is t = (x - nu_m) 2/gamma = (x - x_c) 2/gamma - A_m with A_m staged per tile: one fma per
//      component from the bin's offset to the tile centre (|x - x_c| <= half a tile, so the fma's rounding is ~1e-15 of t's scale).
template <int NM, int K, bool FULL>
__device__ __forceinline__ void fast_mult(const LdsMult &M, const double (&xv)[K], const int (&bin)[K], double (&acc)[K], double xc) {
    double nA[NM], hv[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) {
        const double2 p = M.nh[m];
        nA[m] = -p.x;
        hv[m] = p.y;
    }
    const double g = M.g;
    const int i0 = M.i0, i1 = M.i1;
    const int flags = M.flags;
    if (flags & F_SAFE) {
        const bool has_asym = flags & F_ASYM;
        const double afc = M.fcx, c2sq = M.c2sq, one_m_asym = 1.0 - M.asym;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (FULL || (bin[k] >= i0 && bin[k] < i1)) {
                const double xx = xv[k];
                const double dx = xx - xc;
                double t = fma(dx, g, nA[0]);
                double D = fma(t, t, 1.0);
                double N = hv[0];
#pragma unroll
                for (int m = 1; m < NM; m++) {
                    t = fma(dx, g, nA[m]);
                    const double q = fma(t, t, 1.0);
                    N = fma(N, q, hv[m] * D);
                    D = D * q;
                }
                double res = N * rcp_nr1(D);
                if (has_asym) {
                    const double ta = fma(afc, xx, one_m_asym);
                    res = res * fma(ta, ta, c2sq);
                }
                acc[k] = acc[k] + res;
            }
            if (K > 4 && (k & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the live ranges: four bins in flight
        }
    } else {  // denominators too large to multiply on this tile: plain sum (never taken for sane widths)
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (FULL || (bin[k] >= i0 && bin[k] < i1)) {
                const double xx = xv[k];
                const double dx = xx - xc;
                double res = 0.0;
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    const double t = fma(dx, g, nA[m]);
                    res = res + hv[m] / fma(t, t, 1.0);
                }
                if (flags & F_ASYM) {
                    const double ta = fma(M.fcx, xx, 1.0 - M.asym);
                    res = res * fma(ta, ta, M.c2sq);
                }
                acc[k] = acc[k] + res;
            }
        }
    }
}

template <bool FAST, int NM, int K>
__device__ __forceinline__ void mult_dispatch(const LdsMult &M, const double (&xv)[K], const int (&bin)[K], double (&acc)[K], double xc) {
    if (M.flags & F_FULL) {
        if (FAST) fast_mult<NM, K, true>(M, xv, bin, acc, xc);
        else strict_mult<NM, K, true>(M, xv, bin, acc);
    } else {
        if (FAST) fast_mult<NM, K, false>(M, xv, bin, acc, xc);
        else strict_mult<NM, K, false>(M, xv, bin, acc);
    }
}


// The workgroup's LDS (one object per workgroup, declared by the kernel: a fused kernel overlays it with its other roles' scratch).
template <int MODE, int WGS>
struct __attribute__((aligned(16))) TileLds {
    static constexpr bool FARFIELD = (MODE == M_FAST);
    // multiplet list and (after the multiplet loop) the coefficient-reduction rows share one region
    static constexpr int LDS_BYTES = FARFIELD ? (WGS * ROW * 8 > CHUNK * (int)sizeof(LdsMult) ? WGS * ROW * 8 : CHUNK * (int)sizeof(LdsMult))
                                              : CHUNK * (int)sizeof(LdsMult);
    unsigned char buf[LDS_BYTES];
    double coef[NC];               // FARFIELD: the tile's far-field polynomial
    double part[WGS / 16][NC];
    double red[2 * (WGS / 64)];
    double lt[TAMCMC_MAX_HARVEY];  // FAST: ln(1e-3*tau_k)
    double lto[TAMCMC_MAX_HARVEY]; // DELTA: ln(1e-3*tau_k) of the base point
    int n, nfar, anyfar;
    unsigned short slot[CHUNK * 7];  // FARFIELD: this chunk's far COMPONENTS, packed: (position in the list) << 3 | m
};

// One tile of one evaluation: table slot `sb` (its multiplets, noise row, background series), result row `b` (partials / model).
// COH: the two partial sums are written through to memory (device-scope stores) because another workgroup of the SAME launch reads
// them (the fused sampler step's settle tail); MI355X has one L2 per XCD and plain stores stay in the writer's.
template <int MODE, int WGS, int K, bool WRITE_MODEL, bool DELTA, bool COH = false>
__device__ __forceinline__ void tile_compute(const LoglikeArgs &a, const int tile, const int b, const int sb, TileLds<MODE, WGS> &S, const int mbeg,
                                             const int mend, const int nh, const int nn) {
    constexpr bool FAST = (MODE != M_STRICT);
    static_assert(!DELTA || (FAST && !WRITE_MODEL), "DELTA launches are FAST-mode, logL-only");
    constexpr bool FARFIELD = (MODE == M_FAST);
    LdsMult *s_m = (LdsMult *)S.buf;
    double *s_rows = (double *)S.buf;
    int &s_n = S.n, &s_nfar = S.nfar, &s_anyfar = S.anyfar;
    unsigned short *s_slot = S.slot;
    double *s_coef = S.coef;
    double(*s_part)[NC] = S.part;
    double *s_red = S.red;
    double *s_lt = S.lt, *s_lto = S.lto;
    const int tid = threadIdx.x;
    // Timing instrumentation exists in the probe build only (-DTAMCMC_PROBE, tools/): the product library has no code path that can
    // skip a phase or return a wrong log-likelihood.
#ifdef TAMCMC_PROBE
#define KSTAMP(k) do { if (a.dbg && b == 0 && tile == a.ntiles / 2 && tid == 0) a.dbg[k] = (long)wall_clock64(); } while (0)
#define PROBE_SKIP(bit) (a.probe & (bit))
    const long wg_t0 = (a.dbg && a.dbg[7] == 77) ? (long)wall_clock64() : 0;  // per-workgroup timeline
#else
#define KSTAMP(k) do { } while (0)
#define PROBE_SKIP(bit) false
#endif
    KSTAMP(0);
    constexpr int TILE = WGS * K;
    const int t0 = tile * TILE;
    const int t1 = min(t0 + TILE, a.Nx);
    if (DELTA) {  // tiles outside the affected bin range contribute exactly 0 (workgroup-uniform exit before any barrier)
        if (a.d_done && a.d_done[(size_t)b * a.ntiles + tile]) return;  // a far-only tile of a light evaluation: k_fd_far (kernels.hip) did it
        const int lo = a.d_range[2 * b], hi = a.d_range[2 * b + 1];
        if (t1 <= lo || t0 >= hi) {
            if (tid == 0) {
                double *p = a.partials + ((size_t)b * a.ntiles + tile) * 2;
                p[0] = 0.0;
                p[1] = 0.0;
            }
            return;
        }
    }

    tamcmc_multiplet g;
    // DELTA, bit 1 of the evaluation's flags ("full table"): a perturbation that moves most multiplets (a splitting coefficient, the
    // asymmetry) is cheaper as the WHOLE perturbed model minus the base model row M0 (third plane of model0) than as +new / -old row
    // pairs -- per rows instead of up to 2 per.  The table then holds every row of the perturbed point, the background is the base
    // point's (the noise parameters did not change: its prebuilt series, a.bg_poly rows by base point), and dM = M - M0 per bin.  The
    // unchanged rows and the background are summed by the same code in the same order as in the base launch: they cancel exactly.
    const bool fullnew = DELTA && (a.d_flags[b] & 2);
    const bool dsub = DELTA && !fullnew;  // the evaluation carries -old rows / the old noise row: differences are formed term by term
    // prebuilt background series of this (evaluation, tile), one coefficient per lane: it stays in those lanes' registers and enters the
    // tile polynomial where the far-field sums are closed (no wait for it here, no trip through LDS)
    const bool bg_prebuilt = FARFIELD && a.bg_poly && (!DELTA || fullnew);
    double bg_pre = 0.0;
    if (bg_prebuilt && tid < NH) bg_pre = a.bg_poly[((size_t)(DELTA ? a.d_row[b] : sb) * a.ntiles + tile) * NH + tid];
    double xv[K], yv[K], acc[K];
    int bin[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        bin[k] = t0 + k * WGS + tid;
        const int bi = min(bin[k], a.Nx - 1);
        xv[k] = a.x[bi];
        acc[k] = 0.0;
    }

    const double *nz = a.noise + (size_t)sb * a.noise_stride;
    // DELTA: background difference only when the noise parameters changed (same Harvey count on both sides)
    const bool bg = !DELTA || fullnew || (a.d_flags[b] & 1);
    const double *nzo = DELTA ? a.d_noise_old + (size_t)b * a.noise_stride : nz;
    // tile geometry for the far field: centre and half-width of the nominal tile on the regular grid
    const double h = 0.5 * (double)TILE * a.step;
    const double xc = a.x0 + ((double)t0 + 0.5 * (double)TILE - 0.5) * a.step;
    // FARFIELD: the background H/(1+(a x)^p) + N0 is analytic on the tile with its singularities ~x_c away, so it joins the
    // tile polynomial: u(s) = (a x_c)^p (1+eps s)^p (binomial series), then the reciprocal series of 1+u.
    const bool harvey_poly = FARFIELD && bg && bg::series_valid(xc, h);
    if (FAST && bg && !harvey_poly) {
        if (tid < nh) {
            s_lt[tid] = log(1e-3 * nz[3 * tid + 1]);
            if (dsub) s_lto[tid] = log(1e-3 * nzo[3 * tid + 1]);
        }
    }
    if (FARFIELD) {
        if (tid < NC) s_coef[tid] = 0.0;
        if (tid == 0) s_anyfar = harvey_poly ? 1 : 0;
        __syncthreads();
        if (harvey_poly && !bg_prebuilt) {
            // lane = Harvey term (wave 1 when there are four waves: wave 0 is about to compact the first chunk); the NH
            // series coefficients of the terms are added to the tile polynomial in term order by the last of these lanes
            const int hl = (WGS > 64) ? tid - 64 : tid;
            double f[NH];
#pragma unroll
            for (int k = 0; k < NH; k++) f[k] = 0.0;
            const bool lane_new = (hl >= 0 && hl < nh), lane_old = dsub && (hl >= 32 && hl < 32 + nh);
            if (lane_new || lane_old) {
                const double *nq = lane_new ? nz : nzo;
                const int ht = lane_new ? hl : hl - 32;
                bg::harvey_term_series(lane_new ? nq[3 * ht] : -nq[3 * ht], nq[3 * ht + 1], nq[3 * ht + 2], xc, h, f);
            }
            // lanes hl = 0..nh-1 live in ONE wave: sum their series in lane order with shuffles, lane 0 adds the white noise
            if (hl >= 0 && hl < 64) {
#pragma unroll
                for (int k = 0; k < NH; k++) {
                    double v = f[k];
                    double tot = 0.0;
                    for (int t = 0; t < nh; t++) tot = tot + __shfl(v, t, 64);
                    if (dsub)
                        for (int t = 0; t < nh; t++) tot = tot + __shfl(v, 32 + t, 64);
                    const double wn = dsub ? (nz[nn - 1] - nzo[nn - 1]) : nz[nn - 1];
                    if (hl == 0) s_coef[k] = tot + (k == 0 ? wn : 0.0);
                }
            }
        }
    }

    // the table builder already summed the series of this (evaluation, tile) (bg_series.h, same arithmetic): still in registers
    bool bg_in_regs = harvey_poly && bg_prebuilt;
    bool any_near = false;  // (workgroup-uniform) a chunk held a multiplet of the near field
    KSTAMP(1);
    for (int c0 = mbeg; c0 < mend; c0 += CHUNK) {
        __syncthreads();  // previous chunk fully consumed
        if (tid < 64) {
            const int idx = c0 + tid;
            // the whole row in ONE round trip (clamped index: lanes past the end stage nothing); with a.per its address does not wait for the range
            g = a.mults[a.per > 0 ? min(idx, mbeg + a.per - 1) : min(idx, mend - 1)];
            const int i0 = g.i0, i1 = g.i1;
            const bool ov = (idx < mend) && (i0 < t1) && (i1 > t0) && !PROBE_SKIP(16);
            // every overlapping multiplet is staged with its per-multiplet scalars hoisted: the NEAR ones first (in table order: the
            // per-bin loop below walks them without looking at a flag), the far ones -- they only feed the tile polynomial -- behind
            // them.  Pass 1 decides near / far (its temporaries die at the ballot), pass 2 builds the LDS image in place.
            const int l = g.l;
            const int nm = 2 * l + 1;
            const bool full = (i0 <= t0 && i1 >= t1);
            bool far = false;
            if (FARFIELD && ov && full) {
                const double ig = 2.0 * rcp_nr2(g.gamma);
                const double beta2 = (ig * h) * (ig * h);
                const double r2 = (g.asym != 0.0) ? RHO_MAX2_ASYM : RHO_MAX2;
                far = true;
#pragma unroll
                for (int m = 0; m < 7; m++) {
                    const double A = ig * (g.nu[m] - xc);
                    if (m < nm && !(beta2 <= r2 * fma(A, A, 1.0))) far = false;  // rho^2 = beta^2/(A^2+1); also rejects NaN
                }
            }
            const unsigned long long mask = __ballot(ov);
            const unsigned long long fmask = FARFIELD ? __ballot(far) : 0ull;
            const unsigned long long lt = (1ull << tid) - 1ull;
            const int n_near = __popcll(mask & ~fmask);
            const int pos = far ? n_near + __popcll(fmask & lt) : __popcll(mask & ~fmask & lt);
            if (ov) {
                LdsMult &d = s_m[pos];
                int flags = 0;
                if (full) flags |= F_FULL;
                if (g.asym != 0.0) flags |= F_ASYM;
                // FAST: reciprocal + Newton steps instead of the two IEEE divides by nu_c (exact zeros when asym = 0 either way)
                const double ifc = FAST ? rcp_nr2(g.fc) : 0.0;
                const double c2 = FAST ? 0.5 * g.gamma * g.asym * ifc : 0.5 * g.gamma * g.asym / g.fc;
                d.c2sq = c2 * c2;
                d.asym = g.asym;
                double Am[7];
                if (FAST) {
                    const double ig = 2.0 * rcp_nr2(g.gamma);
                    d.g = ig;
                    d.fcx = g.asym * ifc;
#pragma unroll
                    for (int m = 0; m < 7; m++) Am[m] = ig * (g.nu[m] - xc);  // constant trip count: g stays in registers
                    if (far) flags |= F_FAR;
                    else {
                        // prod_m (1 + ((x-nu_m) ig)^2) < (1e38)^7 = 1e266 on the whole tile?
                        const double bh = ig * h;  // |t| <= |A_m| + beta on the nominal tile (a coarse bound)
                        bool safe = true;
#pragma unroll
                        for (int m = 0; m < 7; m++) {
                            const double dm = fabs(Am[m]) + bh;
                            if (m < nm && !(fma(dm, dm, 1.0) < 1e38)) safe = false;  // also false for NaN/inf inputs
                        }
                        if (safe) flags |= F_SAFE;
                    }
                } else {
                    d.g = g.gamma * g.gamma;
                    d.fcx = g.fc;
                }
                d.i0 = i0; d.i1 = i1; d.l = l; d.flags = flags;
#pragma unroll
                for (int m = 0; m < 7; m++) d.nh[m] = make_double2(FAST ? Am[m] : g.nu[m], g.hv[m]);
            }
            if (FARFIELD) {
                // far components packed densely (no idle lanes for l < 3): offset = components of the far multiplets before this lane
                const int lv = ov ? g.l : 0;
                const unsigned long long f0 = __ballot(far && lv == 0), f1 = __ballot(far && lv == 1), f2 = __ballot(far && lv == 2),
                                         f3 = __ballot(far && lv >= 3);
                if (far) {
                    const int off = __popcll(f0 & lt) + 3 * __popcll(f1 & lt) + 5 * __popcll(f2 & lt) + 7 * __popcll(f3 & lt);
                    const int nmf = 2 * (lv > 3 ? 3 : lv) + 1;
                    for (int m = 0; m < nmf; m++) s_slot[off + m] = (unsigned short)((pos << 3) | m);
                }
                if (tid == 0) {
                    s_nfar = __popcll(f0) + 3 * __popcll(f1) + 5 * __popcll(f2) + 7 * __popcll(f3);  // far components of the chunk
                    if (f0 | f1 | f2 | f3) s_anyfar = 1;
                }
            }
            if (tid == 0) s_n = n_near;
        }
        __syncthreads();
        KSTAMP(2);
        const int n = PROBE_SKIP(1) ? 0 : s_n;
        any_near = any_near || (s_n > 0);
        for (int q = 0; q < n; q++) {
            const LdsMult &M = s_m[q];  // (near multiplets only: the staging pass put them first)
            switch (M.l) {  // wave-uniform
            case 0: mult_dispatch<FAST, 1, K>(M, xv, bin, acc, xc); break;
            case 1: mult_dispatch<FAST, 3, K>(M, xv, bin, acc, xc); break;
            case 2: mult_dispatch<FAST, 5, K>(M, xv, bin, acc, xc); break;
            default: mult_dispatch<FAST, 7, K>(M, xv, bin, acc, xc); break;
            }
        }
        KSTAMP(3);
        if (FARFIELD && s_nfar > 0 && !PROBE_SKIP(2)) {  // workgroup-uniform
            // AFTER the near-field loop (its registers are dead): one lane per (far multiplet, m) slot computes the NC Taylor
            // coefficients of its component; the lanes' vectors are summed in a fixed order into the tile polynomial
            double fcoef[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) fcoef[k] = 0.0;
            const int nslots = s_nfar;
            for (int slot = tid; slot < nslots; slot += WGS) {
                const int e = s_slot[slot];
                const LdsMult &M = s_m[e >> 3];
                {
                    const double2 nhm = M.nh[e & 7];
                    const double beta = M.g * h;
                    const double A = nhm.x;
                    const double inv = rcp_nr2(fma(A, A, 1.0));
                    const double two_req = 2.0 * beta * A * inv, q2 = beta * beta * inv;
                    double cm = nhm.y * inv;      // c_0
                    double cc = cm * two_req;     // c_1
                    if (!(M.flags & F_ASYM)) {
                        fcoef[0] = fcoef[0] + cm;
                        fcoef[1] = fcoef[1] + cc;
                        // terms beyond rho^n <= 1e-13 are dropped (q2 = rho^2); the loop length is the wave's longest
                        const int nt = (q2 > 1.39e-2) ? 16 : (q2 > 6.8e-3) ? 14 : (q2 > 2.5e-3) ? 12 : (q2 > 5.6e-4) ? 10 : (q2 > 4.6e-5) ? 8 : 6;
#pragma unroll
                        for (int k = 2; k < NC; k++) {
                            if ((k & 1) == 0 && k >= 6 && !__any(k < nt)) break;  // wave-uniform
                            const double cn = fma(two_req, cc, -q2 * cm);
                            fcoef[k] = fcoef[k] + cn;
                            cm = cc;
                            cc = cn;
                        }
                    } else {
                        // times the asymmetry factor (1+asym(x/nu_c-1))^2 + c2^2 = A0 + A1 s + A2 s^2 (M.fcx = asym/nu_c)
                        const double p0 = fma(M.fcx, xc, 1.0 - M.asym), p1 = M.fcx * h;
                        const double A0 = fma(p0, p0, M.c2sq), A1 = 2.0 * p0 * p1, A2 = p1 * p1;
                        double c2 = 0.0, c1 = 0.0, c0k = cm;  // c_{k-2}, c_{k-1}, c_k
                        double nxt = cc;
#pragma unroll
                        for (int k = 0; k < NC; k++) {
                            fcoef[k] = fcoef[k] + fma(A0, c0k, fma(A1, c1, A2 * c2));
                            const double cn = (k == 0) ? nxt : fma(two_req, c0k, -q2 * c1);
                            c2 = c1;
                            c1 = c0k;
                            c0k = cn;
                        }
                    }
                }
            }
            __syncthreads();  // every lane is done with the multiplet list: its LDS region now holds the reduction rows
#pragma unroll
            for (int k = 0; k < NC; k += 2) *(double2 *)&s_rows[tid * ROW + k] = make_double2(fcoef[k], fcoef[k + 1]);
            __syncthreads();
            {
                const int k = tid & 15, part = tid >> 4;
                double sum = 0.0;
#pragma unroll
                for (int r = 0; r < 16; r++) sum = sum + s_rows[(part * 16 + r) * ROW + k];
                s_part[part][k] = sum;
            }
            __syncthreads();
            if (tid < NC) {
                double sum = (bg_in_regs && tid < NH) ? bg_pre : s_coef[tid];  // background first, then the parts: the order of the sum
#pragma unroll
                for (int p = 0; p < WGS / 16; p++) sum = sum + s_part[p][tid];
                s_coef[tid] = sum;
            }
            bg_in_regs = false;
        }
    }
    if (bg_in_regs && tid < NH) s_coef[tid] = bg_pre;  // (no far multiplet on this tile: the polynomial is the background alone)
    KSTAMP(4);
    // the power values: issued before the polynomial evaluation, consumed after it
#pragma unroll
    for (int k = 0; k < K; k++) yv[k] = DELTA ? 0.0 : a.y[min(bin[k], a.Nx - 1)];  // (DELTA: y enters through the base point's y/M0 plane)
    if (FAST) __syncthreads();  // s_lt / s_coef visible (also when the evaluation has no multiplet chunk)
    if constexpr (DELTA && FARFIELD && WGS == 64) {
        // A tile whose changed multiplets are ALL in its far field (most tiles of a perturbed frequency's window: the mode is near for
        // three or four of a hundred) has dM = P(s), the tile polynomial, with |u| = |dM / M0| ~ 1e-7 of a far wing: the change of its
        // likelihood terms, sum_b [(1 - y/M0) u - (1/2 - y/M0) u^2], is a dot product of the polynomial's coefficients (and of their
        // self-convolution) with moments of the base point on this tile (k_fd_moments, kernels.hip) -- no bin is walked.  Omitted:
        // u^3, below 1e-10 of the leading term when max|u| <= 1e-5 (bounded by sum|c_k| max(1/M0)); a tile beyond that walks its bins.
        // (a perturbed noise parameter has no rows: its dM is the difference of two background series, a polynomial wherever the series is valid)
        if (a.fd_mom && (!bg || harvey_poly) && !fullnew && !any_near) {  // workgroup-uniform
            const double *mm = a.fd_mom + ((size_t)a.d_row[b] * a.ntiles + tile) * FD_MOM;
            const double w1 = (tid < NC) ? mm[tid] : 0.0, w2 = (tid < 2 * NC - 1) ? mm[NC + tid] : 0.0, rmax = mm[FD_MOM - 1];
            const double ck = (tid < NC) ? s_coef[tid] : 0.0;
            double ab = fabs(ck);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) ab = ab + __shfl_xor(ab, off, 64);
            if (ab * rmax <= 1e-5) {  // (false for NaN)
                double cv = 0.0;  // (c * c)_tid
                if (tid < 2 * NC - 1) {
                    const int j0 = tid < NC ? 0 : tid - (NC - 1), j1 = tid < NC ? tid : NC - 1;
                    for (int j = j0; j <= j1; j++) cv = fma(s_coef[j], s_coef[tid - j], cv);
                }
                double tot = fma(ck, w1, cv * w2);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot = tot + __shfl_xor(tot, off, 64);
                if (tid == 0) {
                    double *p = a.partials + ((size_t)b * a.ntiles + tile) * 2;
                    p[0] = tot;
                    p[1] = 0.0;
                }
                return;
            }
        }
    }
    if (FARFIELD) {
        if (s_anyfar && !PROBE_SKIP(4)) {  // workgroup-uniform: far multiplets and/or the background series
            const double inv_h = 1.0 / h;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const double sx = (xv[k] - xc) * inv_h;
                double P = s_coef[NC - 1];
#pragma unroll
                for (int q = NC - 2; q >= 0; q--) P = fma(P, sx, s_coef[q]);
                acc[k] = acc[k] + P;
            }
        }
    }

    KSTAMP(5);
    // ---- background + likelihood terms ----
    double s[2] = {0.0, 0.0};
    const double white = nz[nn - 1];
    // FAST: the thread's K bins share ONE logarithm and ONE reciprocal: sum_k ln M_k = ln prod_k M_k and sum_k y_k/M_k = N / prod_k M_k with
    // N <- N M_k + y_k D built beside the product (three instructions per bin instead of a reciprocal with two Newton steps)
    double prod = 1.0, ynum = 0.0;
    double Mk[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        double Mv = acc[k];
        const int bi = min(bin[k], a.Nx - 1);
        const bool valid = bin[k] < a.Nx;
        if (!FAST) {
            for (int hh = 0; hh < nh; hh++) {
                const double tau = nz[3 * hh + 1];
                if (tau != 0.0) {
                    double t = pow((1e-3 * tau) * xv[k], nz[3 * hh + 2]);
                    t = nz[3 * hh] * (1.0 / (t + 1.0));
                    Mv = Mv + t;
                }
            }
            Mv = Mv + white;
            if (valid) {
                s[0] = s[0] + yv[k] * (1.0 / Mv);
                s[1] = s[1] + log(Mv);
            }
        } else {
            if (bg && !harvey_poly) {
                const double lx = a.logx[bi];
                for (int hh = 0; hh < nh; hh++) {
                    const double tau = nz[3 * hh + 1];
                    if (tau != 0.0) {
                        const double t = exp(nz[3 * hh + 2] * (s_lt[hh] + lx));
                        Mv = fma(nz[3 * hh], rcp_nr2(t + 1.0), Mv);
                    }
                    if (dsub) {
                        const double tauo = nzo[3 * hh + 1];
                        if (tauo != 0.0) {
                            const double t = exp(nzo[3 * hh + 2] * (s_lto[hh] + lx));
                            Mv = fma(-nzo[3 * hh], rcp_nr2(t + 1.0), Mv);
                        }
                    }
                }
                Mv = Mv + (dsub ? (white - nzo[nn - 1]) : white);
            }
            if (DELTA) {
                // Mv = dM; u = dM / M0.  Change of the bin's likelihood term y/M + ln M:
                //   y/(M0+dM) - y/M0 + ln(1 + dM/M0) = -(y/M0) u/(1+u) + log1p(u) = sum_n (-1)^n (y/M0 - 1/n) u^n.
                // The base launch left 1/M0 and y/M0 per bin (two planes); a forward-difference step gives |u| ~ 1e-7 .. 1e-3, where five terms
                // of the series are exact to rounding (omitted: u^6, < 1e-10 of the leading term for |u| <= 0.01) and cost a third of two
                // reciprocals and a log1p; a bin beyond that (a step of percent size) takes the closed form -- wave-uniform choice.
                const size_t o = (size_t)a.d_row[b] * a.Nx + (size_t)min(bin[k], a.Nx - 1);
                const double r0 = a.model0[o], yr = a.model0[a.fd_plane + o];
                const double u = (fullnew ? Mv - a.model0[2 * a.fd_plane + o] : Mv) * r0;
                double f;
                if (!__any(valid && !(fabs(u) <= 0.01))) {
                    double pz = fma(u, 0.2 - yr, yr - 0.25);
                    pz = fma(u, pz, (1.0 / 3.0) - yr);
                    pz = fma(u, pz, yr - 0.5);
                    pz = fma(u, pz, 1.0 - yr);
                    f = u * pz;
                } else f = fma(-yr * u, rcp_nr2(1.0 + u), log1p(u));
                if (valid) s[0] = s[0] + f;
            } else if (valid) {
                if (PROBE_SKIP(8)) s[0] = s[0] + yv[k] * Mv;
                else {
                    ynum = fma(ynum, Mv, yv[k] * prod);
                    prod = prod * Mv;
                }
            }
        }
        Mk[k] = Mv;
        if (WRITE_MODEL) {
            if (valid) {
                if (a.fd_rows) {  // base point of a finite-difference batch: what the DELTA launch needs of it (see there)
                    const double r0 = FAST ? rcp_nr2(Mv) : 1.0 / Mv;
                    a.fd_rows[(size_t)b * a.Nx + bin[k]] = r0;
                    a.fd_rows[a.fd_plane + (size_t)b * a.Nx + bin[k]] = yv[k] * r0;
                    a.fd_rows[2 * a.fd_plane + (size_t)b * a.Nx + bin[k]] = Mv;  // (for the "full table" evaluations of the DELTA launch)
                } else a.model[(size_t)b * a.Nx + bin[k]] = Mv;
            }
        }
    }
    if (FAST && !DELTA) {
        if (prod > 1e-280 && prod < 1e280) {
            s[1] = log(prod);
            s[0] = s[0] + ynum * rcp_nr2(prod);
        } else {  // product out of range (or NaN): bin by bin
#pragma unroll
            for (int k = 0; k < K; k++)
                if (bin[k] < a.Nx) {
                    s[1] = s[1] + log(Mk[k]);
                    s[0] = fma(yv[k], rcp_nr2(Mk[k]), s[0]);
                }
        }
    }
    __syncthreads();
    double out[2];
    block_reduce<2, WGS>(s, s_red, out);
    if (tid == 0) {
        double *p = a.partials + ((size_t)b * a.ntiles + tile) * 2;
        if (COH) {
            typedef double __attribute__((address_space(1))) *gdp_t;  // global_store ... sc1 (not flat_)
            __hip_atomic_store((gdp_t)p, out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gdp_t)p + 1, out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            p[0] = out[0];
            p[1] = out[1];
        }
    }
    KSTAMP(6);
#ifdef TAMCMC_PROBE
    if (a.dbg && a.dbg[7] == 77 && tid == 0) {
        long *w = a.dbg + 8 + 2 * ((size_t)b * a.ntiles + tile);
        w[0] = wg_t0;
        w[1] = (long)wall_clock64();
    }
#endif
#undef KSTAMP
#undef PROBE_SKIP
}

// Hook of loglike_tile: which table slot evaluation b reads, and a tail called by every lane of every workgroup that owns a real tile
// after its partial sums are written.
struct NoTail {
    static constexpr bool coherent_partials = false;
    __device__ __forceinline__ int slot(const LoglikeArgs &, int b) const { return b; }  // evaluation b reads table slot b
    __device__ __forceinline__ void operator()(int /*b*/, int /*tile*/, int /*slot*/) const {}
};

// Workgroup `id` of a launch over ntiles x B (tile, evaluation) pairs, XCD-aware: ids id, id+8, id+16, .. share an XCD (round-robin
// dispatch), so all evaluations of one tile are placed on the XCD whose L2 holds that tile's x/y.
// The hook names the table slot of evaluation b (fused sampler step: decided on the device from the previous launch's sums).
template <int MODE, int WGS, int K, bool WRITE_MODEL, bool DELTA, class Tail>
__device__ __forceinline__ void loglike_tile(const LoglikeArgs &a, const int id, TileLds<MODE, WGS> &S, const Tail &tail) {
    const int xcd = id & 7;
    int j = id >> 3, b, tile;
    // launch order = tile_rot, tile_rot+1, ..., wrapping: the caller points tile_rot at the first tile of the mode region so
    // that the long-running tiles (near field) are dispatched first and the cheap far-field-only tiles fill the tail
    if (a.prio_b < 0) {
        b = j % a.B;
        tile = (j / a.B) * 8 + xcd;
    } else {
        // evaluations prio_b and prio_b + 1 lead the launch (fused sampler step: the swap pair's settle is longer than the others';
        // finishing first hides it behind the other chains' tiles); the rest follow in the usual tile-major order
        const int lead = ((a.ntiles + 7) >> 3) * 2;  // (tile group, evaluation) pairs of the two leading evaluations
        if (j < lead) {
            b = a.prio_b + (j & 1);
            tile = (j >> 1) * 8 + xcd;
        } else {
            j -= lead;
            const int rest = a.B - 2, r = j % rest;
            b = r + (r >= a.prio_b ? 2 : 0);
            tile = (j / rest) * 8 + xcd;
        }
    }
    if (tile >= a.ntiles) return;  // padding workgroup: leaves before any barrier
    tile += a.tile_rot;
    if (tile >= a.ntiles) tile -= a.ntiles;
    const int sb = tail.slot(a, b);
    // everything the slot index leads to is requested at once (one memory round trip, not one per dependent step): the table's range,
    // the noise row's lengths; a.per > 0 (device-built tables in fixed-size slots): the range begins at (slot0 + sb) * per
    const int nn = a.nnoise[sb], nh = a.nharvey[sb];
    const int mend = a.offsets[2 * sb + 1];
    const int mbeg = a.per > 0 ? (a.slot0 + sb) * a.per : a.offsets[2 * sb];
    if (nn > 0)  // else: empty evaluation slot (a candidate that was not built, or whose table failed)
        tile_compute<MODE, WGS, K, WRITE_MODEL, DELTA, Tail::coherent_partials>(a, tile, b, sb, S, mbeg, mend, nh, nn);
    tail(b, tile, sb);
}

}  // namespace tile
}  // namespace tamcmc
