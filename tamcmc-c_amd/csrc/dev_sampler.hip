// dev_sampler.hip -- device-resident MCMC iteration (sampler-side algebra on the device).
//
// The host-driven loop (host_mala.cpp) spends ~3/4 of a step on the host (proposal, priors, table build, copies, one sync per
// step).  Here the whole iteration of ALL tempered chains runs on the GPU; the host only enqueues launches and fetches the recorded
// samples once per run() call.  Two launch schemes, same chains bit for bit (same Philox streams, same arithmetic, same orders):
//
// (A) FUSED STEP, one launch per iteration and set of chains (k_step; all chains, or one launch per chain group on two streams once a
//     launch outgrows the GPU's resident waves: see run()) -- used for every stretch of iterations WITHOUT adaptation (the bulk of a run:
//     the reference learns in [Nt_learn[0], Nt_learn[last]) only, config_default.cfg:17-18).  Launch i holds two kinds of 64-lane
//     workgroups:
//       * likelihood tiles of iteration i (loglike_tile.h, the hot kernel's body): chain m's proposal of iteration i is table slot
//         slot[m], chosen by the previous launch.  The LAST tile of a chain to finish (atomic ticket) settles the chain: fixed-order
//         sum of the per-tile partials -> tempered logL -> MH test (MALA.cpp:490-551); for the two chains of the swap pair the second
//         one to finish resolves the parallel-tempering swap (MALA.cpp:397-461); the settled state, the sample/stat record and the
//         slot of the chain's NEXT proposal are written for launch i+1.
//       * branch-ahead candidates of iteration i+1, built WHILE the tiles run: the proposal of i+1 is x + L z(i+1) where x is one of
//         a few known vectors -- the chain's current position (test i rejects) or its proposal of i (accepts), and for the swap pair
//         also the partner's two -- so all 2C+4 candidates (prior, table rows, background series) are prepared in advance by four
//         single-wave roles each (prior | rows | background tiles, two halves).  Nothing but k_loglike's tiles is left on the
//         critical path: an iteration costs one launch of ~C x ntiles tiles plus a short settle tail.
// (B) LOCKSTEP, two kernels per iteration and chain group (k_iterate, k_loglike) -- used where the proposal law is adapted after
//     every test (the next proposal needs the new Cholesky factor, so it cannot be prepared ahead):
//       k_iterate (one workgroup per chain) settles iteration it-1 (MH test, swap, record, Robbins-Monro update MALA.cpp:296-319,
//       Cholesky of (Sigma+eps2 I) sigma MALA.cpp:348-350) and proposes iteration it; k_loglike evaluates.
// All per-iteration state is double-buffered by parity: a workgroup reads parity P and writes parity P^1, so the swap needs no
// inter-workgroup synchronisation inside (B) and a launch never overwrites what it still reads in (A).  Both schemes keep the
// chains' state in the same arrays; a stretch hands over to the next with the parity only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#include "ctx.h"
#include "rgb_prestep.h"
#include "dev_sampler.h"
#include "kernels.h"
#include "loglike_tile.h"
#include "dev_unpack.h"
#include "fd_batch.h"
#include "mode_tables.h"
#include "rng.h"

namespace tamcmc {

namespace {

// value of x in lane LANE (a compile-time constant) for every lane: v_readlane, no LDS round trip like __shfl
template <int LANE>
__device__ __forceinline__ double lane_value(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), LANE), hi = __builtin_amdgcn_readlane((int)(b >> 32), LANE);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a compile-time constant in every copy of its body,
// so that small register arrays indexed by it stay in registers (`#pragma unroll` is a request the optimiser may turn down)
// (the body's call is inlined whatever the caller's size: left as a call, the arrays its lambda captures by reference live in scratch)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_from(F &f) {
    if constexpr (I < N) {
        [[clang::always_inline]] f(std::integral_constant<int, I>{});
        static_for_from<I + 1, N>(f);
    }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_from<0, N>(f);
}

__global__ void k_fill_poly(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
}

constexpr int TB = 256;  // threads of k_iterate (one workgroup per chain)

// Outcome of the Metropolis-Hastings test of chain j for the pending iteration (MALA.cpp:490-551): the values the
// chain holds AFTER the test.
struct AcceptOut {
    int acc;
    double r, logL, logPr, logPost;
};

// MALA.cpp:490-551 for one chain, by ONE lane: S = sum of the chain's per-tile partials, (logPr, status) = the proposal's prior and
// table status, logPost_cur / logL_cur / logPr_cur = what the chain holds.  The same statement sequence serves both launch schemes.
template <class AT>  // AT: DevSamplerArgs, or the same block read through a constant-memory reference (fused settle)
__device__ __forceinline__ AcceptOut mh_outcome(const AT &a, int j, long itp, double S, double logPr, int status, double logL_cur,
                                                double logPr_cur, double logPost_cur, double Tcoef, double init_logL) {
    double logL = (-(double)a.pl * S) / Tcoef;  // call_likelihood, model_def.cpp:399-401
    double logPost;
    if (status != TAMCMC_OK) logL = NAN;
    if (logPr == -INFINITY || isnan(logPr)) { logL = init_logL; logPost = -INFINITY; }  // model_def.cpp:476-480
    else logPost = logL + logPr;
    double u, u1;
    rng_uniform2(a.seed, RNG_ACCEPT, (uint32_t)j, (uint64_t)itp, 0, u, u1);
    double r;
    if (!isnan(logL)) {
        if (logPost == -INFINITY) r = 0.;
        else {
            const double e = exp(logPost - logPost_cur);
            r = fmin(1.0, e);
            if (isnan(r)) r = 0.;
        }
    } else r = 0.;
    AcceptOut o;
    o.acc = (u <= r) ? 1 : 0;
    o.r = r;
    if (o.acc) { o.logL = logL; o.logPr = logPr; o.logPost = logPost; }
    else { o.logL = logL_cur; o.logPr = logPr_cur; o.logPost = logPost_cur; }
    return o;
}

// (B): computed by a whole 256-thread workgroup; every workgroup that needs chain j's outcome (the chain's own workgroup and, in a
// swap step, its partner's) recomputes it from the same inputs -> identical results.
__device__ __forceinline__ void accept_result(const DevSamplerArgs &a, int j, long itp, int P, double *s_red, AcceptOut *s_out) {
    const int tid = threadIdx.x;
    // same reduction order as k_finalize (kernels.hip): strided per-thread sums, shuffle tree, waves in order
    double s1 = 0, s2 = 0;
    for (int t = tid; t < a.ntiles; t += TB) {
        const double *p = a.partials + ((size_t)j * a.ntiles + t) * 2;
        s1 = s1 + p[0];
        s2 = s2 + p[1];
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 = s1 + __shfl_down(s1, off, 64);
        s2 = s2 + __shfl_down(s2, off, 64);
    }
    __syncthreads();
    if (lane == 0) { s_red[2 * wave] = s1; s_red[2 * wave + 1] = s2; }
    __syncthreads();
    if (tid == 0) {
        double t1 = s_red[0], t2 = s_red[1];
        for (int w = 1; w < TB / 64; w++) { t1 = t1 + s_red[2 * w]; t2 = t2 + s_red[2 * w + 1]; }
        const int C = a.C;
        *s_out = mh_outcome(a, j, itp, t1 + t2, a.logPr_prop[P * C + j], a.status_prop[P * C + j], a.logL_cur[P * C + j], a.logPr_cur[P * C + j],
                            a.logPost_cur[P * C + j], a.Tcoefs[j], a.init_logL[j]);
    }
    __syncthreads();
}

// Sum of a chain's per-tile partials by ONE wave in k_finalize's order (kernels.hip): 256 strided per-thread sums (four per lane here),
// a shuffle tree per 64, the four in order.  Every lane returns the total.
__device__ __forceinline__ double wave_sum_in_order(const double (&s1)[TB / 64], const double (&s2)[TB / 64]) {
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int q = 0; q < TB / 64; q++) {
        double a1 = s1[q], a2 = s2[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a1 = a1 + __shfl_down(a1, off, 64);
            a2 = a2 + __shfl_down(a2, off, 64);
        }
        if (q == 0) { t1 = a1; t2 = a2; }
        else { t1 = t1 + a1; t2 = t2 + a2; }
    }
    return __shfl(t1 + t2, 0, 64);
}
__device__ __forceinline__ double wave_partial_sum(const double *base, int ntiles) {
    const int lane = threadIdx.x & 63;
    double s1[TB / 64], s2[TB / 64];
#pragma unroll
    for (int q = 0; q < TB / 64; q++) { s1[q] = 0; s2[q] = 0; }
    for (int t0 = 0; t0 < ntiles; t0 += TB) {  // virtual thread q*64+lane of k_finalize adds tile t0 + q*64 + lane in this round
        double v1[TB / 64], v2[TB / 64];
#pragma unroll
        for (int q = 0; q < TB / 64; q++) {  // the round's loads first: one memory round trip instead of four
            const int t = t0 + q * 64 + lane;
            v1[q] = t < ntiles ? base[2 * t] : 0.0;
            v2[q] = t < ntiles ? base[2 * t + 1] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TB / 64; q++)
            if (t0 + q * 64 + lane < ntiles) { s1[q] = s1[q] + v1[q]; s2[q] = s2[q] + v2[q]; }
    }
    return wave_sum_in_order(s1, s2);
}

// Robbins-Monro adaptation of chain m's proposal law (MALA.cpp:296-319) and Cholesky of (Sigma+eps2 I) sigma
// (MALA.cpp:348-350); `vars` = the chain's position after the MH test, `Pm` = its move probability.
// WP: pointer type of the work matrix A and the vector d in their address space (LDS when the matrix fits there: ds_read/ds_write
// instead of flat accesses, whose latency is several times higher; device memory otherwise); PANELS: the blocked factorisation.
template <class WP, bool PANELS>
__device__ void adapt_chain_as(const DevSamplerArgs &a, int m, long itp, const double *vars, double Pm, WP A, WP d, double *s_red, double *s_scal) {
    const int tid = threadIdx.x, Nv = a.Nv;
    const double g = a.c0 / (1. + (double)itp);
    double *mu = a.mu + (size_t)m * Nv;
    double *cov = a.cov + (size_t)m * Nv * Nv;
    double n2 = 0;
    for (int k = tid; k < Nv; k += TB) {
        const double v = mu[k] + g * (vars[k] - mu[k]);
        d[k] = v;
        n2 += v * v;
    }
    n2 = wg_sum(n2, s_red);
    {
        const double nrm = sqrt(n2);
        const double sc = (nrm <= a.A1) ? 1.0 : a.A1 / nrm;  // p3_fct
        for (int k = tid; k < Nv; k += TB) {
            const double v = (sc == 1.0) ? d[k] : d[k] * sc;
            mu[k] = v;
            d[k] = vars[k] - v;  // deviation from the UPDATED mu (MALA.cpp:311)
        }
    }
    __syncthreads();
#ifdef TAMCMC_PROBE
    if (a.probe == 1) return;
#endif
    // covariance update (MALA.cpp:313-316) and the matrix to factor, A = (Sigma + eps2 I) sigma, in one sweep over Sigma (device memory,
    // read and written once); lanes as a 16 x 16 grid over (row, column): no index arithmetic per element, 128-byte runs per row
    const int gi = tid >> 4, gk = tid & 15;
    n2 = 0;
    constexpr int CB = 8;  // columns of a lane per batch: two rows x CB device-memory reads are in flight before the first use
#pragma clang loop unroll(disable)
    for (int i = gi; i < Nv; i += 32) {
        const int i2 = i + 16;
        const bool two = i2 < Nv;
        const double di = d[i], di2 = two ? d[i2] : 0.0;
#pragma clang loop unroll(disable)
        for (int jb = gk; jb < Nv; jb += 16 * CB) {
            double c0[CB], c1[CB];
            static_for<CB>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int j = jb + 16 * q;
                c0[q] = (j < Nv) ? cov[(size_t)i * Nv + j] : 0.0;
                c1[q] = (two && j < Nv) ? cov[(size_t)i2 * Nv + j] : 0.0;
            });
            static_for<CB>([&](auto qc) {  // row i (the sum of squares keeps the element order of a plain row-by-row sweep per lane)
                constexpr int q = decltype(qc)::value;
                const int j = jb + 16 * q;
                if (j < Nv) {
                    const size_t e = (size_t)i * Nv + j;
                    const double v = c0[q] + g * (di * d[j] - c0[q]);
                    cov[e] = v;
                    A[e] = v;
                    n2 += v * v;
                }
            });
            static_for<CB>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int j = jb + 16 * q;
                if (two && j < Nv) {
                    const size_t e = (size_t)i2 * Nv + j;
                    const double v = c1[q] + g * (di2 * d[j] - c1[q]);
                    cov[e] = v;
                    A[e] = v;
                    n2 += v * v;
                }
            });
        }
    }
#ifdef TAMCMC_PROBE
    if (a.probe == 2) return;
#endif
    n2 = wg_sum(n2, s_red);
    if (tid == 0) {
        const double nrm = sqrt(n2);
        s_scal[0] = (nrm <= a.A1) ? 1.0 : a.A1 / nrm;  // p2_fct
        double v1 = a.sigma[m] + g * (Pm - a.target_acceptance);
        if (v1 < a.epsilon1) v1 = a.epsilon1;  // p1_fct
        if (v1 > a.A1) v1 = a.A1;
        a.sigma[m] = v1;
        s_scal[1] = v1;
    }
    __syncthreads();
    const double sc = s_scal[0], sig = s_scal[1];
    for (int i = gi; i < Nv; i += 16)
        for (int j = gk; j < Nv; j += 16) {
            const size_t e = (size_t)i * Nv + j;
            double v = A[e];
            if (sc != 1.0) { v = v * sc; cov[e] = v; }  // (a covariance of norm > A1 = 1e14: never with sane inputs)
            A[e] = (v + (i == j ? a.epsi2 : 0.0)) * sig;
        }
    __syncthreads();
#ifdef TAMCMC_PROBE
    if (a.probe == 3) return;
#endif
    // Cholesky in place (lower triangle of A).  A matrix that is not positive definite (possible only while gamma = c0/(1+i) > 1,
    // i.e. adaptation before iteration c0) keeps the PREVIOUS factor -- the host engine does the same (host_mala.cpp::factor); the
    // reference hands Eigen's partial result on.  Every element sees the operations of the right-looking algorithm in its order
    // (A_ik -= L_ij L_kj for j ascending, then scaled by 1/d_kk), the host engine's factor to 1-2 ulp (round 3: reciprocal square roots
    // in the panels' diagonal blocks; the sqrt / divide sequence of the host engine was the factorisation's serial chain):
    //   * panels of NB columns: the NB x NB diagonal block is factored by NB lanes of one wave (rows in registers, pivots by
    //     shuffles, no workgroup barrier inside); the panel's columns below it are one forward substitution per row, a row per lane;
    //     then all lanes apply the NB columns to the trailing block in one sweep (a 16 x 16 grid over rows x columns, L_i,panel in
    //     registers along a row).  3 barriers per panel instead of 3 per column; the serial chain is sqrt -> divide per column.
    constexpr int NB = 8;
    bool pd = true;  // positive definite so far
    int j0 = 0;      // columns done by panels
    const int ti = tid >> 4, tk = tid & 15;
    if constexpr (PANELS) {
        if (tid == 0) s_scal[0] = 0.0;  // "not positive definite" flag
        __syncthreads();
        // (1) a panel's NB x NB diagonal block, by the first NB lanes of wave 0 (lane r = row p0+r in registers; pivots by readlane);
        //     called by the whole of wave 0
        auto diag_block = [&](const int p0) __attribute__((always_inline)) {
            double r[NB];
            const int row = p0 + tid;
            static_for<NB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                r[c] = (tid < NB) ? A[(size_t)row * Nv + p0 + c] : 0.0;
            });
            bool bad = false;
            static_for<NB>([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                if (!bad) {  // wave-uniform
                    const double ajj = lane_value<jj>(r[jj]);
                    if (!(ajj > 0.0)) bad = true;
                    else {
                        // 1/sqrt(a_jj): v_rsq_f64 seed + two Newton steps (the serial chain of the factorisation is this step, once per
                        // column: an IEEE sqrt followed by an IEEE divide is ~5x as long); the column is scaled by it, the diagonal is
                        // a_jj / sqrt(a_jj) with one correction step.  1-2 ulp from the sqrt / divide factor of the host engine
                        double y = __builtin_amdgcn_rsq(ajj);
                        y = fma(y, fma(-ajj * y, 0.5 * y, 0.5), y);
                        y = fma(y, fma(-ajj * y, 0.5 * y, 0.5), y);
                        double djj = ajj * y;
                        djj = fma(fma(-djj, djj, ajj), 0.5 * y, djj);
                        if (tid > jj) r[jj] = r[jj] * y;
                        else if (tid == jj) { r[jj] = djj; d[p0 + jj] = y; }  // (d[] is free since the covariance update: reciprocal pivots)
                        static_for<NB - 1 - jj>([&](auto kc) {
                            constexpr int kk = jj + 1 + decltype(kc)::value;
                            const double lk = lane_value<kk>(r[jj]);  // L_(p0+kk),jj
                            if (tid >= kk) r[kk] = r[kk] - r[jj] * lk;
                        });
                    }
                }
            });
            if (bad) { if (tid == 0) s_scal[0] = 1.0; }
            else if (tid < NB)
                static_for<NB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if (c <= tid) A[(size_t)row * Nv + p0 + c] = r[c];
                });
        };
        // (2) the panel's columns below the block, one row per lane: L_i,jj = (A_i,jj - sum_{j' < jj} L_i,j' L_jj,j') / d_jj
        auto below_block = [&](const int p0) __attribute__((always_inline)) {
#pragma clang loop unroll(disable)
            for (int i = p0 + NB + tid; i < Nv; i += TB) {
                double li[NB], Ld[NB][NB];  // the row's panel entries and the diagonal block: every LDS read is requested before the first use
                static_for<NB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    li[c] = A[(size_t)i * Nv + p0 + c];
                    static_for<c>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        Ld[c][q] = A[(size_t)(p0 + c) * Nv + p0 + q];
                    });
                    Ld[c][c] = d[p0 + c];  // reciprocal pivot (diag_block)
                });
                static_for<NB>([&](auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    static_for<jj>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        li[jj] = li[jj] - li[q] * Ld[jj][q];
                    });
                    li[jj] = li[jj] * Ld[jj][jj];
                });
                static_for<NB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    A[(size_t)i * Nv + p0 + c] = li[c];
                });
            }
        };
        // (3) the panel's NB columns applied to columns kb..ke-1 of the trailing block (rows i >= kb, columns <= i); the calling lanes
        //     form an RS x CS grid (ri, rk)
        auto trailing = [&](const int p0, const int kb, const int ke, const int ri, const int rk, auto rs_c, auto cs_c) __attribute__((always_inline)) {
            constexpr int RS = decltype(rs_c)::value, CS = decltype(cs_c)::value;
#pragma clang loop unroll(disable)
            for (int i = kb + ri; i < Nv; i += RS) {
                double li[NB];
                static_for<NB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    li[c] = A[(size_t)i * Nv + p0 + c];
                });
                const int kend = i < ke - 1 ? i : ke - 1;  // last column of the row
                int k = kb + rk;
#pragma clang loop unroll(disable)
                for (; k + CS <= kend; k += 2 * CS) {  // two columns per trip: their LDS reads are in flight together (one wave per SIMD here)
                    double v0 = A[(size_t)i * Nv + k], v1 = A[(size_t)i * Nv + k + CS], l0[NB], l1[NB];
                    static_for<NB>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        l0[c] = A[(size_t)k * Nv + p0 + c];
                        l1[c] = A[(size_t)(k + CS) * Nv + p0 + c];
                    });
                    static_for<NB>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        v0 = v0 - li[c] * l0[c];
                        v1 = v1 - li[c] * l1[c];
                    });
                    A[(size_t)i * Nv + k] = v0;
                    A[(size_t)i * Nv + k + CS] = v1;
                }
                if (k <= kend) {
                    double v = A[(size_t)i * Nv + k];
                    static_for<NB>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        v = v - li[c] * A[(size_t)k * Nv + p0 + c];
                    });
                    A[(size_t)i * Nv + k] = v;
                }
            }
        };
        // Schedule: the next panel's diagonal block (the serial sqrt -> divide chain) is factored by wave 0 WHILE waves 1-3 apply the
        // current panel to the rest of the trailing block; only the next panel's own NB columns are updated ahead of it by all lanes.
#pragma clang loop unroll(disable)
        for (int p = -NB;;) {  // p: the panel being applied (none yet on the first trip, which only factors block 0)
            const int c0 = p + NB;
#ifdef TAMCMC_PROBE
            long pt0 = (long)wall_clock64(), pt1 = pt0;
#endif
            if (p >= 0) {
                below_block(p);
                __syncthreads();
#ifdef TAMCMC_PROBE
                pt1 = (long)wall_clock64();
#endif
                trailing(p, c0, c0 + NB, tid >> 3, tid & 7, std::integral_constant<int, TB / 8>{}, std::integral_constant<int, 8>{});
                __syncthreads();
            }
            j0 = c0;
            if (c0 + NB > Nv) break;  // fewer than NB columns left: the slice above was the whole trailing block
#ifdef TAMCMC_PROBE
            long pt2 = (long)wall_clock64();
#endif
            if (tid < 64) diag_block(c0);
            else if (p >= 0)
                trailing(p, c0 + NB, Nv, (tid - 64) >> 4, tid & 15, std::integral_constant<int, (TB - 64) / 16>{}, std::integral_constant<int, 16>{});
            __syncthreads();
#ifdef TAMCMC_PROBE
            if (m == 0 && tid == 0) {
                const long pt3 = (long)wall_clock64();
                a.counters[4] += pt1 - pt0; a.counters[5] += pt2 - pt1; a.counters[6] += pt3 - pt2; a.counters[7] += 1;
            }
#endif
            if (s_scal[0] != 0.0) { pd = false; break; }  // every lane leaves together, L is not touched
            p = c0;
        }
    }
    // the columns the panels leave (fewer than NB; all of them for wide proposals, whose work matrix is in device memory): one per step
#pragma clang loop unroll(disable)
    for (int j = j0; j < Nv && pd; j++) {
        const double ajj = A[(size_t)j * Nv + j];  // workgroup-uniform (its last update was before the previous step's closing barrier)
        if (!(ajj > 0.0)) { pd = false; break; }   // every lane leaves together, L is not touched
        const double djj = sqrt(ajj);
        if (tid == 0) d[j] = djj;                  // the new diagonal is parked in d[] (free since the covariance update)
        for (int i = j + 1 + tid; i < Nv; i += TB) A[(size_t)i * Nv + j] = A[(size_t)i * Nv + j] / djj;
        __syncthreads();
        for (int i = j + 1 + ti; i < Nv; i += 16) {
            const double lij = A[(size_t)i * Nv + j];
            for (int k = j + 1 + tk; k <= i; k += 16) A[(size_t)i * Nv + k] = A[(size_t)i * Nv + k] - lij * A[(size_t)k * Nv + j];
        }
        __syncthreads();
    }
    if (pd)
        for (int j = j0 + tid; j < Nv; j += TB) A[(size_t)j * Nv + j] = d[j];
    __syncthreads();
#ifdef TAMCMC_PROBE
    if (a.probe == 4) return;
#endif
    double *LT = a.LT + (size_t)m * Nv * Nv;  // the factor transposed (row k of LT = column k of L), written in 128-byte runs
    if (pd)
        for (int k = gi; k < Nv; k += 16)
            for (int i = gk; i < Nv; i += 16) LT[(size_t)k * Nv + i] = (k <= i) ? A[(size_t)i * Nv + k] : 0.0;
    __syncthreads();
}
__device__ __forceinline__ void adapt_chain(const DevSamplerArgs &a, int m, long itp, const double *vars, double Pm, double *A, double *d, double *s_red,
                            double *s_scal) {
    typedef double __attribute__((address_space(3))) *lds_dp_t;
    typedef double __attribute__((address_space(1))) *dev_dp_t;
    if (a.chol_in_lds) adapt_chain_as<lds_dp_t, true>(a, m, itp, vars, Pm, (lds_dp_t)A, (lds_dp_t)d, s_red, s_scal);
    else adapt_chain_as<dev_dp_t, false>(a, m, itp, vars, Pm, (dev_dp_t)A, (dev_dp_t)d, s_red, s_scal);
}

// z ~ N(0, I) of (chain, iteration) into LDS (ends without a barrier) and row i of L z (MALA.cpp:348-355)
__device__ __forceinline__ void normals_into(const DevSamplerArgs &a, int chain, long it, double *s_z) {
    for (int k2 = threadIdx.x; 2 * k2 < a.Nv; k2 += (int)blockDim.x) {
        double z0, z1;
        rng_normal2(a.seed, RNG_PROPOSAL, (uint32_t)chain, (uint64_t)it, (uint32_t)k2, z0, z1);
        s_z[2 * k2] = z0;
        s_z[2 * k2 + 1] = z1;
    }
}
__device__ __forceinline__ double Lz_row(const DevSamplerArgs &a, int chain, int i, const double *s_z) {
    const double *LT = a.LT + (size_t)chain * a.Nv * a.Nv;
    double s = 0;
    for (int k = 0; k <= i; k++) s = s + LT[(size_t)k * a.Nv + i] * s_z[k];
    return s;
}

// The same rows of L z with the loads of a batch issued before the first use (a row's sum stays in ascending k, the order of Lz_row):
// lane i owns rows i and i+64.  A wave on its own has no other wave's loads to hide behind.
__device__ __forceinline__ void Lz_rows_wave(const DevSamplerArgs &a, int chain, const double *s_z, double *out) {
    constexpr int NB = 8;
    const int Nv = a.Nv, lane = threadIdx.x;
    const double *LT = a.LT + (size_t)chain * Nv * Nv;
#pragma clang loop unroll(disable)
    for (int i = lane; i < Nv; i += 64) {
        double s = 0;
        int k0 = 0;
#pragma clang loop unroll(disable)
        for (; k0 + NB <= i + 1; k0 += NB) {  // full batches: NB independent loads, then the NB terms in order
            double l[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) l[u] = LT[(size_t)(k0 + u) * Nv + i];
#pragma unroll
            for (int u = 0; u < NB; u++) s = s + l[u] * s_z[k0 + u];
        }
#pragma clang loop unroll(disable)
        for (; k0 <= i; k0++) s = s + LT[(size_t)k0 * Nv + i] * s_z[k0];
        out[i] = s;
    }
}

__host__ __device__ inline bool is_rgb_model(int id) { return id == TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4_ID || id == TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4_ID; }

// Proposal of iteration `it` for `chain` from the state in LDS (s_vars/s_params): x' = x + L z (MALA.cpp:348-355), L =
// chol((Sigma+eps2) sigma) stored transposed, same Philox streams as the host engine; log-prior; params' -> multiplet table
// written into slot `slot` of the likelihood kernel's input block.  Ends without a barrier.  (B): 256 threads.
__device__ __forceinline__ void propose_common(const DevSamplerArgs &a, const UnpackLds &U, int chain, long it, int slot, double *pv, double *pp,
                               double *logPr_out, int *status_out, double *s_vars, double *s_params, double *s_z, const double *lz = nullptr,
                               const rgb::Slice *rs = nullptr, int rb = 0) {
    const int Np = a.desc.Np, Nv = a.Nv, tid = threadIdx.x;
    const bool rgb = is_rgb_model(a.desc.model_id);
    if (!lz) normals_into(a, chain, it, s_z);
    if (!rgb) unpack_begin(a.desc, U);
    else __syncthreads();
    for (int i = tid; i < Nv; i += TB) {  // lane i owns row i: reads s_vars[i] only, every s_z[k]
        const double s = lz ? lz[i] : Lz_row(a, chain, i, s_z);
        const double v = s_vars[i] + 0.0 + s;
        s_vars[i] = v;
        pv[i] = v;
    }
    __syncthreads();
    for (int k = tid; k < Nv; k += TB) s_params[a.index_to_relax[k]] = s_vars[k];  // update_params_with_vars
    __syncthreads();
    for (int i = tid; i < Np; i += TB) pp[i] = s_params[i];
    if (rgb) {
        // red-giant models (ids 25 / 27): the table needs the mixed-mode solver -- the kernels enqueued right behind this launch
        // (rgb_device_stage) build it.  Here: the log-prior (priors_calc.cpp:319-512; generic terms one per lane, summed by lane 0 in the
        // reference's order) by wave 0 while wave 1 runs the scalar unpack of the proposal (rgb_unpack.h) into the group's workspace slice.
        __shared__ rgb::Prep sP;
        __shared__ rgb::RowIn sR;
        __shared__ double s_w[40], s_noise[3 * TAMCMC_MAX_HARVEY + 4], s_lp;
        __shared__ int32_t s_hn[2];
        __shared__ int s_stp;
        mt::xreal *terms = (mt::xreal *)U.poly;  // (the polynomial tables' LDS is not used by these models; xreal = double on the device)
        const bool spread = a.desc.prior_class == 4 && (size_t)Np * sizeof(mt::xreal) <= sizeof(mt::PolyTab);
        if (tid == 0) *U.status = TAMCMC_OK;
        __syncthreads();
        if (spread)
            for (int i = tid; i < Np; i += TB) {
                int st = TAMCMC_OK;
                terms[i] = pr::generic_prior_term(s_params, Np, a.desc.priors, a.desc.priors_switch, i, &st);
                if (st != TAMCMC_OK) *U.status = st;
            }
        __syncthreads();
        if (tid == 0) {
            int st = *U.status;
            s_lp = (double)pr::prior_serial(a.desc.prior_class, s_params, a.desc.plength, Np, a.desc.priors, a.desc.priors_switch, a.desc.extra, &st,
                                            spread ? terms : nullptr);
            s_stp = st;
        } else if ((tid >> 6) == 1) {
            rgb::WaveLanes x;
            x.w = s_w;
            double fmin;
            rgb::unpack_vector(x, s_params, a.desc.plength, rs->step, a.desc.model_id == TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4_ID, rs->dense, sP, sR, s_noise,
                               &s_hn[0], &s_hn[1], &fmin);
        }
        __syncthreads();
        const double lp = s_lp;
        const int stp = s_stp;
        if (tid == 0 && (stp != TAMCMC_OK || lp == -INFINITY || isnan(lp))) {  // model_def.cpp:472,476-480 skips the model: nothing to solve
            sP.Lp = 0; sP.Lg = 0; sP.status = stp != TAMCMC_OK ? stp : TAMCMC_ERR_BAD_ARG;
            sR.status = sP.status; sR.Nfl0 = sR.Nfl2 = sR.Nfl3 = 0; sR.bias_n = 0;
            s_noise[0] = 1.0;
            s_hn[0] = 0; s_hn[1] = 1;
        }
        __syncthreads();
        static_assert(sizeof(rgb::Prep) % 8 == 0 && sizeof(rgb::RowIn) % 8 == 0, "copied as doubles");
        const double *src = (const double *)&sP;
        double *dst = (double *)&rs->preps[rb];
        for (int i = tid; i < (int)(sizeof(rgb::Prep) / 8); i += TB) dst[i] = src[i];
        src = (const double *)&sR;
        dst = (double *)&rs->rows[rb];
        for (int i = tid; i < (int)(sizeof(rgb::RowIn) / 8); i += TB) dst[i] = src[i];
        for (int i = tid; i < s_hn[1] && i < a.desc.stride; i += TB) a.noise[(size_t)slot * a.desc.stride + i] = s_noise[i];
        if (tid == 0) {
            rs->norm_bits[rb] = 0ull;
            rs->nsol[rb] = 0;
            a.nh[slot] = s_hn[0];
            a.nn[slot] = s_hn[1];
            *logPr_out = lp;
            *status_out = stp;
        }
        return;
    }

    // ---- log-prior, then params' -> multiplet table written into the likelihood kernel's input block ----
    TablePtrs T;
    T.mults = a.mults; T.pairs = a.pairs; T.nh = a.nh; T.nn = a.nn; T.noise = a.noise;
    T.bg = a.bg; T.ntiles = a.ntiles; T.tile_bins = a.tile_bins;
    // four roles beside each other (dev_unpack.h): prior terms + background tiles | table rows | shared scalars + m-visibilities
    const double logPr = wg_log_prior(a.desc, s_params, U, true, true, &T, slot);
    const bool live = (logPr != -INFINITY) && !isnan(logPr);  // model_def.cpp:472,476-480
    wg_unpack(a.desc, s_params, U, slot, T, live, true, true);
    if (tid == 0) {
        *logPr_out = logPr;
        *status_out = *U.status;
    }
}

template <class AT>
__device__ __forceinline__ bool is_swap_iter(const AT &a, long i) {
    return a.dN_mixing > 0 && (i % a.dN_mixing == 0) && i != 0 && a.C > 1;
}
template <class AT>
__device__ __forceinline__ int swap_first(const AT &a, long i, double *u_out) {  // MALA.cpp:397-405
    double u, u2;
    rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)i, 0, u, u2);
    int A = (int)(u2 * (double)(a.C - 1));
    if (A > a.C - 2) A = a.C - 2;
    if (u_out) *u_out = u;
    return A;
}

// Parallel tempering (MALA.cpp:397-461) on the pair's outcomes AFTER their MH tests: does the pair swap, and what does each side
// then hold as tempered logL / prior / posterior.  oA, oB are updated in place; returns 1 when swapped.
template <class AT>
__device__ __forceinline__ int resolve_swap(const AT &a, int A, double u, AcceptOut &oA, AcceptOut &oB) {
    const int B = A + 1;
    const double LA = oA.logL, LB = oB.logL;
    const double LA_TB = LA * a.Tcoefs[A] / a.Tcoefs[B];
    const double LB_TA = LB * a.Tcoefs[B] / a.Tcoefs[A];
    const double e = exp(LA_TB + LB_TA - LA - LB);
    const double rT = fmin(1.0, e);
    if (!(u <= rT)) return 0;
    const double prA = oA.logPr, prB = oB.logPr;
    oA.logL = LB_TA; oA.logPr = prB; oA.logPost = LB_TA + prB;      // A <- B, re-tempered (MALA.cpp:431-435)
    // swap_rule 1 (MALA.cpp:433,444 as executed): B's stored posterior carries B's own old prior
    oB.logL = LA_TB; oB.logPr = prA; oB.logPost = LA_TB + (a.swap_rule == 1 ? prB : prA);
    return 1;
}

// ===============================================================================================================
// (B) LOCKSTEP.  ONE kernel per MCMC iteration besides the likelihood kernel.  Workgroup m:
//   (0) settles the pending iteration it-1 for chain m: MH test (own chain; the swap partner's too when chain m is in the
//       swap pair), adjacent-pair parallel-tempering swap, writes the chain's new current state into the OTHER parity
//       buffer (no workgroup ever writes what another one reads), records the sample, adapts the proposal law;
//   (1) proposes iteration `it` from that state: x' = x + L z, log-prior, params' -> multiplet table.
template <bool PROPOSE>
__global__ void __launch_bounds__(TB) k_iterate(const DevSamplerArgs a, const long it, const int P, const int pending,
                                               const long rec, const int learn_pending, double *scratch, const int c_off,
                                               const int nmain, const int pre_flags, const rgb::Slice rs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int Np = a.desc.Np, Nv = a.Nv, C = a.C;
    if ((int)blockIdx.x >= nmain) {
        // spare workgroup (launched while no adaptation touches L): L z of iteration it+1 for chain c_off + blockIdx.x - nmain
        const int ch = c_off + (int)blockIdx.x - nmain;
        double *z = (double *)s_raw;
        normals_into(a, ch, it + 1, z);
        __syncthreads();
        double *dst = a.lz + ((size_t)((it + 1) & 1) * C + ch) * Nv;
        for (int i = threadIdx.x; i < Nv; i += TB) dst[i] = Lz_row(a, ch, i, z);
        return;
    }
    double *s_params = (double *)s_raw;          // [Np]   current, then proposed parameter vector
    double *s_vars = s_params + Np;              // [Nv]   current, then proposed variables
    double *s_z = s_vars + Nv;                   // [Nv+1] normals / post-test position for the adaptation
    const UnpackLds U = carve_unpack_lds((unsigned char *)(s_z + Nv + 1));
    double *s_red = U.red;
    double *s_A = (double *)(((uintptr_t)(s_z + Nv + 1) + unpack_lds_bytes() + 15) & ~(uintptr_t)15);  // [Nv*Nv + Nv] when learning in LDS
    __shared__ AcceptOut s_own, s_partner;
    __shared__ double s_scal[2];

    const int m = blockIdx.x + c_off, tid = threadIdx.x;  // c_off: first chain of this launch's chain group
    const int Q = P ^ 1;
    const double *curv = a.vars_cur + (size_t)P * C * Nv, *curp = a.params_cur + (size_t)P * C * Np;
    const double *prpv = a.vars_prop + (size_t)P * C * Nv, *prpp = a.params_prop + (size_t)P * C * Np;
    double *newv = a.vars_cur + (size_t)Q * C * Nv, *newp = a.params_cur + (size_t)Q * C * Np;

    // ------------------------------------------------------------------ (0) settle the pending iteration
    if (pending) {
        const long itp = it - 1;
        accept_result(a, m, itp, P, s_red, &s_own);
        int src = m;
        AcceptOut mine = s_own;
        // parallel tempering (MALA.cpp:397-461): adjacent pair, tempered log-likelihoods after the MH tests
        if (is_swap_iter(a, itp)) {
            double u;
            const int A = swap_first(a, itp, &u);
            const int B = A + 1;
            if (m == A || m == B) {  // workgroup-uniform branch
                const int partner = (m == A) ? B : A;
                accept_result(a, partner, itp, P, s_red, &s_partner);
                AcceptOut oA = (m == A) ? s_own : s_partner, oB = (m == A) ? s_partner : s_own;
                const int swapped = resolve_swap(a, A, u, oA, oB);
                if (swapped) { src = partner; mine = (m == A) ? oA : oB; }
                if (m == A && tid == 0) {  // (chain groups: launches of different iterations may overlap)
                    atomicAdd((unsigned long long *)&a.counters[2], 1ull);
                    if (swapped) atomicAdd((unsigned long long *)&a.counters[3], 1ull);
                }
            }
        }
        const int src_acc = (src == m) ? s_own.acc : s_partner.acc;
        const double *sv = (src_acc ? prpv : curv) + (size_t)src * Nv;
        const double *sp = (src_acc ? prpp : curp) + (size_t)src * Np;
        for (int i = tid; i < Nv; i += TB) { const double v = sv[i]; s_vars[i] = v; newv[(size_t)m * Nv + i] = v; }
        for (int i = tid; i < Np; i += TB) { const double v = sp[i]; s_params[i] = v; newp[(size_t)m * Np + i] = v; }
        if (learn_pending) {  // the adaptation sees the chain's OWN position after the MH test, before the swap
            const double *ov = (s_own.acc ? prpv : curv) + (size_t)m * Nv;
            for (int i = tid; i < Nv; i += TB) s_z[i] = ov[i];
        }
        if (tid == 0) {
            a.logL_cur[Q * C + m] = mine.logL;
            a.logPr_cur[Q * C + m] = mine.logPr;
            a.logPost_cur[Q * C + m] = mine.logPost;
            // a swap exchanges the pair's moved / Pmove entries too (MALA.cpp:425-446): what is recorded is the partner's
            a.moved[m] = (src == m) ? s_own.acc : s_partner.acc;
            a.Pmove[m] = (src == m) ? s_own.r : s_partner.r;
            if (m == 0 && a.moved[0]) a.counters[1] += 1;
            a.counters[8 + m] += a.moved[m];  // per-chain count of recorded moves (the acceptance diagnostic, outputs.cpp:1824-1858)
            if (m == 0) a.counters[0] = it;
            if (a.stats && rec >= 0) {  // update_buffer_stat_criteria (MALA.cpp:708)
                double *r = a.stats + ((size_t)rec * C + m) * 3;
                r[0] = mine.logL; r[1] = mine.logPr; r[2] = mine.logPost;
            }
        }
        __syncthreads();
        if (a.samples && rec >= 0)  // update_buffer_params (MALA.cpp:710)
            for (int i = tid; i < Nv; i += TB) a.samples[((size_t)rec * C + m) * Nv + i] = s_vars[i];
        if (learn_pending) {
            double *Aw = a.chol_in_lds ? s_A : scratch + (size_t)m * ((size_t)Nv * Nv + Nv);
            adapt_chain(a, m, itp, s_z, s_own.r, Aw, Aw + (size_t)Nv * Nv, s_red, s_scal);
        }
    } else {
        for (int i = tid; i < Nv; i += TB) { const double v = curv[(size_t)m * Nv + i]; s_vars[i] = v; newv[(size_t)m * Nv + i] = v; }
        for (int i = tid; i < Np; i += TB) { const double v = curp[(size_t)m * Np + i]; s_params[i] = v; newp[(size_t)m * Np + i] = v; }
        if (tid == 0) {
            a.logL_cur[Q * C + m] = a.logL_cur[P * C + m];
            a.logPr_cur[Q * C + m] = a.logPr_cur[P * C + m];
            a.logPost_cur[Q * C + m] = a.logPost_cur[P * C + m];
        }
    }
    if (!PROPOSE) return;
    __syncthreads();

    // ------------------------------------------------------------------ (1) propose iteration `it`
    propose_common(a, U, m, it, m, a.vars_prop + (size_t)Q * C * Nv + (size_t)m * Nv, a.params_prop + (size_t)Q * C * Np + (size_t)m * Np,
                   a.logPr_prop + Q * C + m, a.status_prop + Q * C + m, s_vars, s_params, s_z,
                   (pre_flags & 1) ? a.lz + ((size_t)(it & 1) * C + m) * Nv : nullptr, &rs, (int)blockIdx.x);
}

// ===============================================================================================================
// (A) FUSED STEP.
struct FusedArgs {
    int NS;                // candidate slots per iteration: 2C + 8 (two blocks of four extra slots for a swap pair's cross candidates)
    int xsplit;            // first chain of the second chain group (C: none).  A chain's cross candidates after a swap live in extra block
                           // (chain >= xsplit): the two groups' launches run on different streams, possibly several iterations apart,
                           // and must never write what the other one still reads
    // candidates of iteration i live in candidate set i mod 3: [3][NS]...  (launch i reads the sets of iterations i-1 and i and writes
    // the set of iteration i+1)
    double *cand_vars, *cand_params;
    double *cand_logPr;                // [3][NS][2] the two halves of the log-prior's additive terms (wave_log_prior_part), added in order
    int *cand_rej;                     // [3][NS]    a hard constraint fails: the log-prior is -inf
    int *cand_stP, *cand_stR;          // [3][NS][2], [3][NS] status of the two prior roles / the rows role
    tamcmc_multiplet *mults;           // [3][NS][per]
    int *pairs, *nh, *nn;              // [3][2 NS], [3][NS], [3][NS]
    double *noise;                     // [3][NS][stride]
    double *bg;                        // [3][NS][ntiles][8] or nullptr
    // per chain, by the parity of the iteration: written by the launch of that iteration (commit_chain), read by the next one
    int *slot;                         // [2][C]   table slot of chain m's proposal at that iteration
    double *prop_logPr;                // [2][C]   that proposal's log-prior ...
    int *prop_st;                      // [2][C]   ... and status (prior role's, else rows role's)
    double *quick;                     // [2][C][QN] that iteration's MH and swap tests as thresholds on the sums of the partials (quick_decide)
    double *part;                      // [2][C][ntiles][2] the tiles' partial sums of that iteration
    double *lz;                        // [2][C][Nv] L z of chain m for the iteration of that parity, computed one launch ahead
};

constexpr int QN = 8;  // doubles per quick record (quick_decide): S*, kind, -pl/T, logL held, [pair's first chain: log u_swap, TA/TB - 1, TB/TA - 1], slot
constexpr int ST_L = 1, ST_BR = 2, ST_ENTRY = 4, ST_LZ = 8, ST_FIRST = 16, ST_COMMIT = 32;

// The decide / commit functions are real calls (register budget of the tile path) and get the argument blocks as pointers to their
// device-memory image.  That image is written by the host only, and the pointer is the same in every lane: read through a wave-uniform
// pointer into constant memory, a field costs a scalar load (SGPR, scalar cache) instead of a flat vector load per lane, and the
// pointers found there are known to be global (global_load / global_store instead of flat_).
typedef DevSamplerArgs __attribute__((address_space(4))) ConstArgs;
typedef FusedArgs __attribute__((address_space(4))) ConstFused;
__device__ __forceinline__ const void __attribute__((address_space(4))) *uniform_ptr(const void *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (const void __attribute__((address_space(4))) *)(((unsigned long long)hi << 32) | lo);
}

// What chain m enters iteration `it` with -- the outcome of iteration it-1's MH test and swap (MALA.cpp:397-461, 490-551).
struct Decided {
    int slot;       // candidate slot (set it mod 3) of the chain's proposal at iteration `it`
    int src;        // chain whose post-test position the chain continues from: itself, or its swap partner
    int src_acc;    // 1: that position is src's proposal of iteration it-1 (candidate set (it-1) mod 3, slot src_ps); 0: what src held
    int src_par;    // parity of the state arrays that hold src's position (src_acc == 0) -- the previous iteration's, or, in the first
                    // launch of a stretch, this iteration's (the chains are settled)
    int src_ps;
    int swap_first; // first chain of iteration it-1's swap pair when chain m is in it, else -1
    int swapped;
    double r;       // move probability of src's test (a swap exchanges the pair's moved / Pmove entries too, MALA.cpp:425-446)
    AcceptOut o;    // the scalars the chain holds (re-tempered after a swap)
};

// Iteration it-1 of chain m decided by ONE wave from what launch it-1 left in memory: the tiles' partial sums (summed in k_finalize's
// order), the proposal's prior and status, the scalars the chain held.  Every workgroup of launch `it` that needs the outcome -- each
// likelihood tile of the chain (its table slot), the chain's commit workgroup, the candidate roles built on the chain's vectors --
// recomputes it from the same inputs: same result everywhere, no hand-off inside a launch (no tickets, no device-scope accesses), and no
// settle step at the end of the launch's critical path.  For the two chains of iteration it-1's swap pair both tests are evaluated
// (lanes 0 and 1) and the swap resolved.  Returns the slot; `out` (LDS, may be null) gets the rest, written by lane 0.
__device__ __attribute__((noinline)) int decide(const DevSamplerArgs *ga, const FusedArgs *gf, int m, long it, int q, int settled, Decided *out) {
    const ConstArgs &a = *(const ConstArgs *)uniform_ptr(ga);
    const ConstFused &f = *(const ConstFused *)uniform_ptr(gf);
    const int lane = threadIdx.x & 63, C = a.C;
    if (settled) {  // first launch of a stretch: nothing is pending, the chain's slot was named by the launch that settled it
        const int s = f.slot[q * C + m] & 0xffff;
        if (out && lane == 0) {
            Decided d;
            d.slot = s; d.src = m; d.src_acc = 0; d.src_par = q; d.src_ps = 0; d.swap_first = -1; d.swapped = 0; d.r = 0;
            d.o.acc = 0; d.o.r = 0; d.o.logL = 0; d.o.logPr = 0; d.o.logPost = 0;
            *out = d;
        }
        return s;
    }
    const int p = q ^ 1, ntiles = a.ntiles;
    const long itp = it - 1;
    int A = -1;
    double u = 0;
    if (is_swap_iter(a, itp)) A = swap_first(a, itp, &u);
    const bool in_pair = A >= 0 && (m == A || m == A + 1);
    const int j0 = in_pair ? A : m;
    const int jl = (in_pair && lane == 1) ? A + 1 : j0;  // lane 1 tests the pair's second chain, every other lane repeats lane 0
    // every load before any arithmetic (one memory round trip): the scalars of this lane's chain, the partial sums of one or two chains
    const int ps = f.slot[p * C + jl] & 0xffff, st = f.prop_st[p * C + jl];
    const double pl = f.prop_logPr[p * C + jl], hL = a.logL_cur[p * C + jl], hP = a.logPr_cur[p * C + jl], hQ = a.logPost_cur[p * C + jl];
    const double Tj = a.Tcoefs[jl], il = a.init_logL[jl];
    const double *b0 = f.part + ((size_t)p * C + j0) * ntiles * 2;
    double S0, S1 = 0;
    if (ntiles <= TB) {  // the usual case, both chains' loads in flight together
        double v1[TB / 64], v2[TB / 64], w1[TB / 64], w2[TB / 64];
#pragma unroll
        for (int k = 0; k < TB / 64; k++) {
            const int t = k * 64 + lane;
            const bool in = t < ntiles;
            v1[k] = in ? b0[2 * t] : 0.0;
            v2[k] = in ? b0[2 * t + 1] : 0.0;
            w1[k] = (in && in_pair) ? b0[2 * (ntiles + t)] : 0.0;
            w2[k] = (in && in_pair) ? b0[2 * (ntiles + t) + 1] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < TB / 64; k++) { v1[k] = 0.0 + v1[k]; v2[k] = 0.0 + v2[k]; w1[k] = 0.0 + w1[k]; w2[k] = 0.0 + w2[k]; }  // (wave_partial_sum's first round)
        S0 = wave_sum_in_order(v1, v2);
        if (in_pair) S1 = wave_sum_in_order(w1, w2);
    } else {
        S0 = wave_partial_sum(b0, ntiles);
        if (in_pair) S1 = wave_partial_sum(b0 + (size_t)ntiles * 2, ntiles);
    }
    AcceptOut o = mh_outcome(a, jl, itp, (in_pair && lane == 1) ? S1 : S0, pl, st, hL, hP, hQ, Tj, il);
    AcceptOut o0, o1;
    o0.acc = __shfl(o.acc, 0, 64); o0.r = __shfl(o.r, 0, 64); o0.logL = __shfl(o.logL, 0, 64); o0.logPr = __shfl(o.logPr, 0, 64);
    o0.logPost = __shfl(o.logPost, 0, 64);
    const int ps0 = __shfl(ps, 0, 64);
    Decided d;
    d.swap_first = -1; d.swapped = 0;
    if (!in_pair) {
        d.slot = 2 * m + o0.acc; d.src = m; d.src_acc = o0.acc; d.src_par = p; d.src_ps = ps0; d.r = o0.r; d.o = o0;
    } else {
        o1.acc = __shfl(o.acc, 1, 64); o1.r = __shfl(o.r, 1, 64); o1.logL = __shfl(o.logL, 1, 64); o1.logPr = __shfl(o.logPr, 1, 64);
        o1.logPost = __shfl(o.logPost, 1, 64);
        const int ps1 = __shfl(ps, 1, 64);
        const int accA = o0.acc, accB = o1.acc;
        const double rA = o0.r, rB = o1.r;
        const int swapped = resolve_swap(a, A, u, o0, o1);  // (o0 = chain A's, o1 = chain B's: re-tempered in place)
        d.swap_first = A; d.swapped = swapped; d.src_par = p;
        const int B = A + 1;
        if (swapped) {  // each side continues from the other's post-test position: the extra candidate slots 2C .. 2C+3 (+4: second block)
            if (m == A) { d.slot = 2 * C + (A >= f.xsplit ? 4 : 0) + accB; d.src = B; d.src_acc = accB; d.src_ps = ps1; d.r = rB; d.o = o0; }
            else { d.slot = 2 * C + (B >= f.xsplit ? 4 : 0) + 2 + accA; d.src = A; d.src_acc = accA; d.src_ps = ps0; d.r = rA; d.o = o1; }
        } else {
            if (m == A) { d.slot = 2 * A + accA; d.src = A; d.src_acc = accA; d.src_ps = ps0; d.r = rA; d.o = o0; }
            else { d.slot = 2 * B + accB; d.src = B; d.src_acc = accB; d.src_ps = ps1; d.r = rB; d.o = o1; }
        }
    }
    if (out && lane == 0) *out = d;
    return d.slot;
}

// Chain m's workgroup of launch `it` (one wave): writes what iteration it-1 decided -- the chain's state for iteration `it` (parity q),
// the record of iteration it-1 (update_buffer_params / update_buffer_stat_criteria, MALA.cpp:708-710), the move flags and counters -- and,
// for the launch of iteration it+1, the slot, prior and status of the chain's proposal at iteration `it`.  With ST_COMMIT alone (after the
// last iteration of a stretch) the launch holds nothing else.
__device__ __attribute__((noinline)) void commit_chain(const DevSamplerArgs *ga, const FusedArgs *gf, int m, long it, int q, int settled, long rec,
                                                       Decided *dec) {
    const int slot = decide(ga, gf, m, it, q, settled, dec);
    __syncthreads();
    const ConstArgs &a = *(const ConstArgs *)uniform_ptr(ga);
    const ConstFused &f = *(const ConstFused *)uniform_ptr(gf);
    const int lane = threadIdx.x, C = a.C, Nv = a.Nv, Np = a.desc.Np;
    const Decided d = *dec;
    if (lane == 0) {
        const size_t gs = (size_t)(it % 3) * f.NS + slot;
        const int stP0 = f.cand_stP[2 * gs], stP1 = f.cand_stP[2 * gs + 1], stR = f.cand_stR[gs];
        const double lp = f.cand_rej[gs] ? -INFINITY : f.cand_logPr[2 * gs] + f.cand_logPr[2 * gs + 1];
        const int stm = stP0 != TAMCMC_OK ? stP0 : (stP1 != TAMCMC_OK ? stP1 : stR);
        f.prop_logPr[q * C + m] = lp;
        f.prop_st[q * C + m] = stm;
        // The test of iteration `it` (mh_outcome) as a threshold on S = sum of the tiles' partials, for the next launch's quick_decide:
        // accept <=> log u <= -pl S / T + logPr - logPost_cur <=> S <= S*.  Everything but S is known here.
        double *w = f.quick + ((size_t)q * C + m) * QN;
        const double Tm = a.Tcoefs[m];
        double Sstar = 0, ok = -1;  // (-1: no shortcut, decide() it)
        double u, u1;
        rng_uniform2(a.seed, RNG_ACCEPT, (uint32_t)m, (uint64_t)it, 0, u, u1);
        if (stm == TAMCMC_OK && !(lp == -INFINITY || isnan(lp))) {
            const double cur = settled ? a.logPost_cur[q * C + m] : d.o.logPost;
            Sstar = -((log(u) - lp + cur) * Tm) / (double)a.pl;
            if (isfinite(Sstar)) ok = 1;
        } else if (u > 0.0) ok = 2;  // r = 0 whatever the sums (mh_outcome): rejected
        w[0] = Sstar; w[1] = ok; w[2] = -(double)a.pl / Tm; w[3] = settled ? a.logL_cur[q * C + m] : d.o.logL;
        double lus = 0, k1 = 0, k2 = 0;
        if (is_swap_iter(a, it)) {  // the swap test of iteration `it`, left by the pair's first chain: u <= exp(LA TA/TB + LB TB/TA - LA - LB)
            double us;
            if (swap_first(a, it, &us) == m) {
                const double TB = a.Tcoefs[m + 1];
                lus = log(us); k1 = Tm / TB - 1.0; k2 = TB / Tm - 1.0;
            }
        }
        w[4] = lus; w[5] = k1; w[6] = k2; w[7] = (double)slot;
    }
    if (settled) return;
    const double *sv, *sp;
    if (d.src_acc) {
        const size_t gp = (size_t)((it - 1) % 3) * f.NS + d.src_ps;
        sv = f.cand_vars + gp * Nv;
        sp = f.cand_params + gp * Np;
    } else {
        sv = a.vars_cur + ((size_t)d.src_par * C + d.src) * Nv;
        sp = a.params_cur + ((size_t)d.src_par * C + d.src) * Np;
    }
    double *dv = a.vars_cur + ((size_t)q * C + m) * Nv, *dp = a.params_cur + ((size_t)q * C + m) * Np;
    double *rv = (a.samples && rec >= 0) ? a.samples + ((size_t)rec * C + m) * Nv : nullptr;
    for (int i = lane; i < Nv; i += 64) { const double v = sv[i]; dv[i] = v; if (rv) rv[i] = v; }
    for (int i = lane; i < Np; i += 64) dp[i] = sp[i];
    if (lane == 0) {
        a.logL_cur[q * C + m] = d.o.logL;
        a.logPr_cur[q * C + m] = d.o.logPr;
        a.logPost_cur[q * C + m] = d.o.logPost;
        f.slot[q * C + m] = slot;
        a.moved[m] = d.src_acc;
        a.Pmove[m] = d.r;
        if (m == 0 && d.src_acc) a.counters[1] += 1;
        a.counters[8 + m] += d.src_acc;
        if (m == 0) a.counters[0] = it;
        if (d.swap_first == m) {  // (the pair's first chain counts the swap step)
            atomicAdd((unsigned long long *)&a.counters[2], 1ull);  // (the two chain groups' launches run side by side)
            if (d.swapped) atomicAdd((unsigned long long *)&a.counters[3], 1ull);
        }
        if (a.stats && rec >= 0) {
            double *r = a.stats + ((size_t)rec * C + m) * 3;
            r[0] = d.o.logL; r[1] = d.o.logPr; r[2] = d.o.logPost;
        }
    }
}

// decide() for the workgroups that only need to know WHERE chain m stands -- the likelihood tiles (its table slot), the candidate roles
// (slot and the vector the chain continues from) -- the cheapest way that is still certain.  The MH test of iteration it-1 is a
// comparison of S = the sum of launch it-1's partials with a threshold S* that the previous launch's commit workgroup has left
// (commit_chain: everything in the test but S is known one launch earlier).  S is summed here in any order; when it is further from S*
// than every rounding involved could explain (summation: n eps sum|v| ~ 2e-14 sum|v|; the threshold and the test's own exp / division:
// a few eps of |S*|; the margin is 1e-11 of those magnitudes) the outcome is the exact test's.  The swap test of iteration it-1's pair
// (pairA, named by the host: the same Philox draw) is taken the same way: log u against LA (TA/TB - 1) + LB (TB/TA - 1) with the
// post-test likelihoods from the approximate sums.  Otherwise -- about once in 1e5 tests -- decide() evaluates everything as written.
// A decide() of ~2000 dependent instructions costs a lone wave 5 us at the head of the launch's longest chains; this one ~0.5 us.
// (the shortcut itself, a leaf function: -1 = undecided)
__device__ __attribute__((noinline)) int quick_decide_leaf(const DevSamplerArgs *ga, const FusedArgs *gf, int m, int q, int pairA, Decided *out) {
    const ConstArgs &a = *(const ConstArgs *)uniform_ptr(ga);
    const ConstFused &f = *(const ConstFused *)uniform_ptr(gf);
    const int lane = threadIdx.x & 63, C = a.C;
    const int p = q ^ 1, n2 = 2 * a.ntiles;
    const bool in_pair = pairA >= 0 && (m == pairA || m == pairA + 1);
    const int j0 = in_pair ? pairA : m;
    const double *b0 = f.part + ((size_t)p * C + j0) * n2;
    const double *r0 = f.quick + ((size_t)p * C + j0) * QN;
    // every load first: the records (lane k < QN: field k of chain j0, lane QN + k: of chain j0 + 1), the partial sums
    const double rec = (lane < (in_pair ? 2 * QN : QN)) ? r0[lane] : 0.0;
    double s0 = 0, a0 = 0, s1 = 0, a1 = 0;
    for (int t0 = 0; t0 < n2; t0 += 512) {
        double v[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = t0 + k * 64 + lane;
            v[k] = t < n2 ? b0[t] : 0.0;
            w[k] = (in_pair && t < n2) ? b0[n2 + t] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { s0 += v[k]; a0 += fabs(v[k]); s1 += w[k]; a1 += fabs(w[k]); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off, 64); a0 += __shfl_xor(a0, off, 64); }
    if (in_pair) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); a1 += __shfl_xor(a1, off, 64); }
    }
    const double St0 = __shfl(rec, 0, 64), ok0 = __shfl(rec, 1, 64);
    const int ps0 = (int)__shfl(rec, 7, 64);
    // kind 1: threshold test; kind 2: the proposal cannot be accepted (outside a prior's support, or its table failed: r = 0 and u > 0) --
    // its partial sums may be anything (an empty slot's tiles are skipped)
    int acc0 = 0;
    if (ok0 == 2.0) { s0 = 0; a0 = 0; }
    else if (ok0 > 0 && fabs(s0 - St0) > 1e-11 * (a0 + fabs(St0))) acc0 = s0 < St0 ? 1 : 0;  // (a NaN sum fails the comparison)
    else return -1;
    Decided d;
    d.swap_first = -1; d.swapped = 0; d.src_par = p; d.r = 0;
    d.o.acc = 0; d.o.r = 0; d.o.logL = 0; d.o.logPr = 0; d.o.logPost = 0;  // (the scalars are the commit workgroup's business: decide())
    if (!in_pair) { d.slot = 2 * m + acc0; d.src = m; d.src_acc = acc0; d.src_ps = ps0; }
    else {
        const double St1 = __shfl(rec, QN, 64), ok1 = __shfl(rec, QN + 1, 64);
        const int ps1 = (int)__shfl(rec, QN + 7, 64);
        int acc1 = 0;
        if (ok1 == 2.0) { s1 = 0; a1 = 0; }
        else if (ok1 > 0 && fabs(s1 - St1) > 1e-11 * (a1 + fabs(St1))) acc1 = s1 < St1 ? 1 : 0;
        else return -1;
        const double c0 = __shfl(rec, 2, 64), c1 = __shfl(rec, QN + 2, 64);
        const double LA = acc0 ? c0 * s0 : __shfl(rec, 3, 64), LB = acc1 ? c1 * s1 : __shfl(rec, QN + 3, 64);
        const double lus = __shfl(rec, 4, 64), x = LA * __shfl(rec, 5, 64) + LB * __shfl(rec, 6, 64);
        if (!(fabs(x - lus) > 1e-11 * (fabs(c0) * a0 + fabs(c1) * a1 + fabs(LA) + fabs(LB)))) return -1;
        const int swapped = x > lus ? 1 : 0;
        const int A = pairA, B = pairA + 1;
        d.swap_first = A; d.swapped = swapped;
        if (swapped) {
            if (m == A) { d.slot = 2 * C + (A >= f.xsplit ? 4 : 0) + acc1; d.src = B; d.src_acc = acc1; d.src_ps = ps1; }
            else { d.slot = 2 * C + (B >= f.xsplit ? 4 : 0) + 2 + acc0; d.src = A; d.src_acc = acc0; d.src_ps = ps0; }
        } else {
            if (m == A) { d.slot = 2 * A + acc0; d.src = A; d.src_acc = acc0; d.src_ps = ps0; }
            else { d.slot = 2 * B + acc1; d.src = B; d.src_acc = acc1; d.src_ps = ps1; }
        }
    }
    if (out && lane == 0) *out = d;
    return d.slot;
}

__device__ __forceinline__ int quick_decide(const DevSamplerArgs *ga, const FusedArgs *gf, int m, long it, int q, int settled, int pairA,
                                            Decided *out) {
    if (settled) return decide(ga, gf, m, it, q, 1, out);
    const int s = quick_decide_leaf(ga, gf, m, q, pairA, out);
    return s >= 0 ? s : decide(ga, gf, m, it, q, 0, out);
}

// Hook of the likelihood tiles of the fused step: evaluation b = chain first + b.
struct StepTiles {
    const DevSamplerArgs *ga;
    const FusedArgs *gf;
    long it;
    int q, first, settled, pairA;
    static constexpr bool coherent_partials = false;
    __device__ __forceinline__ int slot(const LoglikeArgs &, int b) const { return quick_decide(ga, gf, first + b, it, q, settled, pairA, nullptr); }
    __device__ __forceinline__ void operator()(int, int, int) const {}
};

// The three kinds of work on one candidate (see candidate_role); the proposal vector is in LDS.
// (They are real function calls -- see candidate_role -- so their arguments are pointers to the DEVICE-MEMORY copies of the argument
// blocks: a reference to a kernel argument would have to be copied to the scratch stack first.)
__device__ __attribute__((noinline)) void role_prior(const DevSamplerArgs *ga, const FusedArgs *gf, size_t gs, int h, const double *s_vars,
                                                     const double *s_params, const UnpackLds *Up) {
    const DevSamplerArgs &a = *ga;
    const FusedArgs &f = *gf;
    const UnpackLds U = *Up;
    const int Nv = a.Nv, Np = a.desc.Np, tid = threadIdx.x;
    if (h == 0) {
        for (int i = tid; i < Nv; i += 64) f.cand_vars[gs * Nv + i] = s_vars[i];
        for (int i = tid; i < Np; i += 64) f.cand_params[gs * Np + i] = s_params[i];
    }
    int rej = 0;
#ifdef TAMCMC_PROBE
    long ps_[4] = {0, 0, 0, 0};
    const double fh = wave_log_prior_part(a.desc, s_params, U, TB - 128, h, &rej, ps_);
    if (tid == 0 && (int)(gs % f.NS) == 2) { long *w = a.counters + 8 + a.C + 16 * h + 4; w[0] += ps_[1] - ps_[0]; w[1] += ps_[2] - ps_[1]; }
#else
    const double fh = wave_log_prior_part(a.desc, s_params, U, TB - 128, h, &rej);  // the proposal kernel's 128 term lanes (dev_unpack.h)
#endif
    if (tid == 0) {
        f.cand_logPr[2 * gs + h] = fh;
        f.cand_stP[2 * gs + h] = *U.status;
        if (h == 0) f.cand_rej[gs] = rej;
    }
}
__device__ __forceinline__ TablePtrs candidate_tables(const DevSamplerArgs &a, const FusedArgs &f, int q_dst) {
    TablePtrs T;
    T.mults = f.mults + (size_t)q_dst * f.NS * a.desc.per; T.pairs = f.pairs + (size_t)q_dst * 2 * f.NS; T.nh = f.nh + (size_t)q_dst * f.NS;
    T.nn = f.nn + (size_t)q_dst * f.NS; T.noise = f.noise + (size_t)q_dst * f.NS * a.desc.stride;
    T.bg = nullptr; T.ntiles = a.ntiles; T.tile_bins = a.tile_bins;
    return T;
}
__device__ __attribute__((noinline)) void role_rows(const DevSamplerArgs *ga, const FusedArgs *gf, int q_dst, int slot, size_t gs,
                                                    const double *s_params, const UnpackLds *Up) {
    const DevSamplerArgs &a = *ga;
    const FusedArgs &f = *gf;
    const UnpackLds U = *Up;
    if (threadIdx.x == 0) mt::shared_scalars_base(a.desc.model_id, s_params, a.desc.plength, *U.S);
    __syncthreads();
    const TablePtrs T = candidate_tables(a, f, q_dst);
    // the table is built whatever the prior says (this role does not know it): a vector outside a prior's support is rejected by
    // the settle step before its likelihood is looked at (model_def.cpp:476-480), a table that cannot be built leaves an empty slot
#ifdef TAMCMC_PROBE
    __shared__ long ps_[8];
    wg_unpack(a.desc, s_params, U, slot, T, true, false, false, true, ps_);
    __syncthreads();
    if (threadIdx.x == 0 && (int)(gs % f.NS) == 2) {
        long *w = a.counters + 8 + a.C + 8 + 4; w[0] += ps_[1] - ps_[0]; w[1] += ps_[2] - ps_[1];
        long *v = a.counters + 8 + a.C + 32; v[0] += ps_[5] - ps_[4]; v[1] += ps_[6] - ps_[5]; v[2] += ps_[7] - ps_[6]; v[3] += 1;
    }
#else
    wg_unpack(a.desc, s_params, U, slot, T, true, false, false, true);
#endif
    if (threadIdx.x == 0) f.cand_stR[gs] = *U.status;
}
__device__ __attribute__((noinline)) void role_background(const DevSamplerArgs *ga, const FusedArgs *gf, int q_dst, int slot, int role,
                                                          const double *s_params, const UnpackLds *Up) {
    const DevSamplerArgs &a = *ga;
    const FusedArgs &f = *gf;
    const UnpackLds U = *Up;
    if (!f.bg) return;
    if (threadIdx.x == 0) mt::shared_scalars_base(a.desc.model_id, s_params, a.desc.plength, *U.S);
    __syncthreads();
    TablePtrs T = candidate_tables(a, f, q_dst);
    T.bg = f.bg + (size_t)q_dst * f.NS * a.ntiles * bg::NH;
    const int quarter = (a.ntiles + 3) / 4, k = role - 3;
    wg_bg_tiles(a.desc, s_params, U.S, slot, T, 0, 64, k * quarter, (k + 1) * quarter);
}

// L z of chain `m` for iteration `itn` into f.lz[parity q_dst] (same streams, same row sums as propose_common), one wave.
// (two separate functions, like the candidate roles: each stays within the register budget of the tile path)
__device__ __attribute__((noinline)) void lz_normals(const DevSamplerArgs *ga, long itn, int m, double *s_z) {
    normals_into(*ga, m, itn, s_z);
}
__device__ __attribute__((noinline)) void lz_rows(const DevSamplerArgs *ga, const FusedArgs *gf, int q_dst, int m, const double *s_z) {
    Lz_rows_wave(*ga, m, s_z, gf->lz + ((size_t)q_dst * ga->C + m) * ga->Nv);
}
__device__ __forceinline__ void lz_block(const DevSamplerArgs *ga, const FusedArgs *gf, long itn, int q_dst, int m, unsigned char *lds) {
    double *s_z = (double *)lds;
    lz_normals(ga, itn, m, s_z);
    __syncthreads();
    lz_rows(ga, gf, q_dst, m, s_z);
}

// One role of one candidate slot of iteration `itn`, by ONE wave.  Slot s < 2C: chain s/2, built on the position it enters iteration
// itn-1 with (even) or on its proposal of iteration itn-1 (odd); slots 2C..2C+3 (only when itn-1 swaps a pair A,B): chain A on B's two
// vectors, chain B on A's two.  Roles: 0 = position + first half of the log-prior (and the hard constraints), 1 = table rows + noise row,
// 2 = second half of the log-prior, 3..6 = background series of a quarter of the tiles each, 7 = none.  Every role re-derives the proposal vector itself (no communication between the roles), and -- inside a
// stretch -- first decides iteration itn-2 for the chain it builds on (decide(): where that chain stands at itn-1, which slot it proposes).
// entry: the candidates of iteration itn itself from the settled chains (state parity q_src), even slots only.
__device__ void candidate_role(const DevSamplerArgs &a, const FusedArgs &f, const DevSamplerArgs *ga, const FusedArgs *gf, long itn, int q_src,
                               int q_dst, int slot, int role, bool entry, int settled, int pairA, unsigned char *lds, Decided *dec) {
    if (role > 6) return;
    const int C = a.C, Nv = a.Nv, Np = a.desc.Np, tid = threadIdx.x;
    const int e_dst = (int)(itn % 3);
    int m, src, on_prop;
    if (slot < 2 * C) { m = slot >> 1; src = m; on_prop = slot & 1; }
    else {  // slot = 2C + e, e = 0..3: the pair's cross candidates, stored in the pair's extra block
        if (entry || !is_swap_iter(a, itn - 1)) return;
        const int A = swap_first(a, itn - 1, nullptr), e = slot - 2 * C;
        m = (e < 2) ? A : A + 1;
        src = (e < 2) ? A + 1 : A;
        on_prop = e & 1;
        slot += (m >= f.xsplit) ? 4 : 0;  // in the extra block of the chain that will use it (the group that owns that block never runs
                                          // ahead of itself; the OTHER group's launches may be several iterations ahead)
    }
    if (entry && on_prop) return;  // a stretch starts from settled chains: there is no pending proposal to build on
    if (entry && role == 0 && tid == 0) f.slot[q_src * C + m] = 2 * m;
#ifdef TAMCMC_PROBE
    long pt[6];
    pt[0] = (long)wall_clock64();
#define RSTAMP(k) pt[k] = (long)wall_clock64()
#else
#define RSTAMP(k)
#endif
    double *s_params = (double *)lds;
    double *s_vars = s_params + Np;
    double *s_z = s_vars + Nv;
    const UnpackLds U = carve_unpack_lds((unsigned char *)(s_z + Nv + 1));
    __shared__ UnpackLds s_U;  // handed to the role functions by address
    if (tid == 0) s_U = U;
    const double *bv, *bp;
    if (entry) {
        bv = a.vars_cur + ((size_t)q_src * C + src) * Nv;
        bp = a.params_cur + ((size_t)q_src * C + src) * Np;
    } else {
        const int ps = quick_decide(ga, gf, src, itn - 1, q_src, settled, pairA, dec);
        __syncthreads();
        const Decided d = *dec;
        if (on_prop) {  // src's proposal of iteration itn-1
            const size_t gp = (size_t)((itn - 1) % 3) * f.NS + ps;
            bv = f.cand_vars + gp * Nv;
            bp = f.cand_params + gp * Np;
        } else if (d.src_acc) {  // src enters iteration itn-1 at a proposal of iteration itn-2 that was accepted
            const size_t gp = (size_t)((itn - 2) % 3) * f.NS + d.src_ps;
            bv = f.cand_vars + gp * Nv;
            bp = f.cand_params + gp * Np;
        } else {
            bv = a.vars_cur + ((size_t)d.src_par * C + d.src) * Nv;
            bp = a.params_cur + ((size_t)d.src_par * C + d.src) * Np;
        }
    }
    RSTAMP(1);
    // everything the proposal vector is made of in ONE memory round trip: the base vectors, L z(itn) of chain m (q_dst: iteration itn's
    // parity; computed one launch ahead, lz_block), the scatter indices, the polynomial table
    const double *lz = f.lz + ((size_t)q_dst * C + m) * Nv;
    constexpr int PW = (int)(sizeof(mt::PolyTab) / sizeof(double));
    if (Nv <= 128 && Np <= 128 && PW <= 256) {
        double r_v[2], r_z[2], r_p[2], r_t[4];
        int r_i[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int i = tid + 64 * e;
            r_v[e] = i < Nv ? bv[i] : 0.0; r_z[e] = i < Nv ? lz[i] : 0.0; r_i[e] = i < Nv ? a.index_to_relax[i] : 0;
            r_p[e] = i < Np ? bp[i] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) { const int i = tid + 64 * e; r_t[e] = i < PW ? ((const double *)a.desc.poly)[i] : 0.0; }
#pragma unroll
        for (int e = 0; e < 2; e++) { const int i = tid + 64 * e; if (i < Np) s_params[i] = r_p[e]; }
#pragma unroll
        for (int e = 0; e < 4; e++) { const int i = tid + 64 * e; if (i < PW) ((double *)U.poly)[i] = r_t[e]; }
        if (tid == 0) { *U.status = TAMCMC_OK; *U.reject = 0; }  // (unpack_begin)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int i = tid + 64 * e;
            if (i < Nv) { const double v = r_v[e] + 0.0 + r_z[e]; s_vars[i] = v; s_params[r_i[e]] = v; }  // same expression as propose_common; update_params_with_vars
        }
        __syncthreads();
    } else {
        for (int i = tid; i < Nv; i += 64) s_vars[i] = bv[i];
        for (int i = tid; i < Np; i += 64) s_params[i] = bp[i];
        unpack_begin(a.desc, U);  // (barrier)
        for (int i = tid; i < Nv; i += 64) s_vars[i] = s_vars[i] + 0.0 + lz[i];
        __syncthreads();
        for (int k = tid; k < Nv; k += 64) s_params[a.index_to_relax[k]] = s_vars[k];
        __syncthreads();
    }
    RSTAMP(2);
    const size_t gs = (size_t)e_dst * f.NS + slot;
    // (three separate functions: inlined side by side the roles' code raises the whole kernel's register allocation above the
    // three-waves-per-SIMD budget of the tile path)
#ifdef TAMCMC_PROBE
    if (a.probe & (0x100 << (role > 2 ? 2 : (role == 2 ? 0 : role)))) return;  // timing experiments: one kind of role left out
#endif
    if (role == 0 || role == 2) role_prior(ga, gf, gs, role >> 1, s_vars, s_params, &s_U);
    else if (role == 1) role_rows(ga, gf, e_dst, slot, gs, s_params, &s_U);
    else role_background(ga, gf, e_dst, slot, role, s_params, &s_U);
#ifdef TAMCMC_PROBE
    __syncthreads();
    RSTAMP(3);
    if (tid == 0 && slot == 2 && !entry && role < 4) {  // one slot's first four roles: decide | vectors, L z | the role itself (units of 10 ns)
        long *w = a.counters + 8 + C + 8 * role;
        w[0] += pt[1] - pt[0]; w[1] += pt[2] - pt[1]; w[2] += pt[3] - pt[2]; w[3] += 1;
    }
#endif
#undef RSTAMP
}

// Per-launch scalars of the fused step.
struct StepCtl {
    long it, rec, it_lz;   // iteration of the tiles; record index of iteration it-1 (-1: none); first iteration of the L z blocks
    int q, flags;          // parity of iteration `it`; ST_* bits
    int nbr, nlz;          // workgroups reserved for candidate roles / L z blocks + commits (multiples of 8: keeps the tiles' XCD mapping)
    int n_lz_live, q_lz;   // L z blocks that have work (chain first + e % cnt of iteration it_lz + e / cnt); parity of it_lz
    int first, cnt;        // the chains of this launch: [first, first + cnt) -- all of them, or one chain group (see run(): fused)
    int extra;             // 1: the launch also builds the four extra candidates of its iteration's swap pair (slots 2C..2C+3)
    int pairA;             // first chain of iteration it-1's swap pair, -1: none (quick_slot)
    const DevSamplerArgs *ga;  // device-memory copies of the first two kernel arguments (for the function calls)
    const struct FusedArgs *gf;
};

// Launch `it` of a fused stretch: [0, nbr) candidate roles of iteration it+1 (ST_BR; at the entry of a stretch, ST_ENTRY: of iteration
// `it` itself from the settled chains), [nbr, nbr+nlz): L z of later iterations (ST_LZ) and, in the last cnt of them, the chains' commit
// workgroups (ST_COMMIT), then the likelihood tiles of iteration `it` (ST_L).  ST_FIRST: the chains are settled (nothing to decide).
#define TAMCMC_STEP_BODY                                                                                                      \
    __shared__ tile::TileLds<MODE, 64> lds;                                                                                  \
    __shared__ Decided s_dec;                                                                                                \
    const int id = (int)blockIdx.x;                                                                                          \
    const int settled = (c.flags & ST_FIRST) ? 1 : 0;                                                                        \
    /* the few single-wave workgroups with long dependent chains (roles, L z, commit) issue ahead of the tiles they share a SIMD with */ \
    if (id < c.nbr + c.nlz) __builtin_amdgcn_s_setprio(3);                                                                   \
    if (id < c.nbr) {                                                                                                        \
        const int k = id >> 3, slot = k < 2 * c.cnt ? 2 * c.first + k : 2 * a.C + (k - 2 * c.cnt); /* the group's slots, then the pair's */ \
        if (k >= 2 * c.cnt && !c.extra) return;                                                                              \
        if (c.flags & ST_ENTRY) candidate_role(a, f, c.ga, c.gf, c.it, c.q, c.q, slot, id & 7, true, 1, -1, (unsigned char *)&lds, &s_dec);    \
        else if (c.flags & ST_BR)                                                                                            \
            candidate_role(a, f, c.ga, c.gf, c.it + 1, c.q, c.q ^ 1, slot, id & 7, false, settled, c.pairA, (unsigned char *)&lds, &s_dec); \
        return;                                                                                                              \
    }                                                                                                                        \
    if (id < c.nbr + c.nlz) {                                                                                                \
        const int e = id - c.nbr, k = e - (c.nlz - c.cnt);                                                                   \
        if (k >= 0 && (c.flags & ST_COMMIT)) commit_chain(c.ga, c.gf, c.first + k, c.it, c.q, settled, c.rec, &s_dec);       \
        else if (e < c.n_lz_live)                                                                                            \
            lz_block(c.ga, c.gf, c.it_lz + e / c.cnt, (c.q_lz ^ (e / c.cnt)) & 1, c.first + e % c.cnt, (unsigned char *)&lds);  \
        return;                                                                                                              \
    }                                                                                                                        \
    if (c.flags & ST_L)                                                                                                      \
        tile::loglike_tile<MODE, 64, K, false, false>(la, id - c.nbr - c.nlz, lds, StepTiles{c.ga, c.gf, c.it, c.q, c.first, settled, c.pairA});
// The tile path of K <= 8 bins per lane fits 168 VGPRs = three waves per SIMD; the candidate roles (log-prior, series) would raise the
// kernel's allocation above that, so the occupancy is pinned here (those roles are separate functions, see candidate_role).
template <int MODE, int K>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) k_step(const DevSamplerArgs a, const FusedArgs f, const LoglikeArgs la,
                                                                                      const StepCtl c) {
    TAMCMC_STEP_BODY
}
template <int MODE, int K>
__global__ void __launch_bounds__(64) k_step_wide(const DevSamplerArgs a, const FusedArgs f, const LoglikeArgs la, const StepCtl c) {
    TAMCMC_STEP_BODY
}
#undef TAMCMC_STEP_BODY

// ev0 / ev1 (optional): events stamped at the kernel's own start and end (hipExtLaunchKernelGGL) -- the duration rocprofv3 reports for a
// dispatch, without the time the launch waits in its stream
template <int MODE>
bool launch_step_k(int K, int grid, hipStream_t st, const DevSamplerArgs &a, const FusedArgs &f, const LoglikeArgs &la, const StepCtl &c,
                   hipEvent_t ev0, hipEvent_t ev1) {
    if (K == 4) hipExtLaunchKernelGGL((k_step<MODE, 4>), dim3(grid), dim3(64), 0, st, ev0, ev1, 0, a, f, la, c);
    else if (K == 8) hipExtLaunchKernelGGL((k_step<MODE, 8>), dim3(grid), dim3(64), 0, st, ev0, ev1, 0, a, f, la, c);
    else if (K == 16) hipExtLaunchKernelGGL((k_step_wide<MODE, 16>), dim3(grid), dim3(64), 0, st, ev0, ev1, 0, a, f, la, c);
    else return false;
    return true;
}
hipError_t launch_step(int mode, int K, int grid, hipStream_t st, const DevSamplerArgs &a, const FusedArgs &f, const LoglikeArgs &la,
                       const StepCtl &c, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    if (grid <= 0) return hipSuccess;
    bool ok;
    if (mode == TAMCMC_PRECISION_FAST) ok = launch_step_k<tile::M_FAST>(K, grid, st, a, f, la, c, ev0, ev1);
    else if (mode == TAMCMC_PRECISION_FAST_DIRECT) ok = launch_step_k<tile::M_FAST_DIRECT>(K, grid, st, a, f, la, c, ev0, ev1);
    else ok = launch_step_k<tile::M_STRICT>(K, grid, st, a, f, la, c, ev0, ev1);
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}

#include "dev_mala_impl.h"

}  // namespace

// ---------------------------------------------------------------------------------------------------------------

struct DevSampler::Impl {
    tamcmc_hip_ctx *ctx = nullptr;
    DevSamplerArgs a{};
    std::vector<void *> allocs;
    hipEvent_t ev[64][2];
    int n_ev = 0;
    bool poly_ready = false;
    double *adapt_scratch = nullptr;
    size_t smp_cap = 0, stat_cap = 0;
    size_t lds_base = 0, lds_adapt = 0;
    int parity = 0;  // which of the two state buffers holds the chains' current state
    // chain groups: the chains are split into G contiguous groups, each on its own stream, so that one group's k_iterate
    // overlaps the other groups' k_loglike (an iteration is a serial k_iterate -> k_loglike chain per group)
    bool pre_lz = true;  // (B): spare workgroups compute L z one iteration ahead while L is frozen
    FusedArgs f{};       // (A): candidate slots, tickets
    unsigned char *d_argcopy = nullptr;  // device image of {DevSamplerArgs, FusedArgs} as last launched, and its host shadow
    std::vector<unsigned char> h_argcopy;
    // Langevin step (use_drift): the finite-difference batch object, its device block and scratch, the per-chain work arrays
    bool use_drift = false;
    double delta = 0, fd_step_rel = 1e-7;
    FdBatch fd;
    DevBuf<unsigned char> fd_block;
    DevBuf<double> fd_part, fd_S, fd_model, fd_bg;
    DevBuf<double> fused_bg, fused_part;  // (A): the candidates' background series, the tiles' partial sums by parity (see run())
    MalaArgs mala{};
    bool grad_valid = false;
    int prior_class = 0, model_id = 0;
    std::vector<double> h_priors, h_extra;
    std::vector<int32_t> h_idx, h_sw;
    // (A) carried over between run() calls: the last launch of a fused stretch also prepares the candidates of the iteration that
    // follows and the L z of the one after; a call that continues right there starts without the two entry launches
    long armed_it = -1;
    int armed_q = 0;
    long it_fused = 0, it_lockstep = 0;  // iterations run by each scheme since creation (tamcmc_sampler_get_info)
    int mala_chol_lds = -1;

    hipEvent_t gev[8][2];  // fused step with two chain groups: event pairs around sampled launches of the second group (on its stream)
    int n_gev = 0;
    bool rgb = false;  // ids 25 / 27: k_iterate leaves the table to the pre-step kernels (rgb_device_stage), lockstep scheme
    int rgb_bmax = 0;  // chains per workspace slice (one slice per chain group)
    bool fused_ok = false;
    int fused_mode = -1, fused_K = 0;  // the geometry the (A) buffers were sized for
    int tile_rot = 0;  // launch-order hint of k_loglike (first near-field tile of chain 0's initial table)
    std::vector<int32_t> h_plength;
    int G = 1;
    hipStream_t gst[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_kb[4], ev_ki[4], ev_fork, ev_join[4];
    bool ev_made = false;
    double *d_pack = nullptr, *h_pack = nullptr;  // state download: device gather block and its pinned host image

    // The caller's record buffer as the device sees it when it is pinned, mapped host memory (tamcmc_hip_host_alloc): the settle step then
    // writes the records straight into it (15 KB per iteration over PCIe, posted) and a call ends without its two device-to-host copies.
    // (asked on every call: an address says nothing about what the caller has freed and allocated since the last one)
    double *device_view(const double *host, size_t bytes) {
        if (!host || !bytes) return nullptr;
        double *d = nullptr;
        void *dp = nullptr, *dq = nullptr;
        const char *last = (const char *)host + bytes - 1;
        // hipHostGetDevicePointer fails for pageable memory and returns the device address of THIS address for page-locked, mapped memory.
        // The whole record block [host, host + bytes) must lie inside ONE mapping: the last byte has to be page-locked too and map to the
        // first byte's device address + bytes - 1 -- a pinned buffer shorter than the call's records, or an interior pointer near the
        // end of one, would otherwise make the settle step write outside the mapping (a GPU fault instead of a host-side error)
        if (hipHostGetDevicePointer(&dp, const_cast<double *>(host), 0) == hipSuccess && dp &&
            hipHostGetDevicePointer(&dq, const_cast<char *>(last), 0) == hipSuccess && dq == (char *)dp + bytes - 1)
            d = (double *)dp;
        else (void)hipGetLastError();  // (pageable memory, or not one mapping over the whole block: the staged copy is used)
        return d;
    }

    // End of a call: the host waits for a stream by polling it for up to a millisecond before it blocks.  A blocking wait parks the
    // thread on an interrupt and wakes tens of microseconds after the last kernel has finished -- a tenth of a 20-iteration call (the
    // reference writes its ring buffer every Nbuffer iterations; a caller with short buffers makes short calls).
    // (only when this is the process's one running call: several host threads polling -- co-resident stars, tamcmc_sampler_run_packed --
    // would contend for the runtime's locks with the threads that are still enqueuing)
    static hipError_t wait_stream(hipStream_t st, bool poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0; poll; spin++) {
            const hipError_t e = hipStreamQuery(st);
            if (e != hipErrorNotReady) return e;
            if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1000)) break;
        }
        (void)hipGetLastError();  // (hipErrorNotReady is sticky in the last-error slot)
        return hipStreamSynchronize(st);
    }

    template <typename T>
    hipError_t dalloc(T **p, size_t n) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
        if (e == hipSuccess) { allocs.push_back(q); *p = (T *)q; }
        return e;
    }
};

DevSampler::DevSampler() : impl(new Impl()) {}
DevSampler::~DevSampler() {
    if (!impl) return;
    if (impl->ctx) {
        (void)hipSetDevice(impl->ctx->device);
        (void)hipStreamSynchronize(impl->ctx->stream);
        // after a HIP error in the middle of a call the other chain groups' streams may still hold launches that use the buffers below
        for (int g = 1; g < 4; g++) if (impl->gst[g]) (void)hipStreamSynchronize(impl->gst[g]);
    }
#ifdef TAMCMC_PROBE
    if (impl->a.counters && getenv("TAMCMC_PROBE_STEP")) {
        long h[40];
        (void)hipMemcpy(h, impl->a.counters + 8 + impl->a.C, sizeof h, hipMemcpyDeviceToHost);
        if (h[35] > 0) fprintf(stderr, "build_multiplet, row 20 (us): decode + widths/heights %.2f | window %.2f | m loop %.2f\n", 0.01 * h[32] / h[35], 0.01 * h[33] / h[35], 0.01 * h[34] / h[35]);
        for (int r = 0; r < 4; r++)
            if (h[8 * r + 3] > 0)
                fprintf(stderr, "candidate role %d of slot 2 (us): decide %.2f | vectors + L z %.2f | role %.2f  [inside: %.2f | %.2f]  (%ld launches)\n", r,
                        0.01 * h[8 * r] / h[8 * r + 3], 0.01 * h[8 * r + 1] / h[8 * r + 3], 0.01 * h[8 * r + 2] / h[8 * r + 3],
                        0.01 * h[8 * r + 4] / h[8 * r + 3], 0.01 * h[8 * r + 5] / h[8 * r + 3], h[8 * r + 3]);
    }
    if (impl->a.counters && getenv("TAMCMC_PROBE_ADAPT")) {
        long h[8];
        (void)hipMemcpy(h, impl->a.counters, sizeof h, hipMemcpyDeviceToHost);
        if (h[7] > 0)
            fprintf(stderr, "Cholesky panels of chain 0 (us per panel): columns below %.2f | next panel's columns %.2f | next diagonal block beside the rest of the trailing update %.2f  (%ld panels)\n",
                    0.01 * h[4] / h[7], 0.01 * h[5] / h[7], 0.01 * h[6] / h[7], h[7]);
    }
#endif
    for (void *p : impl->allocs) (void)hipFree(p);
    impl->fd_block.release(); impl->fd_part.release(); impl->fd_S.release(); impl->fd_model.release(); impl->fd_bg.release(); impl->fused_bg.release(); impl->fused_part.release();
    if (impl->h_pack) (void)hipHostFree(impl->h_pack);
    for (int i = 0; i < impl->n_ev; i++) { (void)hipEventDestroy(impl->ev[i][0]); (void)hipEventDestroy(impl->ev[i][1]); }
    for (int i = 0; i < impl->n_gev; i++) { (void)hipEventDestroy(impl->gev[i][0]); (void)hipEventDestroy(impl->gev[i][1]); }
    if (impl->ev_made) {
        (void)hipEventDestroy(impl->ev_fork);
        for (int g = 0; g < 4; g++) { (void)hipEventDestroy(impl->ev_kb[g]); (void)hipEventDestroy(impl->ev_ki[g]); (void)hipEventDestroy(impl->ev_join[g]); }
        for (int g = 1; g < 4; g++) if (impl->gst[g]) (void)hipStreamDestroy(impl->gst[g]);
    }
    delete impl;
}

#define DCHK(call)                                                                   \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            c->err = std::string(#call) + ": " + hipGetErrorString(e_);              \
            return TAMCMC_ERR_HIP;                                                   \
        }                                                                            \
    } while (0)

template <typename T>
static hipError_t up(T *dst, const T *src, size_t n, hipStream_t st) {
    return hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, st);
}

int DevSampler::init(tamcmc_hip_ctx *c, const DevSamplerInit &in) {
    Impl &I = *impl;
    I.ctx = c;
    if (c->Nx <= 0) return TAMCMC_ERR_NO_SPECTRUM;
    if (in.C < 1 || in.C > TAMCMC_MAX_CHAINS) return TAMCMC_ERR_BAD_ARG;
    DCHK(hipSetDevice(c->device));
    DevSamplerArgs &a = I.a;
    a.desc.model_id = in.model_id; a.desc.prior_class = in.prior_class; a.C = in.C; a.desc.Np = in.Np; a.Nv = in.Nv;
    I.rgb = is_rgb_model(in.model_id);
#ifdef TAMCMC_PROBE
    if (const char *ep = getenv("TAMCMC_PROBE_ADAPT")) a.probe = atoi(ep);
#endif
    a.desc.per = I.rgb ? 0 : mt::count_multiplets(in.model_id, in.plength);
    I.h_plength.assign(in.plength, in.plength + 11);
    I.use_drift = in.use_drift != 0; I.delta = in.delta; I.fd_step_rel = in.fd_step_rel > 0 ? in.fd_step_rel : 1e-7;
    I.prior_class = in.prior_class; I.model_id = in.model_id;
    I.h_priors.assign(in.priors, in.priors + 4 * (size_t)in.Np); I.h_extra.assign(in.extra_priors, in.extra_priors + 10);
    I.h_idx.assign(in.index_to_relax, in.index_to_relax + in.Nv); I.h_sw.assign(in.priors_switch, in.priors_switch + in.Np);
    if (a.desc.per < 0) return TAMCMC_ERR_BAD_MODEL;
    a.desc.stride = in.plength[8] > 0 ? in.plength[8] : 1;
    if ((a.desc.stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    {
        int G = in.chain_groups > 0 ? in.chain_groups : (in.C >= 8 ? 2 : 1);
        // red giants: an iteration is a chain of four latency-bound launches per group (proposal + unpack, solver, rows, likelihood);
        // four groups keep the GPU busy while three of them are in their short kernels (C5, 40 chains: 3.4 / 3.8 / 4.0 / 4.05 k
        // iterations/s with 1 / 2 / 3 / 4 groups)
        if (in.chain_groups <= 0 && I.rgb && in.C >= 16) G = 4;
        if (G > 4) G = 4;
        if (G > in.C) G = in.C;
        I.G = G;
    }
    if (I.rgb) {
        if (I.use_drift) return TAMCMC_ERR_BAD_MODEL;  // the finite-difference builder has no red-giant pre-step
        I.rgb_bmax = (in.C + I.G - 1) / I.G;
        int rc = rgb_device_prepare(c, I.rgb_bmax, I.G, in.plength, &a.desc.per, &a.desc.stride);  // one workspace slice per chain group
        if (rc) return rc;
    }
    a.desc.Nx = (int)c->Nx;
    a.desc.x_first = c->hx[0]; a.desc.x_last = c->hx[(size_t)c->Nx - 1]; a.desc.step = c->hx[1] - c->hx[0];
    a.pl = (long)in.likelihood_params;
    a.seed = in.seed; a.dN_mixing = in.dN_mixing; a.swap_rule = in.swap_rule == 1 ? 1 : 0;
    a.c0 = in.c0; a.epsilon1 = in.epsilon1; a.epsi2 = in.epsi2; a.A1 = in.A1; a.target_acceptance = in.target_acceptance;
    const size_t C = (size_t)in.C, Np = (size_t)in.Np, Nv = (size_t)in.Nv;
    const size_t CD = C;
    hipStream_t st = c->stream;
    int *d_pl, *d_idx, *d_sw;
    double *d_pr, *d_ex, *d_T;
    DCHK(I.dalloc(&d_pl, 11)); DCHK(I.dalloc(&d_idx, Nv)); DCHK(I.dalloc(&d_sw, Np));
    DCHK(I.dalloc(&d_pr, 4 * Np)); DCHK(I.dalloc(&d_ex, 10)); DCHK(I.dalloc(&d_T, C));
    DCHK(up(d_pl, in.plength, 11, st)); DCHK(up(d_idx, in.index_to_relax, Nv, st)); DCHK(up(d_sw, in.priors_switch, Np, st));
    DCHK(up(d_pr, in.priors, 4 * Np, st)); DCHK(up(d_ex, in.extra_priors, 10, st)); DCHK(up(d_T, in.Tcoefs, C, st));
    a.desc.plength = d_pl; a.index_to_relax = d_idx; a.desc.priors_switch = d_sw; a.desc.priors = d_pr; a.desc.extra = d_ex; a.Tcoefs = d_T;
    // every per-iteration array exists twice (parity): a workgroup reads parity P and writes parity P^1
    DCHK(I.dalloc(&a.vars_cur, 2 * C * Nv)); DCHK(I.dalloc(&a.params_cur, 2 * C * Np));
    DCHK(I.dalloc(&a.vars_prop, 2 * CD * Nv)); DCHK(I.dalloc(&a.params_prop, 2 * CD * Np));
    DCHK(I.dalloc(&a.logL_cur, 2 * C)); DCHK(I.dalloc(&a.logPr_cur, 2 * C)); DCHK(I.dalloc(&a.logPost_cur, 2 * C));
    DCHK(I.dalloc(&a.init_logL, C)); DCHK(I.dalloc(&a.logPr_prop, 2 * CD)); DCHK(I.dalloc(&a.status_prop, 2 * CD));
    DCHK(I.dalloc(&a.Pmove, C)); DCHK(I.dalloc(&a.moved, C)); DCHK(I.dalloc(&a.counters, 8 + C + 64));  // (+64: timeline stamps of the probe build)
    a.grad_cur = nullptr; a.gradP_cur = nullptr;
    if (I.use_drift) {
        DCHK(I.dalloc(&a.grad_cur, 2 * C * Nv)); DCHK(I.dalloc(&a.gradP_cur, 2 * C * Nv));
        DCHK(I.dalloc(&I.mala.grad_prop, C * Nv)); DCHK(I.dalloc(&I.mala.gradP_prop, C * Nv)); DCHK(I.dalloc(&I.mala.drift_cur, C * Nv));
        DCHK(I.dalloc(&I.mala.out, C * 5));
    }
    DCHK(I.dalloc(&a.lz, 2 * C * Nv)); DCHK(I.dalloc(&a.LT, C * Nv * Nv)); DCHK(I.dalloc(&a.cov, C * Nv * Nv)); DCHK(I.dalloc(&a.mu, C * Nv)); DCHK(I.dalloc(&a.sigma, C));
    DCHK(I.dalloc(&a.mults, CD * (size_t)a.desc.per + 1)); DCHK(I.dalloc(&a.pairs, 2 * CD)); DCHK(I.dalloc(&a.nh, CD)); DCHK(I.dalloc(&a.nn, CD));
    DCHK(I.dalloc(&a.noise, CD * (size_t)a.desc.stride));
    DCHK(hipMemsetAsync(a.counters, 0, (8 + C + 64) * sizeof(long), st));
    DCHK(hipMemsetAsync(a.moved, 0, C * sizeof(int), st));
    DCHK(hipMemsetAsync(a.Pmove, 0, C * sizeof(double), st));
    a.samples = nullptr; a.stats = nullptr;
    // Cholesky workspace: LDS when (Nv^2 + Nv) doubles fit beside the iteration's own LDS, else global scratch
    I.lds_base = (Np + 2 * Nv + 1) * sizeof(double) + unpack_lds_bytes() + 32;
    I.lds_adapt = (Nv * Nv + Nv) * sizeof(double);
    a.chol_in_lds = (I.lds_base + I.lds_adapt <= 150 * 1024) ? 1 : 0;  // (k_iterate also has ~6 KB of static LDS)
    if (!a.chol_in_lds) { DCHK(I.dalloc(&I.adapt_scratch, C * (Nv * Nv + Nv))); I.lds_adapt = 0; }
    if (I.lds_base + I.lds_adapt > 64 * 1024) {
        DCHK(hipFuncSetAttribute((const void *)k_iterate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(I.lds_base + I.lds_adapt)));
        DCHK(hipFuncSetAttribute((const void *)k_iterate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(I.lds_base + I.lds_adapt)));
    }
    // polynomial tables Pslm/Qlm: computed ON the device (its own double arithmetic), read through a uniform pointer
    mt::PolyTab *d_tab;
    DCHK(I.dalloc(&d_tab, 1));
    hipLaunchKernelGGL(k_fill_poly, dim3(1), dim3(64), 0, st, d_tab);
    DCHK(hipGetLastError());
    a.desc.poly = d_tab;
    for (int i = 0; i < 64; i++) { DCHK(hipEventCreate(&I.ev[i][0])); DCHK(hipEventCreate(&I.ev[i][1])); I.n_ev = i + 1; }
    for (int i = 0; i < 8; i++) { DCHK(hipEventCreate(&I.gev[i][0])); DCHK(hipEventCreate(&I.gev[i][1])); I.n_gev = i + 1; }
    {
        const int G = I.G;
        I.gst[0] = st;
        for (int g = 1; g < G; g++) DCHK(hipStreamCreateWithFlags(&I.gst[g], hipStreamNonBlocking));
        DCHK(hipEventCreateWithFlags(&I.ev_fork, hipEventDisableTiming));
        for (int g = 0; g < 4; g++) {
            DCHK(hipEventCreateWithFlags(&I.ev_kb[g], hipEventDisableTiming));
            DCHK(hipEventCreateWithFlags(&I.ev_ki[g], hipEventDisableTiming));
            DCHK(hipEventCreateWithFlags(&I.ev_join[g], hipEventDisableTiming));
        }
        I.ev_made = true;
    }
    {  // (A) fused step: three sets (iteration mod 3) of 2C+8 candidate slots, per-chain hand-over arrays by parity (tables are sized at the first run())
        FusedArgs &f = I.f;
        f.NS = 2 * in.C + 8;
        {   // two chain groups for the fused step (see run()): with the default groups, from 8 chains on
            const int h = (int)(((long)in.C * 1) / 2);
            f.xsplit = (I.G == 2 && h >= 3 && in.C - h >= 3) ? h : in.C;
        }
        const size_t NS = (size_t)f.NS;
        DCHK(I.dalloc(&f.cand_vars, 3 * NS * Nv)); DCHK(I.dalloc(&f.cand_params, 3 * NS * Np)); DCHK(I.dalloc(&f.cand_logPr, 6 * NS));
        DCHK(I.dalloc(&f.cand_stP, 6 * NS)); DCHK(I.dalloc(&f.cand_stR, 3 * NS)); DCHK(I.dalloc(&f.cand_rej, 3 * NS));
        DCHK(I.dalloc(&f.mults, 3 * NS * (size_t)a.desc.per + 1)); DCHK(I.dalloc(&f.pairs, 6 * NS)); DCHK(I.dalloc(&f.nh, 3 * NS)); DCHK(I.dalloc(&f.nn, 3 * NS));
        DCHK(I.dalloc(&f.noise, 3 * NS * (size_t)a.desc.stride));
        DCHK(I.dalloc(&f.slot, 2 * C)); DCHK(I.dalloc(&f.prop_logPr, 2 * C)); DCHK(I.dalloc(&f.prop_st, 2 * C)); DCHK(I.dalloc(&f.quick, 2 * C * QN)); DCHK(I.dalloc(&f.lz, 2 * C * Nv));
        DCHK(hipMemsetAsync(f.nn, 0, 3 * NS * sizeof(int), st));
        DCHK(hipMemsetAsync(f.cand_stP, 0, 6 * NS * sizeof(int), st));
        DCHK(hipMemsetAsync(f.cand_rej, 0, 3 * NS * sizeof(int), st));
        DCHK(hipMemsetAsync(f.cand_stR, 0, 3 * NS * sizeof(int), st));
        f.part = nullptr;
        f.bg = nullptr;
        // the candidate roles borrow the tile workgroup's LDS: a parameter vector too long for it keeps the lockstep scheme
        const size_t role_lds = (Np + 2 * Nv + 1) * sizeof(double) + unpack_lds_bytes() + 32;
        I.fused_ok = !I.rgb && role_lds <= sizeof(tile::TileLds<tile::M_FAST_DIRECT, 64>);
    }
    DCHK(hipStreamSynchronize(st));
    return TAMCMC_OK;
}

int DevSampler::download_gradient(double *grad, double *grad_prior) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    if (!I.use_drift || !I.a.grad_cur || !I.grad_valid) return TAMCMC_ERR_BAD_ARG;
    DCHK(hipSetDevice(c->device));
    const size_t n = (size_t)I.a.C * I.a.Nv;
    DCHK(hipStreamSynchronize(c->stream));
    if (grad) DCHK(hipMemcpy(grad, I.a.grad_cur + (size_t)I.parity * n, n * sizeof(double), hipMemcpyDeviceToHost));
    if (grad_prior) DCHK(hipMemcpy(grad_prior, I.a.gradP_cur + (size_t)I.parity * n, n * sizeof(double), hipMemcpyDeviceToHost));
    return TAMCMC_OK;
}

int DevSampler::download_last_proposal(double *vars_prop, double *grad_prop) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    if (!I.use_drift) return TAMCMC_ERR_BAD_ARG;  // (the random-walk schemes keep several candidate proposals per chain, not one)
    DCHK(hipSetDevice(c->device));
    const size_t n = (size_t)I.a.C * I.a.Nv;
    DCHK(hipStreamSynchronize(c->stream));
    if (vars_prop) DCHK(hipMemcpy(vars_prop, I.a.vars_prop, n * sizeof(double), hipMemcpyDeviceToHost));
    if (grad_prop) DCHK(hipMemcpy(grad_prop, I.mala.grad_prop, n * sizeof(double), hipMemcpyDeviceToHost));
    return TAMCMC_OK;
}

void DevSampler::info(long out[8]) const {
    const Impl &I = *impl;
    out[0] = I.a.Nv; out[1] = I.a.desc.Np;
    out[2] = I.use_drift ? I.mala_chol_lds : I.a.chol_in_lds;
    out[3] = (I.fused_ok && !I.use_drift) ? 1 : 0;
    out[4] = I.G; out[5] = I.it_fused; out[6] = I.it_lockstep; out[7] = I.a.C;
}

int DevSampler::upload_state(const double *vars, const double *params, const double *logL, const double *logPr,
                             const double *logPost, const double *init_logL) {
    Impl &I = *impl;
    I.armed_it = -1;
    I.grad_valid = false;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t C = (size_t)a.C, Np = (size_t)a.desc.Np, Nv = (size_t)a.Nv;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    const size_t P = (size_t)I.parity;
    DCHK(up(a.vars_cur + P * C * Nv, vars, C * Nv, st)); DCHK(up(a.params_cur + P * C * Np, params, C * Np, st));
    DCHK(up(a.logL_cur + P * C, logL, C, st)); DCHK(up(a.logPr_cur + P * C, logPr, C, st)); DCHK(up(a.logPost_cur + P * C, logPost, C, st));
    DCHK(up(a.init_logL, init_logL, C, st));
    DCHK(hipStreamSynchronize(st));
    {  // launch-order hint from chain 0's table at the uploaded position
        std::vector<tamcmc_multiplet> tab((size_t)a.desc.per > 0 ? (size_t)a.desc.per : 1);
        std::vector<double> nz((size_t)a.desc.stride);
        int n = 0, nh = 0, nn = 0;
        const int tb = tile_bins(c->wgs, c->K);
        I.tile_rot = 0;
        if (I.rgb) {  // a little below the lowest radial mode (rgb_stage_params' rule)
            const double *fl0 = params + I.h_plength[0] + I.h_plength[1];
            const double fmin = *std::min_element(fl0, fl0 + I.h_plength[2]);
            const int ntiles = (int)((c->Nx + tb - 1) / tb);
            const double t = (fmin - a.desc.x_first) / a.desc.step / (double)tb - 3.0;
            I.tile_rot = (t > 0 && t < ntiles) ? (int)t : 0;
        } else if (build_mode_table(a.desc.model_id, params, I.h_plength.data(), c->hx.data(), c->Nx, tab.data(), a.desc.per, &n, nz.data(), &nh, &nn) == TAMCMC_OK && n <= a.desc.per)
            I.tile_rot = pick_tile_rot(tab.data(), n, a.desc.x_first, a.desc.step, tb, (int)((c->Nx + tb - 1) / tb));
    }
    return TAMCMC_OK;
}

int DevSampler::upload_proposal(int m, const double *L_rowmajor, const double *cov, const double *mu, double sigma) {
    Impl &I = *impl;
    I.armed_it = -1;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t Nv = (size_t)a.Nv;
    std::vector<double> LT(Nv * Nv);
    for (size_t i = 0; i < Nv; i++)
        for (size_t k = 0; k < Nv; k++) LT[k * Nv + i] = (k <= i) ? L_rowmajor[i * Nv + k] : 0.0;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    DCHK(up(a.LT + (size_t)m * Nv * Nv, LT.data(), Nv * Nv, st));
    DCHK(up(a.cov + (size_t)m * Nv * Nv, cov, Nv * Nv, st));
    DCHK(up(a.mu + (size_t)m * Nv, mu, Nv, st));
    DCHK(up(a.sigma + m, &sigma, 1, st));
    DCHK(hipStreamSynchronize(st));
    return TAMCMC_OK;
}

// gathers the chains' current state into one contiguous block: [vars C Nv | params C Np | logL C | logPr C | logPost C | Pmove C |
// moved C (as double) | counters 4 (as double: exact below 2^53) | per-chain move counts C]
__global__ void __launch_bounds__(256) k_pack_state(const DevSamplerArgs a, const int P, double *out) {
    const size_t C = (size_t)a.C, Np = (size_t)a.desc.Np, Nv = (size_t)a.Nv;
    const size_t n_v = C * Nv, n_p = C * Np, total = n_v + n_p + 6 * C + 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        double v;
        if (i < n_v) v = a.vars_cur[(size_t)P * n_v + i];
        else if (i < n_v + n_p) v = a.params_cur[(size_t)P * n_p + (i - n_v)];
        else {
            const size_t r = i - n_v - n_p, k = r / C, m = r - k * C;
            if (k == 0) v = a.logL_cur[(size_t)P * C + m];
            else if (k == 1) v = a.logPr_cur[(size_t)P * C + m];
            else if (k == 2) v = a.logPost_cur[(size_t)P * C + m];
            else if (k == 3) v = a.Pmove[m];
            else if (k == 4) v = (double)a.moved[m];
            else if (r < 5 * C + 4) v = (double)a.counters[r - 5 * C];
            else v = (double)a.counters[8 + (r - 5 * C - 4)];
        }
        out[i] = v;
    }
}

// One small kernel + ONE copy into pinned memory (eight copies into pageable memory cost ~150 us per tamcmc_sampler_run call).
int DevSampler::download_state(double *vars, double *params, double *logL, double *logPr, double *logPost, double *Pmove,
                               int *moved, long *counters, long *moves_per_chain) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t C = (size_t)a.C, Np = (size_t)a.desc.Np, Nv = (size_t)a.Nv;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    const size_t n_v = C * Nv, n_p = C * Np, total = n_v + n_p + 6 * C + 4;
    if (!I.d_pack) {
        DCHK(I.dalloc(&I.d_pack, total));
        DCHK(hipHostMalloc((void **)&I.h_pack, total * sizeof(double), hipHostMallocDefault));
    }
    hipLaunchKernelGGL(k_pack_state, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, I.parity, I.d_pack);
    DCHK(hipMemcpyAsync(I.h_pack, I.d_pack, total * sizeof(double), hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    const double *h = I.h_pack;
    if (vars) std::memcpy(vars, h, n_v * 8);
    if (params) std::memcpy(params, h + n_v, n_p * 8);
    const double *sc = h + n_v + n_p;
    if (logL) std::memcpy(logL, sc, C * 8);
    if (logPr) std::memcpy(logPr, sc + C, C * 8);
    if (logPost) std::memcpy(logPost, sc + 2 * C, C * 8);
    if (Pmove) std::memcpy(Pmove, sc + 3 * C, C * 8);
    if (moved) for (size_t m = 0; m < C; m++) moved[m] = (int)sc[4 * C + m];
    if (counters) for (int k = 0; k < 4; k++) counters[k] = (long)sc[5 * C + (size_t)k];
    if (moves_per_chain) for (size_t m = 0; m < C; m++) moves_per_chain[m] = (long)sc[5 * C + 4 + m];
    return TAMCMC_OK;
}

int DevSampler::download_proposal(int m, double *cov, double *mu, double *sigma) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t Nv = (size_t)a.Nv;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    if (cov) DCHK(hipMemcpyAsync(cov, a.cov + (size_t)m * Nv * Nv, Nv * Nv * 8, hipMemcpyDeviceToHost, st));
    if (mu) DCHK(hipMemcpyAsync(mu, a.mu + (size_t)m * Nv, Nv * 8, hipMemcpyDeviceToHost, st));
    if (sigma) DCHK(hipMemcpyAsync(sigma, a.sigma + m, 8, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    return TAMCMC_OK;
}

// n_iter iterations starting at iteration counter `it0`; learn[i] != 0 -> adaptation after iteration it0+i.
// Stretches without adaptation run as fused steps (A), the others in lockstep (B); see the head of this file.
static std::atomic<int> g_running_calls{0};
struct RunningCall {
    RunningCall() { g_running_calls.fetch_add(1, std::memory_order_relaxed); }
    ~RunningCall() { g_running_calls.fetch_sub(1, std::memory_order_relaxed); }
};

// TAMCMC_CALL_TIMELINE=1: host-side stamps of every run() call on stderr (entry -> first launch -> all launches enqueued -> streams idle)
static const bool g_call_timeline = getenv("TAMCMC_CALL_TIMELINE") != nullptr;
struct CallTimeline {
    std::chrono::steady_clock::time_point t[5];
    int n = 0;
    void mark() { if (g_call_timeline && n < 5) t[n++] = std::chrono::steady_clock::now(); }
    ~CallTimeline() {
        if (!g_call_timeline || n < 2) return;
        fprintf(stderr, "run() timeline (us):");
        for (int k = 1; k < n; k++) fprintf(stderr, " %.1f", std::chrono::duration<double, std::micro>(t[k] - t[k - 1]).count());
        fprintf(stderr, "\n");
    }
};

int DevSampler::run(long it0, long n_iter, const char *learn, double *samples, double *stats) {
    RunningCall running_call;
    CallTimeline tl;
    tl.mark();
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    if (n_iter <= 0) return TAMCMC_OK;
    if (I.use_drift) return run_mala(it0, n_iter, learn, samples, stats);
    DCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t C = (size_t)a.C, Nv = (size_t)a.Nv;
    const int tb = tile_bins(c->wgs, c->K);
    a.ntiles = (a.desc.Nx + tb - 1) / tb;
    DCHK(c->d_part.reserve(C * (size_t)a.ntiles * 2));
    a.partials = c->d_part.p;
    a.tile_bins = tb;
    a.bg = nullptr;
    const bool use_fused = I.fused_ok && c->step_scheme != 1 && c->wgs == 64 && (c->K == 4 || c->K == 8 || c->K == 16);
    const size_t NS = (size_t)I.f.NS;
    I.f.bg = nullptr;
    if (c->precision == TAMCMC_PRECISION_FAST) {
        // background series per (slot, tile).  (B): C slots in the context's scratch (rewritten every iteration).  (A): 2 x NS slots of
        // the sampler's OWN -- the candidates prepared by the last launch of a call are carried over to the next call, and anything else
        // that runs on the context in between (another sampler, a batched evaluation) rewrites the context's scratch
        DCHK(c->d_bg.reserve(C * (size_t)a.ntiles * 8));
        a.bg = c->d_bg.p;
        if (use_fused) {
            DCHK(I.fused_bg.reserve(3 * NS * (size_t)a.ntiles * 8));
            I.f.bg = I.fused_bg.p;
        }
    }
    if (use_fused) {  // (A): the tiles' partial sums by iteration parity (launch i writes one half and reads the other)
        DCHK(I.fused_part.reserve(2 * C * (size_t)a.ntiles * 2));
        I.f.part = I.fused_part.p;
    }
    // record buffers: at least 256 iterations' worth and grown geometrically, so that a caller that records in buffers of a fixed
    // length (the reference's Nbuffer) or a short call after a shorter one never pays an allocation -- nor, with it, new kernel
    // arguments -- in its steady state (older, smaller buffers are released with the sampler)
    auto grown = [](size_t need, size_t have, size_t unit) { const size_t floor_ = 256 * unit; return std::max(std::max(need, floor_), have * 2); };
    double *zc_smp = I.device_view(samples, (size_t)n_iter * C * Nv * 8), *zc_st = I.device_view(stats, (size_t)n_iter * C * 3 * 8);
    if (samples && !zc_smp && I.smp_cap < (size_t)n_iter * C * Nv) {
        const size_t cap = grown((size_t)n_iter * C * Nv, I.smp_cap, C * Nv);
        DCHK(I.dalloc(&a.samples, cap));
        I.smp_cap = cap;
    }
    if (stats && !zc_st && I.stat_cap < (size_t)n_iter * C * 3) {
        const size_t cap = grown((size_t)n_iter * C * 3, I.stat_cap, C * 3);
        DCHK(I.dalloc(&a.stats, cap));
        I.stat_cap = cap;
    }
    DevSamplerArgs args = a;
    args.samples = samples ? (zc_smp ? zc_smp : a.samples) : nullptr;
    args.stats = stats ? (zc_st ? zc_st : a.stats) : nullptr;
    // chain groups [goff[g], goff[g+1]) of the lockstep scheme
    const int G = I.G;
    int goff[5];
    for (int g = 0; g <= G; g++) goff[g] = (int)(((long)a.C * g) / G);
    auto group_of = [&](int chain) { int g = 0; while (g + 1 < G && chain >= goff[g + 1]) g++; return g; };
    LoglikeArgs la[4], lf[3];
    auto fill_common = [&](LoglikeArgs &l, int B) {
        l.x = c->dx.p; l.y = c->dy.p; l.logx = c->dlogx.p; l.Nx = a.desc.Nx; l.B = B; l.ntiles = a.ntiles;
        l.x0 = a.desc.x_first; l.step = a.desc.step; l.noise_stride = a.desc.stride; l.model = nullptr; l.tile_rot = I.tile_rot;
    };
    for (int g = 0; g < G; g++) {
        LoglikeArgs &l = la[g];
        const int first = goff[g];
        fill_common(l, goff[g + 1] - goff[g]);
        l.mults = a.mults; l.offsets = a.pairs + 2 * first; l.noise = a.noise + (size_t)first * a.desc.stride;
        l.per = a.desc.per; l.slot0 = first;  // (chain m's table is rows [m per, (m+1) per) of a.mults: dev_unpack.h, rgb_prestep.hip)
        l.nharvey = a.nh + first; l.nnoise = a.nn + first; l.partials = a.partials + (size_t)first * a.ntiles * 2;
        l.bg_poly = a.bg ? a.bg + (size_t)first * a.ntiles * 8 : nullptr;
    }
    for (int e = 0; e < 3; e++) {  // (A): evaluation m = chain m, its table in a slot of candidate set e = iteration mod 3 (decide())
        LoglikeArgs &l = lf[e];
        const FusedArgs &f = I.f;
        fill_common(l, a.C);
        l.mults = f.mults + (size_t)e * NS * a.desc.per; l.offsets = f.pairs + (size_t)e * 2 * NS; l.noise = f.noise + (size_t)e * NS * a.desc.stride;
        l.nharvey = f.nh + (size_t)e * NS; l.nnoise = f.nn + (size_t)e * NS; l.partials = f.part;
        l.bg_poly = f.bg ? f.bg + (size_t)e * NS * a.ntiles * 8 : nullptr;
        l.per = a.desc.per; l.slot0 = 0;
    }

    tl.mark();  // (set-up of the call done: buffers, argument blocks)
    int used_ev = 0;
    std::vector<std::pair<int, long>> fused_ev;  // (event pair, launches it brackets) of the fused stretches of this call
    // fused step with two chain groups: the launches of an iteration overlap, so the stretch's elapsed time is not a launch duration;
    // sampled launches of the second group are bracketed on their own stream instead
    bool s1_open = false;      // the second group's stream still holds launches the context stream has not waited for
    int g_used = 0;            // gev pairs used in this call
    long g_launches = 0, g_iters = 0;  // launches / iterations of the split stretches of this call
    int P = I.parity;
    double kernel_ms = 0;
    long n_launch = 0, n_eval = 0;
    auto drain_events = [&](double launches_represented, long evals) -> int {  // call after a stream sync
        if (used_ev) {
            double tot = 0;
            for (int e = 0; e < used_ev; e++) {
                float ms = 0;
                DCHK(hipEventElapsedTime(&ms, I.ev[e][0], I.ev[e][1]));
                tot += ms;
            }
            kernel_ms += tot / used_ev * launches_represented;
            n_launch += (long)launches_represented;
            n_eval += evals;
        }
        used_ev = 0;
        return TAMCMC_OK;
    };

    // ---- (B) one iteration per round over [ia, ib): k_iterate settles iteration it-1 and proposes iteration it
    auto lockstep = [&](long ia, long ib) -> int {
        I.armed_it = -1;
        I.it_lockstep += ib - ia;
        // the extra streams start after everything already enqueued on the context stream
        if (G > 1) {
            DCHK(hipEventRecord(I.ev_fork, st));
            for (int g = 1; g < G; g++) DCHK(hipStreamWaitEvent(I.gst[g], I.ev_fork, 0));
        }
        const long len = ib - ia;
        const long ev_every = len > 32 ? len / 32 : 1;
        int pending = 0, have_pre = 0;
        for (long i = ia; i <= ib; i++) {
            const long it = it0 + i;
            const int learn_p = (pending && learn && learn[i - 1]) ? 1 : 0;
            // L z of iteration it+1 can be computed by spare workgroups of THIS launch when no adaptation rewrites L in this
            // launch (learn_p) nor in the next one before its proposal (learn[i])
            const int make_pre = (I.pre_lz && i + 1 < ib && !learn_p && !(learn && learn[i])) ? 1 : 0;
            const int pre_flags = (have_pre ? 1 : 0) | (make_pre ? 2 : 0);
            const size_t lds = I.lds_base + ((learn_p && a.chol_in_lds) ? I.lds_adapt : 0);
            const long rec = (pending && (samples || stats)) ? i - 1 : (long)-1;
            // does settling iteration it-1 swap a pair that straddles two groups? (same draw as the kernel: Philox is host/device)
            int gA = -1, gB = -1;
            if (pending && G > 1) {
                const long itp = it - 1;
                if (a.dN_mixing > 0 && (itp % a.dN_mixing == 0) && itp != 0 && a.C > 1) {
                    double u, u2;
                    rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)itp, 0, u, u2);
                    int A = (int)(u2 * (double)(a.C - 1));
                    if (A > a.C - 2) A = a.C - 2;
                    if (group_of(A) != group_of(A + 1)) { gA = group_of(A); gB = group_of(A + 1); }
                }
            }
            if (gA >= 0) {  // each of the two groups needs the other's k_loglike(it-1) before it settles the pair
                DCHK(hipEventRecord(I.ev_kb[gA], I.gst[gA]));
                DCHK(hipEventRecord(I.ev_kb[gB], I.gst[gB]));
                DCHK(hipStreamWaitEvent(I.gst[gA], I.ev_kb[gB], 0));
                DCHK(hipStreamWaitEvent(I.gst[gB], I.ev_kb[gA], 0));
            }
            for (int g = 0; g < G; g++) {
                const int cnt = goff[g + 1] - goff[g];
                if (i < ib)
                    hipLaunchKernelGGL(k_iterate<true>, dim3(make_pre ? 2 * cnt : cnt), dim3(TB), lds, I.gst[g], args, it, P, pending, rec, learn_p,
                                       I.adapt_scratch, goff[g], cnt, pre_flags, I.rgb ? rgb_device_slice(c, I.rgb_bmax, g) : rgb::Slice());
                else  // settle the last iteration of this stretch (MH test, swap, record, adaptation); nothing is proposed
                    hipLaunchKernelGGL(k_iterate<false>, dim3(cnt), dim3(TB), lds, I.gst[g], args, it, P, pending, rec, learn_p, I.adapt_scratch,
                                       goff[g], cnt, 0, rgb::Slice());
            }
            have_pre = make_pre;
            if (gA >= 0) {  // ... and must not overwrite (next iteration) what the other group's settle is still reading
                DCHK(hipEventRecord(I.ev_ki[gA], I.gst[gA]));
                DCHK(hipEventRecord(I.ev_ki[gB], I.gst[gB]));
                DCHK(hipStreamWaitEvent(I.gst[gA], I.ev_ki[gB], 0));
                DCHK(hipStreamWaitEvent(I.gst[gB], I.ev_ki[gA], 0));
            }
            P ^= 1;
            pending = 1;
            if (i < ib && I.rgb) {  // the proposals' tables: solver, then sort / zeta / rows (and the FAST background series)
                RgbDeviceTables T;
                T.mults = a.mults; T.pairs = a.pairs; T.nh = a.nh; T.nn = a.nn; T.noise = a.noise; T.stride = a.desc.stride;
                T.status = a.status_prop + (size_t)P * C;
                T.bg = a.bg; T.ntiles = a.ntiles; T.tile_bins = a.tile_bins;
                for (int g = 0; g < G; g++) {
                    int rc = rgb_device_stage(c, goff[g], goff[g + 1] - goff[g], I.rgb_bmax, g, a.desc.per, T, I.gst[g]);
                    if (rc) return rc;
                }
            }
            if (i < ib) {
                for (int g = 0; g < G; g++) {
                    const bool timed = g == 0 && c->timing && ((i - ia) % ev_every == 0) && used_ev < I.n_ev - 16;  // (the top 16 pairs: fused stretches)
                    if (timed) DCHK(hipEventRecord(I.ev[used_ev][0], I.gst[g]));
                    DCHK(launch_loglike(la[g], c->precision, c->wgs, c->K, false, I.gst[g]));
                    if (timed) { DCHK(hipEventRecord(I.ev[used_ev][1], I.gst[g])); used_ev++; }
                }
            }
        }
        // join: the context stream continues after every group
        for (int g = 1; g < G; g++) {
            DCHK(hipEventRecord(I.ev_join[g], I.gst[g]));
            DCHK(hipStreamWaitEvent(st, I.ev_join[g], 0));
        }
        if (c->timing) {  // (with chain groups every launch carries C/G evaluations and overlaps the other groups' kernels)
            DCHK(hipStreamSynchronize(st));
            return drain_events((double)len * G, len * (long)a.C);
        }
        return TAMCMC_OK;
    };

    // ---- (A) fused steps over [ia, ib) (no adaptation inside): one launch per iteration on the context stream, or two (one per chain group)
    auto fused = [&](long ia, long ib) -> int {
        const FusedArgs &f = I.f;
        const int nbr = 8 * f.NS;                                     // candidate roles, a multiple of 8 (keeps the tiles' XCD mapping)
        const int nlz2 = ((2 * a.C + 7) / 8) * 8;
        const int ntiles_pad = ((a.ntiles + 7) / 8) * 8;
        const long len = ib - ia;
        I.it_fused += len;
        int q = P;
        StepCtl sc{};
        {  // device-memory image of the two argument blocks (re-uploaded only when a pointer or size changed since the last run)
            const size_t n1 = (sizeof(DevSamplerArgs) + 15) & ~(size_t)15, n2 = sizeof(FusedArgs);
            std::vector<unsigned char> img(n1 + n2, 0);
            std::memcpy(img.data(), &args, sizeof(DevSamplerArgs));
            std::memcpy(img.data() + n1, &f, sizeof(FusedArgs));
            if (!I.d_argcopy) DCHK(I.dalloc(&I.d_argcopy, n1 + n2));
            if (img != I.h_argcopy) {
                DCHK(hipMemcpyAsync(I.d_argcopy, img.data(), n1 + n2, hipMemcpyHostToDevice, st));
                DCHK(hipStreamSynchronize(st));  // (img is a stack object)
                I.h_argcopy = img;
                I.armed_it = -1;                 // (the carried-over candidates were built for the old buffers)
            }
            sc.ga = (const DevSamplerArgs *)I.d_argcopy;
            sc.gf = (const FusedArgs *)(I.d_argcopy + n1);
        }
        sc.first = 0; sc.cnt = a.C; sc.extra = 1;
        // does the context stream hold work of this call that the second group's stream has to wait for?  (Every entry point of the
        // library returns with its streams idle, so a call that starts on carried-over candidates has nothing to wait for: the event
        // hop would only delay the second group's first launch by 10-30 us.)
        bool st_has_work = ia > 0;
        if (!(I.armed_it == it0 + ia && I.armed_q == q)) {
            st_has_work = true;
            // entry: L z of the first two iterations, then the candidates of iteration ia built on the settled chains (state of parity q)
            sc.it = it0 + ia; sc.rec = -1; sc.q = q; sc.flags = ST_LZ; sc.nbr = 0; sc.nlz = nlz2; sc.n_lz_live = 2 * a.C; sc.it_lz = it0 + ia; sc.q_lz = q;
            DCHK(launch_step(c->precision, c->K, nlz2, st, args, f, lf[0], sc));
            sc.flags = ST_ENTRY; sc.nbr = nbr; sc.nlz = 0; sc.n_lz_live = 0;
            DCHK(launch_step(c->precision, c->K, nbr, st, args, f, lf[0], sc));
        }
        // the likelihood kernel's time for the roofline: two events around the whole stretch, i.e. the average includes the time between
        // two launches
        const bool timed = c->timing && fused_ev.size() < 16;
        const int fe = I.n_ev - 1 - (int)fused_ev.size();
        // (it pays once one launch no longer fits the GPU's resident waves -- 20 chains x 196 tiles: 27.7 -> 23.9 us, x 782 tiles: 59.8 ->
        // 49.6 us -- and costs below that: 8 chains x 196 tiles 20.5 -> 23.7 us, 20 chains x 20 tiles 33.5 -> 35.6 us; tools/groups_probe.py)
        const bool split_ok = f.xsplit < a.C && c->step_scheme != 2 && (c->step_scheme == 3 || (long)a.C * a.ntiles >= 2500);
        const bool split = split_ok;
        if (timed && !split) DCHK(hipEventRecord(I.ev[fe][0], st));
        // Two chain groups, each with its own launch per iteration on its own stream: a launch is a chain of dependent steps (sums ->
        // decision -> table rows -> tile, ~18 us even for five chains) that leaves most of the GPU idle at its two ends; the two
        // groups' launches fill each other's ends.  Nothing is shared between the groups' launches except at a swap whose pair straddles
        // the groups: see straddles() below.  (Same chains bit for bit: the launches' contents are the same.)
        const int first1 = f.xsplit;
        hipStream_t s1 = I.gst[1];
        long n_split = 0, next_sample = len >= 97 ? 48 : len / 2;
        bool s1_ahead = false, s1_must_wait = st_has_work;  // s1 holds launches st has not waited for / s1 has not seen st's latest launches
        auto swap_pair_of = [&](long it) -> int {
            if (!(a.C >= 2 && a.dN_mixing > 0 && (it % a.dN_mixing == 0) && it != 0)) return -1;
            double u, u2;
            rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)it, 0, u, u2);
            int A = (int)(u2 * (double)(a.C - 1));
            if (A > a.C - 2) A = a.C - 2;
            return A;
        };
        // Launch i of a stretch: the tiles of iteration i (each decides iteration i-1 for its chain first), the chains' commit workgroups
        // (state, record and counters of iteration i-1), the candidates of iteration i+1, the L z of iteration i+2.  settled: the chains
        // are settled (first launch of a stretch: nothing to decide or commit).
        auto launch_group = [&](int first, int cnt, int A, long i, bool settled, hipStream_t stream, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) -> int {
            const bool owns_pair = A >= first && A + 1 < first + cnt;
            sc.it = it0 + i; sc.rec = ((samples || stats) && !settled) ? i - 1 : (long)-1; sc.q = q;
            sc.flags = ST_L | ST_BR | ST_LZ | ST_COMMIT | (settled ? ST_FIRST : 0);
#ifdef TAMCMC_PROBE  // timing experiments only (results are wrong): leave kinds of workgroups out of the launch
            if (const char *ep = getenv("TAMCMC_PROBE_STEP")) {
                const int pm = atoi(ep);
                if (pm & 1) sc.flags &= ~ST_BR;
                if (pm & 2) sc.flags &= ~ST_COMMIT;
                if (pm & 4) sc.flags &= ~ST_LZ;
                if (pm & 8) sc.flags |= ST_FIRST;
                if (pm & 16) sc.flags &= ~ST_L;
            }
#endif
            sc.first = first; sc.cnt = cnt; sc.extra = owns_pair ? 1 : 0; sc.pairA = settled ? -1 : swap_pair_of(it0 + i - 1);
            sc.nbr = 8 * (2 * cnt + (owns_pair ? 4 : 0)); sc.nlz = ((2 * cnt + 7) / 8) * 8; sc.n_lz_live = cnt; sc.it_lz = it0 + i + 2; sc.q_lz = q;
            LoglikeArgs lq = lf[(it0 + i) % 3];
            lq.B = cnt;
            lq.partials = f.part + ((size_t)q * a.C + first) * a.ntiles * 2;
            DCHK(launch_step(c->precision, c->K, sc.nbr + sc.nlz + ntiles_pad * cnt, stream, args, f, lq, sc, e0, e1));
            return TAMCMC_OK;
        };
        // After the last iteration of a stretch: the commit workgroups alone (iteration ib-1 decided, the chains settled in parity q).
        auto launch_close = [&](int first, int cnt, hipStream_t stream) -> int {
            sc.it = it0 + ib; sc.rec = (samples || stats) ? ib - 1 : (long)-1; sc.q = q;
            sc.flags = ST_COMMIT;
            sc.first = first; sc.cnt = cnt; sc.extra = 0;
            sc.nbr = 0; sc.nlz = ((cnt + 7) / 8) * 8; sc.n_lz_live = 0; sc.it_lz = 0; sc.q_lz = 0;
            DCHK(launch_step(c->precision, c->K, sc.nlz, stream, args, f, lf[0], sc));
            return TAMCMC_OK;
        };
        // A swap whose pair straddles the two groups: the launch of that iteration builds the pair's cross candidates on both chains'
        // vectors, the next one decides the swap from both chains' sums and commits each side from the other's vectors -- those two
        // launches are joint (all chains, on the context stream, after both groups' earlier launches)
        auto straddles = [&](int A) { return split_ok && A == first1 - 1; };
        for (long i = ia; i < ib; i++) {
            const int A = swap_pair_of(it0 + i);
            const bool settled = i == ia;
            const bool joint = !split_ok || straddles(A) || (!settled && straddles(swap_pair_of(it0 + i - 1)));
            if (!joint) {
                if (s1_must_wait) {
                    DCHK(hipEventRecord(I.ev_fork, st));
                    DCHK(hipStreamWaitEvent(s1, I.ev_fork, 0));
                    s1_must_wait = false;
                }
                int rc = launch_group(0, first1, A, i, settled, st);
                if (rc) return rc;
                // (sampled launches: every 97th iteration, or the middle one of a short stretch -- the first two-group iteration at or
                // after it: a joint iteration there must not leave a short call without a measured launch)
                const bool sample = timed && g_used < I.n_gev && (i - ia) >= next_sample;
                rc = sample ? launch_group(first1, a.C - first1, A, i, settled, s1, I.gev[g_used][0], I.gev[g_used][1])
                            : launch_group(first1, a.C - first1, A, i, settled, s1);
                if (rc) return rc;
                if (sample) { g_used++; next_sample += 97; }
                s1_ahead = true;
                n_split++;
            } else {
                if (s1_ahead) {
                    DCHK(hipEventRecord(I.ev_join[1], s1));
                    DCHK(hipStreamWaitEvent(st, I.ev_join[1], 0));
                    s1_ahead = false;
                }
                int rc = launch_group(0, a.C, A, i, settled, st);
                if (rc) return rc;
                s1_must_wait = true;
            }
            q ^= 1;
        }
        if (s1_ahead) {  // (the last launches were one per group: iteration ib-1's swap pair lies inside one of them)
            int rc = launch_close(0, first1, st);
            if (rc) return rc;
            rc = launch_close(first1, a.C - first1, s1);
            if (rc) return rc;
        } else {
            int rc = launch_close(0, a.C, st);
            if (rc) return rc;
        }
        if (s1_ahead) {
            if (ib >= n_iter) s1_open = true;  // the call's last stretch: the host waits for both streams below (no event hop on the GPU)
            else {
                DCHK(hipEventRecord(I.ev_join[1], s1));
                DCHK(hipStreamWaitEvent(st, I.ev_join[1], 0));
            }
        }
        if (timed) {  // (read after the call's final synchronisation)
            if (!split) {
                DCHK(hipEventRecord(I.ev[fe][1], st));
                fused_ev.push_back({fe, len});
            } else { g_launches += 2 * n_split + (len - n_split); g_iters += len; }
        }
        P = q;
        I.armed_it = it0 + ib;
        I.armed_q = q;
        return TAMCMC_OK;
    };

    // ---- split [0, n_iter) into stretches: quiet ones (no adaptation, at least MIN_FUSED long) run fused
    const long MIN_FUSED = 3;  // a stretch pays one entry launch
    auto quiet_end = [&](long from) { long q2 = from; while (q2 < n_iter && !(learn && learn[q2])) q2++; return q2; };
    long i = 0;
    while (i < n_iter) {
        const long jn = quiet_end(i);
        if (use_fused && jn - i >= MIN_FUSED) {
            int rc = fused(i, jn);
            if (rc) return rc;
            i = jn;
            continue;
        }
        long k = i;  // lockstep up to the start of the next long quiet stretch
        for (;;) {
            const long q2 = quiet_end(k);
            if (use_fused && q2 - k >= MIN_FUSED && k > i) break;
            k = q2;
            while (k < n_iter && learn && learn[k]) k++;
            if (k >= n_iter) break;
        }
        int rc = lockstep(i, k);
        if (rc) return rc;
        i = k;
    }
    I.parity = P;
    DCHK(hipGetLastError());
    tl.mark();  // (every launch enqueued)
    if (s1_open && ((samples && !zc_smp) || (stats && !zc_st))) {  // (the copies below read what the second group's launches write)
        DCHK(hipEventRecord(I.ev_join[1], I.gst[1]));
        DCHK(hipStreamWaitEvent(st, I.ev_join[1], 0));
        s1_open = false;
    }
    if (samples && !zc_smp) DCHK(hipMemcpyAsync(samples, a.samples, (size_t)n_iter * C * Nv * 8, hipMemcpyDeviceToHost, st));
    if (stats && !zc_st) DCHK(hipMemcpyAsync(stats, a.stats, (size_t)n_iter * C * 3 * 8, hipMemcpyDeviceToHost, st));
    const bool poll = g_running_calls.load(std::memory_order_relaxed) == 1;
    if (s1_open) DCHK(Impl::wait_stream(I.gst[1], poll));
    DCHK(Impl::wait_stream(st, poll));
    tl.mark();  // (streams idle)
    for (const auto &e : fused_ev) {
        float ms = 0;
        DCHK(hipEventElapsedTime(&ms, I.ev[e.first][0], I.ev[e.first][1]));
        kernel_ms += ms;
        n_launch += e.second;
        n_eval += e.second * (long)a.C;
    }
    if (g_launches > 0) {  // split stretches: (average duration of the sampled launches) x (launches); one launch = one group's chains
        double tot = 0;
        for (int e = 0; e < g_used; e++) {
            float ms = 0;
            DCHK(hipEventElapsedTime(&ms, I.gev[e][0], I.gev[e][1]));
            tot += ms;
        }
        if (g_used > 0) {
            kernel_ms += tot / g_used * (double)g_launches;
            n_launch += g_launches;
            n_eval += g_iters * (long)a.C;
        }
    }
    c->kernel_ms += kernel_ms;
    c->launches += n_launch;
    c->evals += n_eval;
    return TAMCMC_OK;
}

// The Langevin engine (use_drift): per iteration k_mala_settle (settle it-1, propose it) -> the finite-difference batch of the proposals
// (k_fd_unpack, base k_loglike with model rows, k_loglike<DELTA>, k_finalize x2) -> k_mala_test.  See dev_mala_impl.h.
int DevSampler::run_mala(long it0, long n_iter, const char *learn, double *samples, double *stats) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    DCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    I.armed_it = -1;
    const size_t C = (size_t)a.C, Nv = (size_t)a.Nv, Np = (size_t)a.desc.Np;
    {   // the batch's layout follows the context's options (arithmetic mode, geometry, windowed differences): re-laid out when they change
        FdBatch nb;
        int rc = nb.layout(c, I.model_id, I.prior_class, a.C, (int64_t)Np, I.h_plength.data(), a.Nv);
        if (rc) return rc;
        const bool need_bg = c->precision == TAMCMC_PRECISION_FAST && !I.fd_bg.p;  // (a switch to FAST between two calls keeps every size)
        if (nb.total_bytes != I.fd.total_bytes || nb.windowed != I.fd.windowed || nb.ntiles != I.fd.ntiles || !I.fd_block.p || need_bg) {
            I.fd = nb;
            rc = fd_ensure_poly(c);
            if (rc) return rc;
            DCHK(I.fd_block.reserve(nb.total_bytes));
            std::vector<unsigned char> hb(nb.in_bytes, 0);
            std::memcpy(hb.data() + nb.o_pr, I.h_priors.data(), 4 * Np * 8);
            std::memcpy(hb.data() + nb.o_ex, I.h_extra.data(), 10 * 8);
            std::memcpy(hb.data() + nb.o_sw, I.h_sw.data(), Np * 4);
            std::memcpy(hb.data() + nb.o_pl, I.h_plength.data(), 11 * 4);
            std::memcpy(hb.data() + nb.o_idx, I.h_idx.data(), Nv * 4);
            DCHK(hipMemcpyAsync(I.fd_block.p, hb.data(), nb.in_bytes, hipMemcpyHostToDevice, st));
            DCHK(hipStreamSynchronize(st));
            DCHK(I.fd_part.reserve(nb.nS * (size_t)nb.ntiles * 2));
            DCHK(I.fd_S.reserve(nb.nS));
            if (nb.windowed) DCHK(I.fd_model.reserve(3 * C * (size_t)c->Nx + 2 * C * (size_t)nb.ntiles * FD_MOM + ((size_t)nb.B * nb.ntiles + 7) / 8));  // three planes (1/M0, y/M0, M0 of the base points) + tile moments (two layouts) + done flags
            if (c->precision == TAMCMC_PRECISION_FAST) DCHK(I.fd_bg.reserve((size_t)(nb.windowed ? a.C : nb.B) * nb.ntiles * 8));
            I.grad_valid = false;
        }
    }
    FdBatch &fd = I.fd;
    unsigned char *db = I.fd_block.p;
    MalaArgs M = I.mala;
    M.S = I.fd_S.p; M.lpp = (const double *)(db + fd.o_lpp); M.lpm = (const double *)(db + fd.o_lpm); M.st = (const int *)(db + fd.o_st);
    M.h = (double *)(db + fd.o_h); M.E = fd.E; M.windowed = fd.windowed ? 1 : 0; M.fd_step_rel = I.fd_step_rel; M.delta = I.delta;
    if (samples && I.smp_cap < (size_t)n_iter * C * Nv) {
        DCHK(I.dalloc(&a.samples, (size_t)n_iter * C * Nv));
        I.smp_cap = (size_t)n_iter * C * Nv;
    }
    if (stats && I.stat_cap < (size_t)n_iter * C * 3) {
        DCHK(I.dalloc(&a.stats, (size_t)n_iter * C * 3));
        I.stat_cap = (size_t)n_iter * C * 3;
    }
    DevSamplerArgs args = a;
    if (!samples) args.samples = nullptr;
    if (!stats) args.stats = nullptr;
    const size_t lds_settle = (4 * Nv + Np + 1 + 8) * sizeof(double) + 32;
    const size_t lds_test0 = (5 * Nv + 8) * sizeof(double) + 32, lds_adapt = (Nv * Nv + Nv) * sizeof(double);
    const bool chol_lds = lds_test0 + lds_adapt <= 156 * 1024;
    args.chol_in_lds = chol_lds ? 1 : 0;
    I.mala_chol_lds = args.chol_in_lds;
    I.it_lockstep += n_iter;
    if (!chol_lds && !I.adapt_scratch) DCHK(I.dalloc(&I.adapt_scratch, C * (Nv * Nv + Nv)));
    if (lds_test0 + lds_adapt > 64 * 1024 && chol_lds)
        DCHK(hipFuncSetAttribute((const void *)k_mala_test, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_test0 + lds_adapt)));
    int P = I.parity;
    auto batch = [&](const double *d_params, bool timed) -> int {
        return fd.enqueue(c, db, d_params, I.fd_part.p, I.fd_S.p, I.fd_model.p, I.fd_bg.p, timed ? I.ev[0][0] : nullptr, timed ? I.ev[0][1] : nullptr);
    };
    if (!I.grad_valid) {  // gradient at the chains' current positions (start of a run, new positions from outside)
        hipLaunchKernelGGL(k_mala_steps, dim3(1), dim3(256), 0, st, args, M);
        int rc = batch(a.params_cur + (size_t)P * C * Np, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_mala_ginit, dim3(a.C), dim3(TB), 2 * Nv * sizeof(double), st, args, M, P);
        DCHK(hipGetLastError());
        I.grad_valid = true;
    }
    double kernel_ms = 0;
    long n_timed = 0, fd_bins_sampled = 0;
    int pending = 0;
    for (long i = 0; i <= n_iter; i++) {
        const long it = it0 + i;
        const long rec = (pending && (samples || stats)) ? i - 1 : (long)-1;
        if (i < n_iter) hipLaunchKernelGGL(k_mala_settle<true>, dim3(a.C), dim3(TB), lds_settle, st, args, M, it, P, pending, rec);
        else hipLaunchKernelGGL(k_mala_settle<false>, dim3(a.C), dim3(TB), lds_settle, st, args, M, it, P, pending, rec);
        P ^= 1;
        pending = 1;
        if (i == n_iter) break;
        const bool timed = c->timing && (i == 0 || i == n_iter / 2);  // two sampled batches per call (an event read needs a synchronisation)
        int rc = batch(a.params_prop, timed);
        if (rc) return rc;
        const int learn_i = (learn && learn[i]) ? 1 : 0;
        // (the adaptation's work area is reserved in every step when it fits: without adaptation the triangular solves keep the factor there)
        hipLaunchKernelGGL(k_mala_test, dim3(a.C), dim3(TB), lds_test0 + (chol_lds ? lds_adapt : 0), st, args, M, it, P, learn_i,
                           I.adapt_scratch);
        if (timed) {
            DCHK(hipStreamSynchronize(st));
            float ms = 0;
            DCHK(hipEventElapsedTime(&ms, I.ev[0][0], I.ev[0][1]));
            kernel_ms += ms;
            n_timed++;
            if (fd.windowed) {  // what the delta launch really touched (roofline bookkeeping, as fd_run does)
                std::vector<int> rg((size_t)2 * fd.B);
                DCHK(hipMemcpy(rg.data(), db + fd.o_drange, rg.size() * sizeof(int), hipMemcpyDeviceToHost));
                long bins = 0;
                for (int q2 = 0; q2 < fd.B; q2++) bins += rg[2 * (size_t)q2 + 1] - rg[2 * (size_t)q2];
                fd_bins_sampled += bins - fd.bins_not_walked();
            }
        }
    }
    I.parity = P;
    DCHK(hipGetLastError());
    if (samples) DCHK(hipMemcpyAsync(samples, a.samples, (size_t)n_iter * C * Nv * 8, hipMemcpyDeviceToHost, st));
    if (stats) DCHK(hipMemcpyAsync(stats, a.stats, (size_t)n_iter * C * 3 * 8, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    if (n_timed) {  // (the sampled batches stand for all of them)
        c->kernel_ms += kernel_ms / n_timed * n_iter;
        c->launches += n_iter;
        c->evals += n_iter * (long)fd.B;
        c->fd_bins += fd_bins_sampled / n_timed * n_iter;
        c->fd_delta_evals += fd.windowed ? n_iter * (long)fd.B : 0;
    }
    return TAMCMC_OK;
}

}  // namespace tamcmc
