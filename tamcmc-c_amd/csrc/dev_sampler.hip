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

// Data exchanged between workgroups of ONE launch (the tiles' partial sums, the swap pair's outcomes) goes through device-scope
// accesses that bypass the per-XCD L2 (MI355X: eight L2s, not coherent with each other for ordinary loads/stores).  A full
// __threadfence() per tile would write back and invalidate the XCD's whole L2 -- including the resident spectrum -- 4000 times per launch.
// (global address space spelled out: the accesses must be global_load/global_store ... sc1, not flat_ -- MI355X_MICROARCH.md, cross-workgroup
// hand-offs: sc1 stores, vmcnt(0), an agent-scope atomic add; the workgroup whose add came last reads with sc1 loads)
typedef double __attribute__((address_space(1))) *gdp_t;
typedef const double __attribute__((address_space(1))) *gcdp_t;
__device__ __forceinline__ double coherent_load(const double *p) { return __hip_atomic_load((gcdp_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(double *p, double v) { __hip_atomic_store((gdp_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every earlier memory operation of this wave has completed (write-through stores have reached memory) before anything later issues
__device__ __forceinline__ void drain_memory_ops() {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// (A): the same sum by ONE wave, in k_finalize's order: 256 strided per-thread sums (four per lane here), shuffle tree per 64, the
// four in order.  Every lane returns the total.  (device-scope loads: the partials were written by other workgroups of this launch)
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
            v1[q] = t < ntiles ? coherent_load(base + 2 * t) : 0.0;
            v2[q] = t < ntiles ? coherent_load(base + 2 * t + 1) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TB / 64; q++)
            if (t0 + q * 64 + lane < ntiles) { s1[q] = s1[q] + v1[q]; s2[q] = s2[q] + v2[q]; }
    }
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
                           // (chain >= xsplit), a pair (A, A+1) uses pair counter (A >= xsplit): the two groups' launches run on
                           // different streams, possibly several iterations apart, and must never write what the other one still reads
    // candidates of the iteration with parity q: [2][NS]...
    double *cand_vars, *cand_params, *cand_logPr;
    int *cand_stP, *cand_stR;          // status of the prior role / the rows role
    tamcmc_multiplet *mults;           // [2][NS][per]
    int *pairs, *nh, *nn;              // [2][2 NS], [2][NS], [2][NS]
    double *noise;                     // [2][NS][stride]
    double *bg;                        // [2][NS][ntiles][8] or nullptr
    int *slot;                         // [2][C]   table slot of chain m's proposal at the iteration of that parity
    double *lz;                        // [2][C][Nv] L z of chain m for the iteration of that parity, computed one launch ahead
    unsigned *ticket;                  // [2][C][TK] two levels: [0] counts the chain's tile GROUPS that are complete, [1 + g] the tiles
                                       //          of group g = tile mod NG that have delivered their partial sums; one 128-byte line
                                       //          each (device-scope atomics on one line serialise in the memory-side atomic unit)
    unsigned *pair_ticket;             // [2][2]   (parity, extra block) chains of the swap pair that have done their MH test
    double *acc;                       // [2][C][5] MH outcome of chain m (acc, r, logL, logPr, logPost), read by the partner that resolves the swap
};

constexpr int ST_L = 1, ST_BR = 2, ST_ENTRY = 4, ST_LZ = 8;
constexpr int NG = 8;                  // tile groups per chain (ticket level 1)
constexpr int TKS = 32;                // unsigned per ticket line
constexpr int TK = (1 + NG) * TKS;     // unsigned per chain

// The settle functions are real calls (register budget) and get the argument blocks as pointers to their device-memory image.  That image
// is written by the host only, and the pointer is the same in every lane: read through a wave-uniform pointer into constant memory, a
// field costs a scalar load (SGPR, scalar cache) instead of a flat vector load per lane, and the pointers found there are known to be
// global (global_load / global_store instead of flat_).
typedef DevSamplerArgs __attribute__((address_space(4))) ConstArgs;
typedef FusedArgs __attribute__((address_space(4))) ConstFused;
__device__ __forceinline__ const void __attribute__((address_space(4))) *uniform_ptr(const void *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (const void __attribute__((address_space(4))) *)(((unsigned long long)hi << 32) | lo);
}

// The scalar part of a chain's settled state (ONE lane): what it holds, the slot of its next proposal, the record of its statistics.
template <class AT, class FT>
__device__ __forceinline__ void fused_scalars(const AT &a, const FT &f, int m, int src_acc, double src_r, const AcceptOut &o,
                                              int next_slot, long it, int q, long rec) {
    const int C = a.C, q1 = q ^ 1;
    a.logL_cur[q1 * C + m] = o.logL;
    a.logPr_cur[q1 * C + m] = o.logPr;
    a.logPost_cur[q1 * C + m] = o.logPost;
    f.slot[q1 * C + m] = next_slot;
    a.moved[m] = src_acc;     // a swap exchanges the pair's moved / Pmove entries too (MALA.cpp:425-446)
    a.Pmove[m] = src_r;
    if (m == 0 && src_acc) a.counters[1] += 1;
    a.counters[8 + m] += src_acc;
    if (m == 0) a.counters[0] = it + 1;
    if (a.stats && rec >= 0) {  // update_buffer_stat_criteria (MALA.cpp:708)
        double *r = a.stats + ((size_t)rec * C + m) * 3;
        r[0] = o.logL; r[1] = o.logPr; r[2] = o.logPost;
    }
}

// Writes chain m's settled state for the next iteration: position = chain `src`'s post-test position (its own, or the swap partner's),
// scalars from `o` (already re-tempered after a swap); records the sample; names the slot of chain m's next proposal.
template <class AT, class FT>
__device__ __forceinline__ void fused_finalize(const AT &a, const FT &f, int m, int src, int src_acc, double src_r,
                                               const AcceptOut &o, int next_slot, long it, int q, long rec) {
    const int lane = threadIdx.x, C = a.C, Nv = a.Nv, Np = a.desc.Np, q1 = q ^ 1;
    const double *sv, *sp;
    if (src_acc) {
        const int ps = f.slot[q * C + src] & 0xffff;
        sv = f.cand_vars + ((size_t)q * f.NS + ps) * Nv;
        sp = f.cand_params + ((size_t)q * f.NS + ps) * Np;
    } else {
        sv = a.vars_cur + ((size_t)q * C + src) * Nv;
        sp = a.params_cur + ((size_t)q * C + src) * Np;
    }
    double *dv = a.vars_cur + ((size_t)q1 * C + m) * Nv, *dp = a.params_cur + ((size_t)q1 * C + m) * Np;
    double *rv = (a.samples && rec >= 0) ? a.samples + ((size_t)rec * C + m) * Nv : nullptr;  // update_buffer_params (MALA.cpp:710)
    for (int i = lane; i < Nv; i += 64) { const double v = sv[i]; dv[i] = v; if (rv) rv[i] = v; }
    for (int i = lane; i < Np; i += 64) dp[i] = sp[i];
    if (lane == 0) fused_scalars(a, f, m, src_acc, src_r, o, next_slot, it, q, rec);
}

// Chain m of the swap pair (A, A+1) has done its MH test (outcome in the arguments): publish it; the second of the two to get here
// resolves the swap (MALA.cpp:397-461) and writes both chains' settled states.
__device__ __attribute__((noinline)) void fused_settle_pair(const DevSamplerArgs *ga, const FusedArgs *gf, int m, int A, double u, int o_acc, double o_r,
                                                            double o_logL, double o_logPr, double o_logPost, long it, int q, long rec) {
    const ConstArgs &a = *(const ConstArgs *)uniform_ptr(ga);
    const ConstFused &f = *(const ConstFused *)uniform_ptr(gf);
    const int lane = threadIdx.x, C = a.C;
    AcceptOut o;
    o.acc = o_acc; o.r = o_r; o.logL = o_logL; o.logPr = o_logPr; o.logPost = o_logPost;
    if (lane == 0) {
        double *w = f.acc + ((size_t)q * C + m) * 5;
        coherent_store(w, (double)o.acc); coherent_store(w + 1, o.r); coherent_store(w + 2, o.logL); coherent_store(w + 3, o.logPr);
        coherent_store(w + 4, o.logPost);
    }
    drain_memory_ops();
    unsigned first = 0;
    const int xb = (A >= f.xsplit) ? 1 : 0;  // the pair's extra block
    if (lane == 0) first = atomicAdd(&f.pair_ticket[2 * q + xb], 1u);
    first = __shfl(first, 0, 64);
    if (first == 0) return;  // the partner is still being evaluated: its last tile does the rest
    const int partner = (m == A) ? A + 1 : A;
    AcceptOut op = {0, 0., 0., 0., 0.};
    if (lane == 0) {
        const double *w = f.acc + ((size_t)q * C + partner) * 5;
        op.acc = (int)coherent_load(w); op.r = coherent_load(w + 1); op.logL = coherent_load(w + 2); op.logPr = coherent_load(w + 3);
        op.logPost = coherent_load(w + 4);
    }
    op.acc = __shfl(op.acc, 0, 64); op.r = __shfl(op.r, 0, 64);
    op.logL = __shfl(op.logL, 0, 64); op.logPr = __shfl(op.logPr, 0, 64); op.logPost = __shfl(op.logPost, 0, 64);
    AcceptOut oA = (m == A) ? o : op, oB = (m == A) ? op : o;
    const int accA = oA.acc, accB = oB.acc;
    const double rA = oA.r, rB = oB.r;
    const int swapped = resolve_swap(a, A, u, oA, oB);
    if (lane == 0) {
        atomicAdd((unsigned long long *)&a.counters[2], 1ull);
        if (swapped) atomicAdd((unsigned long long *)&a.counters[3], 1ull);
    }
    const int B = A + 1;
    if (swapped) {  // each side continues from the other's post-test position: the extra candidate slots 2C .. 2C+3
        fused_finalize(a, f, A, B, accB, rB, oA, 2 * C + (A >= f.xsplit ? 4 : 0) + accB, it, q, rec);
        fused_finalize(a, f, B, A, accA, rA, oB, 2 * C + (B >= f.xsplit ? 4 : 0) + 2 + accA, it, q, rec);
    } else {
        fused_finalize(a, f, A, A, accA, rA, oA, 2 * A + accA, it, q, rec);
        fused_finalize(a, f, B, B, accB, rB, oB, 2 * B + accB, it, q, rec);
    }
}

// The chain's settle step, run by the wave of the chain's LAST tile (a real function call with pointer arguments, like the candidate
// roles: inlined into the tile body it would raise the kernel's register allocation above three waves per SIMD).
__device__ __attribute__((noinline)) void fused_settle(const DevSamplerArgs *ga, const FusedArgs *gf, int m, int ps, long it, int q, long rec) {
    const ConstArgs &a = *(const ConstArgs *)uniform_ptr(ga);
    const ConstFused &f = *(const ConstFused *)uniform_ptr(gf);
    const int lane = threadIdx.x, C = a.C, Nv = a.Nv, Np = a.desc.Np;
    // Everything that does not depend on the sums is requested first (this wave is the launch's critical tail): the proposal's prior and
    // status, what the chain holds, and BOTH vectors the chain may continue from (its position and its proposal).
    // (ps = the slot of the chain's proposal: the tile that calls knows it, no load needed to find the proposal's data)
    int stP = 0, stR = 0;
    double c_logPr = 0, h_logL = 0, h_logPr = 0, h_logPost = 0, Tm = 1, il = 0;
    if (lane == 0) {
        stP = f.cand_stP[q * f.NS + ps]; stR = f.cand_stR[q * f.NS + ps]; c_logPr = f.cand_logPr[q * f.NS + ps];
        h_logL = a.logL_cur[q * C + m]; h_logPr = a.logPr_cur[q * C + m]; h_logPost = a.logPost_cur[q * C + m];
        Tm = a.Tcoefs[m]; il = a.init_logL[m];
    }
    constexpr int ME = 2;  // vector elements per lane held in registers (longer vectors take the generic copy)
    const bool in_regs = Nv <= 64 * ME && Np <= 64 * ME;
    double r_pv[ME], r_cv[ME], r_pp[ME], r_cp[ME];
    if (in_regs) {
        const double *pv = f.cand_vars + ((size_t)q * f.NS + ps) * Nv, *pp = f.cand_params + ((size_t)q * f.NS + ps) * Np;
        const double *cv = a.vars_cur + ((size_t)q * C + m) * Nv, *cp = a.params_cur + ((size_t)q * C + m) * Np;
#pragma unroll
        for (int e = 0; e < ME; e++) {
            const int i = lane + 64 * e;
            r_pv[e] = i < Nv ? pv[i] : 0.0; r_cv[e] = i < Nv ? cv[i] : 0.0;
            r_pp[e] = i < Np ? pp[i] : 0.0; r_cp[e] = i < Np ? cp[i] : 0.0;
        }
    }
    // ---- the chain's MH test (MALA.cpp:490-551)
    const double S = wave_partial_sum(a.partials + (size_t)m * a.ntiles * 2, a.ntiles);
    AcceptOut o = {0, 0., 0., 0., 0.};
    if (lane == 0) o = mh_outcome(a, m, it, S, c_logPr, stP != TAMCMC_OK ? stP : stR, h_logL, h_logPr, h_logPost, Tm, il);
    o.acc = __shfl(o.acc, 0, 64); o.r = __shfl(o.r, 0, 64);
    o.logL = __shfl(o.logL, 0, 64); o.logPr = __shfl(o.logPr, 0, 64); o.logPost = __shfl(o.logPost, 0, 64);
    // ---- parallel tempering (MALA.cpp:397-461): the second chain of the pair to get here resolves the swap for both
    int A = -1;
    double u = 0;
    if (is_swap_iter(a, it)) A = swap_first(a, it, &u);
    if (A < 0 || (m != A && m != A + 1)) {  // not in the swap pair (or no swap step at this iteration)
        if (!in_regs) { fused_finalize(a, f, m, m, o.acc, o.r, o, 2 * m + o.acc, it, q, rec); return; }
        // the chain keeps its own position or takes its own proposal: both are in registers
        const int q1 = q ^ 1;
        double *dv = a.vars_cur + ((size_t)q1 * C + m) * Nv, *dp = a.params_cur + ((size_t)q1 * C + m) * Np;
        double *rv = (a.samples && rec >= 0) ? a.samples + ((size_t)rec * C + m) * Nv : nullptr;  // update_buffer_params (MALA.cpp:710)
#pragma unroll
        for (int e = 0; e < ME; e++) {
            const int i = lane + 64 * e;
            if (i < Nv) { const double v = o.acc ? r_pv[e] : r_cv[e]; dv[i] = v; if (rv) rv[i] = v; }
            if (i < Np) dp[i] = o.acc ? r_pp[e] : r_cp[e];
        }
        if (lane == 0) fused_scalars(a, f, m, o.acc, o.r, o, 2 * m + o.acc, it, q, rec);
        return;
    }
    fused_settle_pair(ga, gf, m, A, u, o.acc, o.r, o.logL, o.logPr, o.logPost, it, q, rec);
}

// Tail of the likelihood tiles of the fused step: every tile's wave calls it once its partial sums are written.
struct SettleTail {
    const DevSamplerArgs &a;
    const FusedArgs &f;
    const DevSamplerArgs *ga;
    const FusedArgs *gf;
    long it, rec;
    int q, first;  // first: chain of the launch's evaluation 0 (a launch covers the chains of one group, or all of them)
    static constexpr bool coherent_partials = true;
    __device__ __forceinline__ void operator()(int b, int tile, int ps) const {
        const int lane = threadIdx.x, C = a.C, m = first + b;
        drain_memory_ops();  // this tile's two partial sums (write-through stores) are in memory before the ticket counts the tile
        unsigned *tk = f.ticket + ((size_t)q * C + m) * TK;
        const int g = tile % NG, in_group = (a.ntiles - g + NG - 1) / NG;  // tiles g, g+NG, ... < ntiles
        unsigned old = 0;
        if (lane == 0) {
            old = atomicAdd(&tk[(1 + g) * TKS], 1u);
            if (old == (unsigned)(in_group - 1)) old = atomicAdd(&tk[0], 1u) + 0x10000u;  // the group's last tile reports the group
        }
        old = __shfl(old, 0, 64);
        const int ngroups = a.ntiles < NG ? a.ntiles : NG;
        if (old != 0x10000u + (unsigned)(ngroups - 1)) return;  // not the chain's last tile (wave-uniform)
        fused_settle(ga, gf, m, ps, it, q, rec);
    }
};

// The three kinds of work on one candidate (see candidate_role); the proposal vector is in LDS.
// (They are real function calls -- see candidate_role -- so their arguments are pointers to the DEVICE-MEMORY copies of the argument
// blocks: a reference to a kernel argument would have to be copied to the scratch stack first.)
__device__ __attribute__((noinline)) void role_prior(const DevSamplerArgs *ga, const FusedArgs *gf, size_t gs, const double *s_vars,
                                                     const double *s_params, const UnpackLds *Up) {
    const DevSamplerArgs &a = *ga;
    const FusedArgs &f = *gf;
    const UnpackLds U = *Up;
    const int Nv = a.Nv, Np = a.desc.Np, tid = threadIdx.x;
    for (int i = tid; i < Nv; i += 64) f.cand_vars[gs * Nv + i] = s_vars[i];
    for (int i = tid; i < Np; i += 64) f.cand_params[gs * Np + i] = s_params[i];
    const double logPr = wave_log_prior(a.desc, s_params, U, TB - 128);  // the proposal kernel's 128 term lanes (dev_unpack.h)
    if (tid == 0) { f.cand_logPr[gs] = logPr; f.cand_stP[gs] = *U.status; }
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
    wg_unpack(a.desc, s_params, U, slot, T, true, false, false, true);
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
    const int half = (a.ntiles + 1) / 2;
    wg_bg_tiles(a.desc, s_params, U.S, slot, T, 0, 64, role == 2 ? 0 : half, role == 2 ? half : a.ntiles);
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

// One role of one candidate slot of iteration `itn`, by ONE wave.  Slot s < 2C: chain s/2, built on its current position (even) or
// on its proposal of iteration itn-1 (odd); slots 2C..2C+3 (only when itn-1 swaps a pair A,B): chain A on B's two vectors, chain B on
// A's two.  Roles: 0 = position + log-prior, 1 = table rows + noise row, 2 / 3 = background series of the lower / upper half of the
// tiles.  Every role re-derives the proposal vector itself (no communication between the roles).
__device__ void candidate_role(const DevSamplerArgs &a, const FusedArgs &f, const DevSamplerArgs *ga, const FusedArgs *gf, long itn, int q_src,
                               int q_dst, int slot, int role, bool entry, unsigned char *lds) {
    const int C = a.C, Nv = a.Nv, Np = a.desc.Np, tid = threadIdx.x;
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
    if (role == 0 && slot < 2 * C && !on_prop && tid == 0) {  // housekeeping for the launch that evaluates these candidates
        for (int g = 0; g <= NG; g++) f.ticket[((size_t)q_dst * C + m) * TK + g * TKS] = 0u;
        // (by the first chain of iteration itn's swap pair: with chain groups that chain's launches are the ones that use the counter)
        if (is_swap_iter(a, itn) && m == swap_first(a, itn, nullptr)) f.pair_ticket[2 * q_dst + (m >= f.xsplit ? 1 : 0)] = 0u;
        if (entry) f.slot[q_dst * C + m] = 2 * m;
    }
    double *s_params = (double *)lds;
    double *s_vars = s_params + Np;
    double *s_z = s_vars + Nv;
    const UnpackLds U = carve_unpack_lds((unsigned char *)(s_z + Nv + 1));
    __shared__ UnpackLds s_U;  // handed to the role functions by address
    if (tid == 0) s_U = U;
    const double *bv, *bp;
    if (on_prop) {
        const int ps = f.slot[q_src * C + src] & 0xffff;
        bv = f.cand_vars + ((size_t)q_src * f.NS + ps) * Nv;
        bp = f.cand_params + ((size_t)q_src * f.NS + ps) * Np;
    } else {
        bv = a.vars_cur + ((size_t)q_src * C + src) * Nv;
        bp = a.params_cur + ((size_t)q_src * C + src) * Np;
    }
    for (int i = tid; i < Nv; i += 64) s_vars[i] = bv[i];
    for (int i = tid; i < Np; i += 64) s_params[i] = bp[i];
    const double *lz = f.lz + ((size_t)q_dst * C + m) * Nv;  // L z(itn) of chain m, computed one launch ahead (lz_block)
    unpack_begin(a.desc, U);  // (barrier)
    for (int i = tid; i < Nv; i += 64) s_vars[i] = s_vars[i] + 0.0 + lz[i];  // same expression as propose_common
    __syncthreads();
    for (int k = tid; k < Nv; k += 64) s_params[a.index_to_relax[k]] = s_vars[k];  // update_params_with_vars
    __syncthreads();
    const size_t gs = (size_t)q_dst * f.NS + slot;
    // (three separate functions: inlined side by side the roles' code raises the whole kernel's register allocation above the
    // three-waves-per-SIMD budget of the tile path)
    if (role == 0) role_prior(ga, gf, gs, s_vars, s_params, &s_U);
    else if (role == 1) role_rows(ga, gf, q_dst, slot, gs, s_params, &s_U);
    else role_background(ga, gf, q_dst, slot, role, s_params, &s_U);
}

// Per-launch scalars of the fused step.
struct StepCtl {
    long it, rec, it_lz;   // iteration of the tiles / candidates; record index (-1: none); first iteration of the L z blocks
    int q, flags;          // parity of iteration `it`; ST_* bits
    int nbr, nlz;          // workgroups reserved for candidate roles / L z blocks (multiples of 8: keeps the tiles' XCD mapping)
    int n_lz_live, q_lz;   // L z blocks that have work (chain first + e % cnt of iteration it_lz + e / cnt); parity of it_lz
    int first, cnt;        // the chains of this launch: [first, first + cnt) -- all of them, or one chain group (see run(): fused)
    int extra;             // 1: the launch also builds the four extra candidates of its iteration's swap pair (slots 2C..2C+3)
    const DevSamplerArgs *ga;  // device-memory copies of the first two kernel arguments (for the candidate roles' function calls)
    const struct FusedArgs *gf;
};

// Launch `it` of a fused stretch: [0, nbr) candidate roles of iteration it+1 (ST_BR; at the entry of a stretch, ST_ENTRY: of iteration
// `it` itself from the settled chains), [nbr, nbr+nlz) L z of later iterations (ST_LZ), then the likelihood tiles of iteration `it` (ST_L).
#define TAMCMC_STEP_BODY                                                                                                      \
    __shared__ tile::TileLds<MODE, 64> lds;                                                                                  \
    const int id = (int)blockIdx.x;                                                                                          \
    if (id < c.nbr) {                                                                                                        \
        const int k = id >> 2, slot = k < 2 * c.cnt ? 2 * c.first + k : 2 * a.C + (k - 2 * c.cnt); /* the group's slots, then the pair's */ \
        if (k >= 2 * c.cnt && !c.extra) return;                                                                              \
        if (c.flags & ST_ENTRY) candidate_role(a, f, c.ga, c.gf, c.it, c.q, c.q, slot, id & 3, true, (unsigned char *)&lds);    \
        else if (c.flags & ST_BR)                                                                                            \
            candidate_role(a, f, c.ga, c.gf, c.it + 1, c.q, c.q ^ 1, slot, id & 3, false, (unsigned char *)&lds);              \
        return;                                                                                                              \
    }                                                                                                                        \
    if (id < c.nbr + c.nlz) {                                                                                                \
        const int e = id - c.nbr;                                                                                            \
        if (e < c.n_lz_live)                                                                                                 \
            lz_block(c.ga, c.gf, c.it_lz + e / c.cnt, (c.q_lz ^ (e / c.cnt)) & 1, c.first + e % c.cnt, (unsigned char *)&lds);  \
        return;                                                                                                              \
    }                                                                                                                        \
    if (c.flags & ST_L)                                                                                                      \
        tile::loglike_tile<MODE, 64, K, false, false>(la, id - c.nbr - c.nlz, lds, SettleTail{a, f, c.ga, c.gf, c.it, c.rec, c.q, c.first});
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
    DevBuf<double> fused_bg;  // (A): the candidates' background series (see run())
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
    if (impl->a.counters && getenv("TAMCMC_PROBE_ADAPT")) {
        long h[8];
        (void)hipMemcpy(h, impl->a.counters, sizeof h, hipMemcpyDeviceToHost);
        if (h[7] > 0)
            fprintf(stderr, "Cholesky panels of chain 0 (us per panel): columns below %.2f | next panel's columns %.2f | next diagonal block beside the rest of the trailing update %.2f  (%ld panels)\n",
                    0.01 * h[4] / h[7], 0.01 * h[5] / h[7], 0.01 * h[6] / h[7], h[7]);
    }
#endif
    for (void *p : impl->allocs) (void)hipFree(p);
    impl->fd_block.release(); impl->fd_part.release(); impl->fd_S.release(); impl->fd_model.release(); impl->fd_bg.release(); impl->fused_bg.release();
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
    DCHK(I.dalloc(&a.Pmove, C)); DCHK(I.dalloc(&a.moved, C)); DCHK(I.dalloc(&a.counters, 8 + C));
    a.grad_cur = nullptr; a.gradP_cur = nullptr;
    if (I.use_drift) {
        DCHK(I.dalloc(&a.grad_cur, 2 * C * Nv)); DCHK(I.dalloc(&a.gradP_cur, 2 * C * Nv));
        DCHK(I.dalloc(&I.mala.grad_prop, C * Nv)); DCHK(I.dalloc(&I.mala.gradP_prop, C * Nv)); DCHK(I.dalloc(&I.mala.drift_cur, C * Nv));
        DCHK(I.dalloc(&I.mala.out, C * 5));
    }
    DCHK(I.dalloc(&a.lz, 2 * C * Nv)); DCHK(I.dalloc(&a.LT, C * Nv * Nv)); DCHK(I.dalloc(&a.cov, C * Nv * Nv)); DCHK(I.dalloc(&a.mu, C * Nv)); DCHK(I.dalloc(&a.sigma, C));
    DCHK(I.dalloc(&a.mults, CD * (size_t)a.desc.per + 1)); DCHK(I.dalloc(&a.pairs, 2 * CD)); DCHK(I.dalloc(&a.nh, CD)); DCHK(I.dalloc(&a.nn, CD));
    DCHK(I.dalloc(&a.noise, CD * (size_t)a.desc.stride));
    DCHK(hipMemsetAsync(a.counters, 0, (8 + C) * sizeof(long), st));
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
    {  // (A) fused step: 2C+4 candidate slots per iteration parity, tickets, outcomes (tables are sized at the first run())
        FusedArgs &f = I.f;
        f.NS = 2 * in.C + 8;
        {   // two chain groups for the fused step (see run()): with the default groups, from 8 chains on
            const int h = (int)(((long)in.C * 1) / 2);
            f.xsplit = (I.G == 2 && h >= 3 && in.C - h >= 3) ? h : in.C;
        }
        const size_t NS = (size_t)f.NS;
        DCHK(I.dalloc(&f.cand_vars, 2 * NS * Nv)); DCHK(I.dalloc(&f.cand_params, 2 * NS * Np)); DCHK(I.dalloc(&f.cand_logPr, 2 * NS));
        DCHK(I.dalloc(&f.cand_stP, 2 * NS)); DCHK(I.dalloc(&f.cand_stR, 2 * NS));
        DCHK(I.dalloc(&f.mults, 2 * NS * (size_t)a.desc.per + 1)); DCHK(I.dalloc(&f.pairs, 4 * NS)); DCHK(I.dalloc(&f.nh, 2 * NS)); DCHK(I.dalloc(&f.nn, 2 * NS));
        DCHK(I.dalloc(&f.noise, 2 * NS * (size_t)a.desc.stride));
        DCHK(I.dalloc(&f.slot, 2 * C)); DCHK(I.dalloc(&f.ticket, 2 * C * TK)); DCHK(I.dalloc(&f.lz, 2 * C * Nv)); DCHK(I.dalloc(&f.pair_ticket, 4)); DCHK(I.dalloc(&f.acc, 2 * C * 5));
        DCHK(hipMemsetAsync(f.nn, 0, 2 * NS * sizeof(int), st));
        DCHK(hipMemsetAsync(f.cand_stP, 0, 2 * NS * sizeof(int), st));
        DCHK(hipMemsetAsync(f.cand_stR, 0, 2 * NS * sizeof(int), st));
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

int DevSampler::run(long it0, long n_iter, const char *learn, double *samples, double *stats) {
    RunningCall running_call;
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
            DCHK(I.fused_bg.reserve(2 * NS * (size_t)a.ntiles * 8));
            I.f.bg = I.fused_bg.p;
        }
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
    LoglikeArgs la[4], lf[2];
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
    for (int q = 0; q < 2; q++) {  // (A): evaluation m = chain m, its table in slot f.slot[q][m] of parity q's candidate block
        LoglikeArgs &l = lf[q];
        const FusedArgs &f = I.f;
        fill_common(l, a.C);
        l.mults = f.mults + (size_t)q * NS * a.desc.per; l.offsets = f.pairs + (size_t)q * 2 * NS; l.noise = f.noise + (size_t)q * NS * a.desc.stride;
        l.nharvey = f.nh + (size_t)q * NS; l.nnoise = f.nn + (size_t)q * NS; l.partials = a.partials;
        l.bg_poly = f.bg ? f.bg + (size_t)q * NS * a.ntiles * 8 : nullptr;
        l.slot_map = f.slot + (size_t)q * C;
        l.per = a.desc.per; l.slot0 = 0;
    }

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
        const int nbr = 4 * f.NS;                                     // candidate roles, a multiple of 8 (keeps the tiles' XCD mapping)
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
            DCHK(launch_step(c->precision, c->K, nlz2, st, args, f, lf[q], sc));
            sc.flags = ST_ENTRY; sc.nbr = nbr; sc.nlz = 0; sc.n_lz_live = 0;
            DCHK(launch_step(c->precision, c->K, nbr, st, args, f, lf[q], sc));
        }
        // the likelihood kernel's time for the roofline: two events around the whole stretch, i.e. the average includes the time between
        // two launches
        const bool timed = c->timing && fused_ev.size() < 16;
        const int fe = I.n_ev - 1 - (int)fused_ev.size();
        // (it pays once one launch no longer fits the GPU's resident waves -- 20 chains x 196 tiles: 27.7 -> 23.9 us, x 782 tiles: 59.8 ->
        // 49.6 us -- and costs below that: 8 chains x 196 tiles 20.5 -> 23.7 us, 20 chains x 20 tiles 33.5 -> 35.6 us; tools/groups_probe.py)
        const bool split = f.xsplit < a.C && c->step_scheme != 2 && (c->step_scheme == 3 || (long)a.C * a.ntiles >= 2500);
        if (timed && !split) DCHK(hipEventRecord(I.ev[fe][0], st));
        // Two chain groups, each with its own launch per iteration on its own stream: a launch is a chain of dependent steps (slot ->
        // table rows -> tile -> ticket -> settle, ~20 us even for five chains) that leaves most of the GPU idle at its two ends; the two
        // groups' launches fill each other's ends (two 10-chain stars side by side: 24.2 us per iteration each, one 20-chain launch: 27.8).
        // Nothing is shared between the groups' launches except at a swap whose pair straddles the groups: that iteration is ONE launch
        // over all chains on the context stream, with an event each way.  (Same chains bit for bit: the launches' contents are the same.)
        const int first1 = f.xsplit;
        hipStream_t s1 = I.gst[1];
        long n_split = 0;
        bool s1_ahead = false, s1_must_wait = st_has_work;  // s1 holds launches st has not waited for / s1 has not seen st's latest launches
        auto swap_pair_of = [&](long it) -> int {
            if (!(a.C >= 2 && a.dN_mixing > 0 && (it % a.dN_mixing == 0) && it != 0)) return -1;
            double u, u2;
            rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)it, 0, u, u2);
            int A = (int)(u2 * (double)(a.C - 1));
            if (A > a.C - 2) A = a.C - 2;
            return A;
        };
        auto launch_group = [&](int first, int cnt, int A, long i, hipStream_t stream, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) -> int {
            // every launch also prepares the next iteration's candidates and the L z after that -- the last one too (see armed_it)
            const bool owns_pair = A >= first && A + 1 < first + cnt;
            sc.it = it0 + i; sc.rec = (samples || stats) ? i : (long)-1; sc.q = q;
            sc.flags = ST_L | ST_BR | ST_LZ;
            sc.first = first; sc.cnt = cnt; sc.extra = owns_pair ? 1 : 0;
            sc.nbr = 4 * (2 * cnt + (owns_pair ? 4 : 0)); sc.nlz = ((cnt + 7) / 8) * 8; sc.n_lz_live = cnt; sc.it_lz = it0 + i + 2; sc.q_lz = q;
            LoglikeArgs lq = lf[q];
            lq.B = cnt;
            lq.slot_map = lf[q].slot_map + first;
            lq.partials = lf[q].partials + (size_t)first * a.ntiles * 2;
            if (owns_pair && cnt >= 3) lq.prio_b = A - first;  // this iteration's swap pair leads the launch
            DCHK(launch_step(c->precision, c->K, sc.nbr + sc.nlz + ntiles_pad * cnt, stream, args, f, lq, sc, e0, e1));
            return TAMCMC_OK;
        };
        for (long i = ia; i < ib; i++) {
            const int A = swap_pair_of(it0 + i);
            if (split && A != first1 - 1) {
                if (s1_must_wait) {
                    DCHK(hipEventRecord(I.ev_fork, st));
                    DCHK(hipStreamWaitEvent(s1, I.ev_fork, 0));
                    s1_must_wait = false;
                }
                int rc = launch_group(0, first1, A, i, st);
                if (rc) return rc;
                const bool sample = timed && g_used < I.n_gev && (len >= 97 ? ((i - ia) % 97 == 48) : (i - ia == len / 2));
                rc = sample ? launch_group(first1, a.C - first1, A, i, s1, I.gev[g_used][0], I.gev[g_used][1]) : launch_group(first1, a.C - first1, A, i, s1);
                if (rc) return rc;
                if (sample) g_used++;
                s1_ahead = true;
                n_split++;
            } else {
                if (s1_ahead) {
                    DCHK(hipEventRecord(I.ev_join[1], s1));
                    DCHK(hipStreamWaitEvent(st, I.ev_join[1], 0));
                    s1_ahead = false;
                }
                int rc = launch_group(0, a.C, A, i, st);
                if (rc) return rc;
                s1_must_wait = true;
            }
            q ^= 1;
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
            if (nb.windowed) DCHK(I.fd_model.reserve(2 * C * (size_t)c->Nx));  // two planes: 1/M0, y/M0 of the base points
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
        hipLaunchKernelGGL(k_mala_test, dim3(a.C), dim3(TB), lds_test0 + ((learn_i && chol_lds) ? lds_adapt : 0), st, args, M, it, P, learn_i,
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
                fd_bins_sampled += bins;
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
