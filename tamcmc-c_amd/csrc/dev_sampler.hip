// dev_sampler.hip -- device-resident MCMC iteration (sampler-side algebra on the device).
//
// The host-driven loop (host_mala.cpp) spends ~3/4 of a step on the host (proposal, priors, table build, copies,
// one sync per step).  Here one MCMC iteration of ALL tempered chains is TWO kernels on the context's stream, with no
// host round trip and no copy in between:
//   k_iterate  (one workgroup per chain)
//      (0) settles the previous iteration: per-chain partial sums -> tempered logL, MH test (MALA.cpp:490-551),
//          adjacent-pair parallel-tempering swap (MALA.cpp:397-461), sample/stat record (the outputs.cpp buffers),
//          in learning phases the Robbins-Monro update of mu, Sigma, sigma (MALA.cpp:296-319) and the Cholesky
//          factor of (Sigma+eps2 I) sigma (MALA.cpp:348-350);
//      (1) proposes the next one: z ~ N(0,I) (Philox, same streams as the host engine), x' = x + L z
//          (MALA.cpp:339-369), params' (model_def.cpp:484-492), log-prior with its terms spread over the lanes
//          (priors_calc.cpp), params' -> multiplet table + noise row (models.cpp unpackers) written straight into
//          the likelihood kernel's input block.
//   k_loglike  (kernels.hip)  the hot kernel, unchanged.
// All per-iteration state is double-buffered by parity: a workgroup reads parity P (any chain) and writes parity P^1
// (its own chain only), so the swap needs no inter-workgroup synchronisation: the two workgroups of a swap pair
// both recompute both MH tests from the same inputs and reach the same decision.
// The host only enqueues launches and fetches the recorded samples once per run() call.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "ctx.h"
#include "dev_sampler.h"
#include "kernels.h"
#include "dev_unpack.h"
#include "mode_tables.h"
#include "rng.h"

namespace tamcmc {

namespace {

__global__ void k_fill_poly(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
}

constexpr int TB = 256;  // threads of k_iterate (one workgroup per chain)

// Outcome of the Metropolis-Hastings test of chain j for the pending iteration (MALA.cpp:490-551): the values the
// chain holds AFTER the test.  Computed by a whole workgroup; every workgroup that needs chain j's outcome (the chain's
// own workgroup and, in a swap step, its partner's) recomputes it from the same inputs -> identical results.
struct AcceptOut {
    int acc;
    double r, logL, logPr, logPost;
};

__device__ void accept_result(const DevSamplerArgs &a, int j, long itp, int P, double *s_red, AcceptOut *s_out) {
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
        const double S = t1 + t2;
        const int C = a.C;
        double logL = (-(double)a.pl * S) / a.Tcoefs[j];  // call_likelihood, model_def.cpp:399-401
        const double logPr = a.logPr_prop[P * C + j];
        double logPost;
        if (a.status_prop[P * C + j] != TAMCMC_OK) logL = NAN;
        if (logPr == -INFINITY || isnan(logPr)) { logL = a.init_logL[j]; logPost = -INFINITY; }  // model_def.cpp:476-480
        else logPost = logL + logPr;
        double u, u1;
        rng_uniform2(a.seed, RNG_ACCEPT, (uint32_t)j, (uint64_t)itp, 0, u, u1);
        double r;
        if (!isnan(logL)) {
            if (logPost == -INFINITY) r = 0.;
            else {
                const double e = exp(logPost - a.logPost_cur[P * C + j]);
                r = fmin(1.0, e);
                if (isnan(r)) r = 0.;
            }
        } else r = 0.;
        AcceptOut o;
        o.acc = (u <= r) ? 1 : 0;
        o.r = r;
        if (o.acc) { o.logL = logL; o.logPr = logPr; o.logPost = logPost; }
        else { o.logL = a.logL_cur[P * C + j]; o.logPr = a.logPr_cur[P * C + j]; o.logPost = a.logPost_cur[P * C + j]; }
        *s_out = o;
    }
    __syncthreads();
}

// Robbins-Monro adaptation of chain m's proposal law (MALA.cpp:296-319) and Cholesky of (Sigma+eps2 I) sigma
// (MALA.cpp:348-350); `vars` = the chain's position after the MH test, `Pm` = its move probability.
__device__ void adapt_chain(const DevSamplerArgs &a, int m, long itp, const double *vars, double Pm, double *A, double *d,
                            double *s_red, double *s_scal) {
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
    n2 = 0;
    for (int e = tid; e < Nv * Nv; e += TB) {
        const int i = e / Nv, j = e - i * Nv;
        const double v = cov[e] + g * (d[i] * d[j] - cov[e]);
        cov[e] = v;
        n2 += v * v;
    }
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
    for (int e = tid; e < Nv * Nv; e += TB) {
        const int i = e / Nv, j = e - i * Nv;
        double v = cov[e];
        if (sc != 1.0) { v = v * sc; cov[e] = v; }
        A[e] = (v + (i == j ? a.epsi2 : 0.0)) * sig;
    }
    __syncthreads();
    // right-looking Cholesky in place (lower triangle of A): column scale + trailing update per step.  A matrix that is not
    // positive definite (possible only while gamma = c0/(1+i) > 1, i.e. adaptation before iteration c0) keeps the PREVIOUS
    // factor -- the host engine does the same (host_mala.cpp::factor); the reference hands Eigen's partial result on.
    for (int j = 0; j < Nv; j++) {
        const double ajj = A[(size_t)j * Nv + j];  // workgroup-uniform
        if (!(ajj > 0.0)) return;                  // every lane leaves together, L is not touched
        const double djj = sqrt(ajj);
        __syncthreads();
        if (tid == 0) A[(size_t)j * Nv + j] = djj;
        for (int i = j + 1 + tid; i < Nv; i += TB) A[(size_t)i * Nv + j] = A[(size_t)i * Nv + j] / djj;
        __syncthreads();
        const int rem = Nv - j - 1;
        for (int e = tid; e < rem * rem; e += TB) {
            const int i = j + 1 + e / rem, k = j + 1 + e % rem;
            if (k <= i) A[(size_t)i * Nv + k] = A[(size_t)i * Nv + k] - A[(size_t)i * Nv + j] * A[(size_t)k * Nv + j];
        }
        __syncthreads();
    }
    double *LT = a.LT + (size_t)m * Nv * Nv;
    for (int e = tid; e < Nv * Nv; e += TB) {
        const int i = e / Nv, k = e - i * Nv;
        LT[(size_t)k * Nv + i] = (k <= i) ? A[e] : 0.0;
    }
    __syncthreads();
}

// z ~ N(0, I) of (chain, iteration) into LDS (ends without a barrier) and row i of L z (MALA.cpp:348-355)
__device__ __forceinline__ void normals_into(const DevSamplerArgs &a, int chain, long it, double *s_z) {
    for (int k2 = threadIdx.x; 2 * k2 < a.Nv; k2 += TB) {
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

// Proposal of iteration `it` for `chain` from the state in LDS (s_vars/s_params): x' = x + L z (MALA.cpp:348-355), L =
// chol((Sigma+eps2) sigma) stored transposed, same Philox streams as the host engine; log-prior; params' -> multiplet table
// written into slot `slot` of the likelihood kernel's input block.  Ends without a barrier.
__device__ void propose_common(const DevSamplerArgs &a, const UnpackLds &U, int chain, long it, int slot, double *pv, double *pp,
                               double *logPr_out, int *status_out, double *s_vars, double *s_params, double *s_z, long *dbg,
                               const double *lz = nullptr) {
    const int Np = a.desc.Np, Nv = a.Nv, tid = threadIdx.x;
#define PSTAMP(k) do { if (dbg && tid == 0) dbg[k] = (long)wall_clock64(); } while (0)
    if (!lz) normals_into(a, chain, it, s_z);
    unpack_begin(a.desc, U);
    for (int i = tid; i < Nv; i += TB) {  // lane i owns row i: reads s_vars[i] only, every s_z[k]
        const double s = lz ? lz[i] : Lz_row(a, chain, i, s_z);
        const double v = s_vars[i] + 0.0 + s;
        s_vars[i] = v;
        pv[i] = v;
    }
    __syncthreads();
    PSTAMP(2);
    for (int k = tid; k < Nv; k += TB) s_params[a.index_to_relax[k]] = s_vars[k];  // update_params_with_vars
    __syncthreads();
    for (int i = tid; i < Np; i += TB) pp[i] = s_params[i];

    // ---- log-prior, then params' -> multiplet table written into the likelihood kernel's input block ----
    TablePtrs T;
    T.mults = a.mults; T.pairs = a.pairs; T.nh = a.nh; T.nn = a.nn; T.noise = a.noise;
    T.bg = a.bg; T.ntiles = a.ntiles; T.tile_bins = a.tile_bins;
    // four roles beside each other (dev_unpack.h): prior terms + background tiles | table rows | shared scalars + m-visibilities
    const double logPr = wg_log_prior(a.desc, s_params, U, true, dbg, true, &T, slot);
    PSTAMP(4);
    const bool live = (logPr != -INFINITY) && !isnan(logPr);  // model_def.cpp:472,476-480
    wg_unpack(a.desc, s_params, U, slot, T, live, dbg, true, true);
    if (tid == 0) {
        *logPr_out = logPr;
        *status_out = *U.status;
    }
    PSTAMP(6);
#undef PSTAMP
}

// ONE kernel per MCMC iteration besides the likelihood kernel.  Workgroup m:
//   (0) settles the pending iteration it-1 for chain m: MH test (own chain; the swap partner's too when chain m is in the
//       swap pair), adjacent-pair parallel-tempering swap, writes the chain's new current state into the OTHER parity
//       buffer (no workgroup ever writes what another one reads), records the sample, adapts the proposal law;
//   (1) proposes iteration `it` from that state: x' = x + L z, log-prior, params' -> multiplet table.
template <bool PROPOSE>
__global__ void __launch_bounds__(TB) k_iterate(const DevSamplerArgs a, const long it, const int P, const int pending,
                                               const long rec, const int learn_pending, double *scratch, const int c_off,
                                               const int nmain, const int pre_flags) {
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
#define STAMP(k) do { if (PROPOSE && a.dbg && m == 0 && tid == 0) a.dbg[k] = (long)wall_clock64(); } while (0)
    STAMP(0);
    const double *curv = a.vars_cur + (size_t)P * C * Nv, *curp = a.params_cur + (size_t)P * C * Np;
    const double *prpv = a.vars_prop + (size_t)P * C * Nv, *prpp = a.params_prop + (size_t)P * C * Np;
    double *newv = a.vars_cur + (size_t)Q * C * Nv, *newp = a.params_cur + (size_t)Q * C * Np;

    // ------------------------------------------------------------------ (0) settle the pending iteration
    if (pending) {
        const long itp = it - 1;
        accept_result(a, m, itp, P, s_red, &s_own);
        int src = m;
        double logL_new = s_own.logL, logPr_new = s_own.logPr, logPost_new = s_own.logPost;
        // parallel tempering (MALA.cpp:397-461): adjacent pair, tempered log-likelihoods after the MH tests
        const bool swap_step = a.dN_mixing > 0 && (itp % a.dN_mixing == 0) && itp != 0 && C > 1;
        if (swap_step) {
            double u, u2;
            rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)itp, 0, u, u2);
            int A = (int)(u2 * (double)(C - 1));
            if (A > C - 2) A = C - 2;
            const int B = A + 1;
            if (m == A || m == B) {  // workgroup-uniform branch
                const int partner = (m == A) ? B : A;
                accept_result(a, partner, itp, P, s_red, &s_partner);
                const double LA = (m == A) ? s_own.logL : s_partner.logL;
                const double LB = (m == A) ? s_partner.logL : s_own.logL;
                const double LA_TB = LA * a.Tcoefs[A] / a.Tcoefs[B];
                const double LB_TA = LB * a.Tcoefs[B] / a.Tcoefs[A];
                const double e = exp(LA_TB + LB_TA - LA - LB);
                const double rT = fmin(1.0, e);
                const bool swapped = (u <= rT);
                if (swapped) {
                    src = partner;
                    logPr_new = s_partner.logPr;
                    logL_new = (m == A) ? LB_TA : LA_TB;  // re-tempered value of the partner's likelihood
                    logPost_new = logL_new + logPr_new;
                }
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
            a.logL_cur[Q * C + m] = logL_new;
            a.logPr_cur[Q * C + m] = logPr_new;
            a.logPost_cur[Q * C + m] = logPost_new;
            // a swap exchanges the pair's moved / Pmove entries too (MALA.cpp:425-446): what is recorded is the partner's
            a.moved[m] = (src == m) ? s_own.acc : s_partner.acc;
            a.Pmove[m] = (src == m) ? s_own.r : s_partner.r;
            if (m == 0 && a.moved[0]) a.counters[1] += 1;
            if (m == 0) a.counters[0] = it;
            if (a.stats && rec >= 0) {  // update_buffer_stat_criteria (MALA.cpp:708)
                double *r = a.stats + ((size_t)rec * C + m) * 3;
                r[0] = logL_new; r[1] = logPr_new; r[2] = logPost_new;
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
    STAMP(1);
    if (!PROPOSE) return;
    __syncthreads();

    // ------------------------------------------------------------------ (1) propose iteration `it`
    propose_common(a, U, m, it, m, a.vars_prop + (size_t)Q * C * Nv + (size_t)m * Nv, a.params_prop + (size_t)Q * C * Np + (size_t)m * Np,
                   a.logPr_prop + Q * C + m, a.status_prop + Q * C + m, s_vars, s_params, s_z, (a.dbg && m == 0) ? a.dbg : nullptr,
                   (pre_flags & 1) ? a.lz + ((size_t)(it & 1) * C + m) * Nv : nullptr);
#undef STAMP
}


// ---------------------------------------------------------------------------------------------------------------
// Speculative rounds for stretches WITHOUT adaptation (the proposal law is frozen, e.g. the Acquire phase).
// A Metropolis-Hastings chain rejects ~3 proposals out of 4; when iteration i is rejected, the proposal of i+1 is
// x + L z_{i+1} from the SAME x.  A round therefore evaluates, per chain, the candidates of iterations d, d+1, .., d+D-1
// all built on the chain's current x; the next round tests them in order and consumes iterations up to and including the
// first acceptance (later candidates are discarded).  Every random number is addressed by (chain, iteration), so each
// chain follows exactly the trajectory of the one-iteration-per-round engine -- the chains merely stop advancing in
// lockstep.  Parallel-tempering swaps (iteration s, pair A_s) are the only coupling: a chain of the pair proposes no
// candidate beyond s, and after its MH test of s it waits (phase 1) until its partner has also tested s; both
// workgroups then resolve the swap from the same inputs.
struct SpecOut {
    long done;                     // iterations whose MH test is done, after this round's tests
    int phase;                     // 1 = waiting for the partner at the swap of iteration done-1
    int src_chain, src_cand;       // where the chain's state vector lives (parity P): cand -1 = src_chain's current vector
    int n_rec, acc_last, swap_last;  // iterations consumed by this round's tests; was the last one accepted; is it a swap step of this chain
    int resolved, swapped;
    int nprop_new;
    double r_last;
    double logL0, logPr0, logPost0;  // before this round's tests
    double logL, logPr, logPost;     // after the tests (and after the swap once resolved)
};

__device__ __forceinline__ bool is_swap_iter(const DevSamplerArgs &a, long i) {
    return a.dN_mixing > 0 && (i % a.dN_mixing == 0) && i != 0 && a.C > 1;
}
__device__ __forceinline__ int swap_first(const DevSamplerArgs &a, long i, double *u_out) {  // MALA.cpp:397-405
    double u, u2;
    rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)i, 0, u, u2);
    int A = (int)(u2 * (double)(a.C - 1));
    if (A > a.C - 2) A = a.C - 2;
    if (u_out) *u_out = u;
    return A;
}
__device__ __forceinline__ bool in_swap_pair(const DevSamplerArgs &a, int c, long i) {
    if (!is_swap_iter(a, i)) return false;
    const int A = swap_first(a, i, nullptr);
    return c == A || c == A + 1;
}

// Sums of the per-tile partials of chain c's candidates (wave w -> candidate w), in k_finalize's order: 256 strided
// per-thread sums (four per lane here), shuffle tree per 64, the four in order.  No barrier inside.
__device__ void spec_sums(const DevSamplerArgs &a, int c, int np, double *s_S) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= np) return;  // wave-uniform
    const double *base = a.partials + (size_t)(c * a.D + w) * a.ntiles * 2;
    double t1 = 0, t2 = 0;
    for (int q = 0; q < TB / 64; q++) {
        double s1 = 0, s2 = 0;
        for (int t = q * 64 + lane; t < a.ntiles; t += TB) {
            s1 = s1 + base[2 * t];
            s2 = s2 + base[2 * t + 1];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s1 = s1 + __shfl_down(s1, off, 64);
            s2 = s2 + __shfl_down(s2, off, 64);
        }
        if (q == 0) { t1 = s1; t2 = s2; }
        else { t1 = t1 + s1; t2 = t2 + s2; }
    }
    if (lane == 0) s_S[w] = t1 + t2;
}

// thread 0: the MH tests of chain c's candidates, in order (same arithmetic as accept_result)
__device__ void spec_tests(const DevSamplerArgs &a, int c, int P, int first, long it_a, const double *S, SpecOut &o) {
    const int C = a.C, D = a.D;
    long d = first ? it_a : a.sp_done[P * C + c];
    const int np = first ? 0 : a.sp_nprop[P * C + c];
    int ph = first ? 0 : a.sp_phase[P * C + c];
    o.logL0 = a.logL_cur[P * C + c]; o.logPr0 = a.logPr_cur[P * C + c]; o.logPost0 = a.logPost_cur[P * C + c];
    o.logL = o.logL0; o.logPr = o.logPr0; o.logPost = o.logPost0;
    o.src_chain = c; o.src_cand = -1;
    o.n_rec = 0; o.acc_last = 0; o.r_last = 0; o.resolved = 0; o.swapped = 0;
    for (int k = 0; k < np; k++) {
        const int slot = c * D + k;
        double logL = (-(double)a.pl * S[k]) / a.Tcoefs[c];  // call_likelihood, model_def.cpp:399-401
        const double logPr = a.logPr_prop[(size_t)P * C * D + slot];
        double logPost;
        if (a.status_prop[(size_t)P * C * D + slot] != TAMCMC_OK) logL = NAN;
        if (logPr == -INFINITY || isnan(logPr)) { logL = a.init_logL[c]; logPost = -INFINITY; }  // model_def.cpp:476-480
        else logPost = logL + logPr;
        double u, u1;
        rng_uniform2(a.seed, RNG_ACCEPT, (uint32_t)c, (uint64_t)d, 0, u, u1);
        double r;
        if (!isnan(logL)) {
            if (logPost == -INFINITY) r = 0.;
            else {
                const double e = exp(logPost - o.logPost0);
                r = fmin(1.0, e);
                if (isnan(r)) r = 0.;
            }
        } else r = 0.;
        const int acc = (u <= r) ? 1 : 0;
        o.n_rec++;
        d++;
        o.r_last = r;
        o.acc_last = acc;
        if (acc) {
            o.src_cand = k;
            o.logL = logL; o.logPr = logPr; o.logPost = logPost;
            break;
        }
    }
    o.swap_last = (o.n_rec > 0 && in_swap_pair(a, c, d - 1)) ? 1 : 0;
    if (o.swap_last) ph = 1;
    o.done = d;
    o.phase = ph;
}

__global__ void __launch_bounds__(TB) k_spec(const DevSamplerArgs a, const int P, const int first, const long it_a, const long it_b,
                                            const long rec_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int Np = a.desc.Np, Nv = a.Nv, C = a.C, D = a.D;
    double *s_params = (double *)s_raw;
    double *s_vars = s_params + Np;
    double *s_z = s_vars + Nv;
    const UnpackLds U = carve_unpack_lds((unsigned char *)(s_z + Nv + 1));
    __shared__ SpecOut s_me, s_pt;
    __shared__ double s_S[4], s_Sp[4];

    const int c = blockIdx.x / D, j = blockIdx.x - c * D, tid = threadIdx.x;
    const int Q = P ^ 1;
    const size_t CD = (size_t)C * D;

    // ---- (0) this chain's MH tests
    const int np = first ? 0 : a.sp_nprop[P * C + c];
    spec_sums(a, c, np, s_S);
    __syncthreads();
    if (tid == 0) spec_tests(a, c, P, first, it_a, s_S, s_me);
    __syncthreads();
    // ---- (1) swap: both chains of the pair have tested iteration s -> resolve (each side recomputes the other's tests)
    if (s_me.phase == 1) {  // workgroup-uniform
        const long sidx = s_me.done - 1;
        const int A = swap_first(a, sidx, nullptr);
        const int partner = (c == A) ? A + 1 : A;
        const int npp = first ? 0 : a.sp_nprop[P * C + partner];
        spec_sums(a, partner, npp, s_Sp);
        __syncthreads();
        if (tid == 0) {
            spec_tests(a, partner, P, first, it_a, s_Sp, s_pt);
            if (s_pt.phase == 1 && s_pt.done == s_me.done) {
                double u;
                swap_first(a, sidx, &u);
                const int B = A + 1;
                const double LA = (c == A) ? s_me.logL : s_pt.logL;
                const double LB = (c == A) ? s_pt.logL : s_me.logL;
                const double LA_TB = LA * a.Tcoefs[A] / a.Tcoefs[B];
                const double LB_TA = LB * a.Tcoefs[B] / a.Tcoefs[A];
                const double e = exp(LA_TB + LB_TA - LA - LB);
                const double rT = fmin(1.0, e);
                const bool swapped = (u <= rT);
                if (swapped) {
                    s_me.src_chain = partner;
                    s_me.src_cand = s_pt.src_cand;
                    s_me.logPr = s_pt.logPr;
                    s_me.logL = (c == A) ? LB_TA : LA_TB;  // re-tempered value of the partner's likelihood
                    s_me.logPost = s_me.logL + s_me.logPr;
                }
                s_me.phase = 0;
                s_me.resolved = 1;
                s_me.swapped = swapped ? 1 : 0;
                if (c == A && j == 0) {  // several pairs (of different iterations) may resolve in one round
                    atomicAdd((unsigned long long *)&a.counters[2], 1ull);
                    if (swapped) atomicAdd((unsigned long long *)&a.counters[3], 1ull);
                }
            }
        }
        __syncthreads();
    }
    // ---- (2) how many candidates the chain proposes now: none while it waits; none past the end of the stretch; none past a
    //          swap step it takes part in
    if (tid == 0) {
        int n = 0;
        if (s_me.phase == 0) {
            for (int k = 0; k < D; k++) {
                const long i = s_me.done + k;
                if (i >= it_b) break;
                n++;
                if (in_swap_pair(a, c, i)) break;
            }
        }
        s_me.nprop_new = n;
    }
    __syncthreads();
    // ---- (3) the chain's state -> LDS (and, by the chain's first workgroup, -> the other parity buffer + records)
    const double *curv = a.vars_cur + (size_t)P * C * Nv, *curp = a.params_cur + (size_t)P * C * Np;
    const double *cndv = a.vars_prop + (size_t)P * CD * Nv, *cndp = a.params_prop + (size_t)P * CD * Np;
    const int sc = s_me.src_chain, sk = s_me.src_cand;
    const double *sv = (sk < 0) ? curv + (size_t)sc * Nv : cndv + ((size_t)sc * D + sk) * Nv;
    const double *sp = (sk < 0) ? curp + (size_t)sc * Np : cndp + ((size_t)sc * D + sk) * Np;
    for (int i = tid; i < Nv; i += TB) s_vars[i] = sv[i];
    for (int i = tid; i < Np; i += TB) s_params[i] = sp[i];
    __syncthreads();
    if (j == 0) {
        double *newv = a.vars_cur + (size_t)Q * C * Nv + (size_t)c * Nv, *newp = a.params_cur + (size_t)Q * C * Np + (size_t)c * Np;
        for (int i = tid; i < Nv; i += TB) newv[i] = s_vars[i];
        for (int i = tid; i < Np; i += TB) newp[i] = s_params[i];
        // records (update_buffer_params / update_buffer_stat_criteria, MALA.cpp:708-710): rejected iterations hold the old
        // state; the last tested iteration holds the new one when accepted; a swap step is recorded once the swap is resolved
        const long d0 = s_me.done - s_me.n_rec;
        const int deferred_now = s_me.swap_last;                        // the last tested iteration waits for the swap
        const int last_new = (s_me.n_rec > 0 && (s_me.acc_last || deferred_now)) ? 1 : 0;
        const int n_old = s_me.n_rec - last_new;
        // final-state record: an accepted (non-swap) last iteration, or the swap step resolved in this round
        const int rec_final = (s_me.n_rec > 0 && s_me.acc_last && !deferred_now) || s_me.resolved;
        if (a.samples) {
            const double *ov = curv + (size_t)c * Nv;
            for (int t = 0; t < n_old; t++)
                for (int i = tid; i < Nv; i += TB) a.samples[((size_t)(d0 + t - rec_base) * C + c) * Nv + i] = ov[i];
            if (rec_final)
                for (int i = tid; i < Nv; i += TB) a.samples[((size_t)(s_me.done - 1 - rec_base) * C + c) * Nv + i] = s_vars[i];
        }
        if (tid == 0) {
            if (a.stats) {
                for (int t = 0; t < n_old; t++) {
                    double *r = a.stats + ((size_t)(d0 + t - rec_base) * C + c) * 3;
                    r[0] = s_me.logL0; r[1] = s_me.logPr0; r[2] = s_me.logPost0;
                }
                if (rec_final) {
                    double *r = a.stats + ((size_t)(s_me.done - 1 - rec_base) * C + c) * 3;
                    r[0] = s_me.logL; r[1] = s_me.logPr; r[2] = s_me.logPost;
                }
            }
            a.logL_cur[Q * C + c] = s_me.logL;
            a.logPr_cur[Q * C + c] = s_me.logPr;
            a.logPost_cur[Q * C + c] = s_me.logPost;
            a.sp_done[Q * C + c] = s_me.done;
            a.sp_nprop[Q * C + c] = s_me.nprop_new;
            a.sp_phase[Q * C + c] = s_me.phase;
            if (s_me.n_rec > 0) {
                a.moved[c] = s_me.acc_last;
                a.Pmove[c] = s_me.r_last;
                if (c == 0 && s_me.acc_last) a.counters[1] += 1;
            }
            atomicAdd((unsigned long long *)&a.counters[4], (unsigned long long)s_me.nprop_new);  // candidates handed to the likelihood kernel
        }
    }
    // ---- (4) candidate j: the proposal of iteration done+j, or an empty slot
    const int slot = c * D + j;
    if (j < s_me.nprop_new) {
        propose_common(a, U, c, s_me.done + j, slot, a.vars_prop + ((size_t)Q * CD + slot) * Nv, a.params_prop + ((size_t)Q * CD + slot) * Np,
                       a.logPr_prop + (size_t)Q * CD + slot, a.status_prop + (size_t)Q * CD + slot, s_vars, s_params, s_z,
                       (a.dbg && c == 0 && j == 0) ? a.dbg : nullptr);
    } else if (tid == 0) {
        a.pairs[2 * slot] = slot * a.desc.per;
        a.pairs[2 * slot + 1] = slot * a.desc.per;
        a.nh[slot] = 0;
        a.nn[slot] = 0;  // the likelihood kernel skips the slot
    }
}

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
    bool pre_lz = true;  // spare workgroups compute L z one iteration ahead (TAMCMC_PRE_LZ=0 disables)
    int tile_rot = 0;  // launch-order hint of k_loglike (first near-field tile of chain 0's initial table)
    std::vector<int32_t> h_plength;
    int G = 1;
    hipStream_t gst[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_kb[4], ev_ki[4], ev_fork, ev_join[4];
    bool ev_made = false;
    double *d_pack = nullptr, *h_pack = nullptr;  // state download: device gather block and its pinned host image

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
    }
    for (void *p : impl->allocs) (void)hipFree(p);
    if (impl->h_pack) (void)hipHostFree(impl->h_pack);
    for (int i = 0; i < impl->n_ev; i++) { (void)hipEventDestroy(impl->ev[i][0]); (void)hipEventDestroy(impl->ev[i][1]); }
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
    a.desc.per = mt::count_multiplets(in.model_id, in.plength);
    I.h_plength.assign(in.plength, in.plength + 11);
    if (a.desc.per < 0) return TAMCMC_ERR_BAD_MODEL;
    a.desc.stride = in.plength[8] > 0 ? in.plength[8] : 1;
    if ((a.desc.stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    a.desc.Nx = (int)c->Nx;
    a.desc.x_first = c->hx[0]; a.desc.x_last = c->hx[(size_t)c->Nx - 1]; a.desc.step = c->hx[1] - c->hx[0];
    a.pl = (long)in.likelihood_params;
    a.seed = in.seed; a.dN_mixing = in.dN_mixing;
    a.c0 = in.c0; a.epsilon1 = in.epsilon1; a.epsi2 = in.epsi2; a.A1 = in.A1; a.target_acceptance = in.target_acceptance;
    const size_t C = (size_t)in.C, Np = (size_t)in.Np, Nv = (size_t)in.Nv;
    {  // candidate slots per chain of the speculative rounds (1 = one iteration per round everywhere)
        const char *ed = getenv("TAMCMC_SPEC_DEPTH");
        int D = ed ? atoi(ed) : 1;  // measured on MI355X (C3, 20 chains): extra candidates cost ~1 us each, more than they save
        if (D < 1) D = 1;
        if (D > TB / 64) D = TB / 64;  // spec_sums: one wave per candidate
        a.D = D;
    }
    const size_t CD = C * (size_t)a.D;
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
    DCHK(I.dalloc(&a.sp_done, 2 * C)); DCHK(I.dalloc(&a.sp_nprop, 2 * C)); DCHK(I.dalloc(&a.sp_phase, 2 * C));
    DCHK(I.dalloc(&a.Pmove, C)); DCHK(I.dalloc(&a.moved, C)); DCHK(I.dalloc(&a.counters, 8));
    a.dbg = nullptr;
    if (const char *es = getenv("TAMCMC_DEBUG_STAMPS")) {
        // 1: phase stamps of one workgroup per kernel; 2: also a (start, end) pair of EVERY k_loglike workgroup of chain group 0
        const size_t n = 16 + 8 + 2 * CD * 4096;
        DCHK(I.dalloc(&a.dbg, n));
        DCHK(hipMemsetAsync(a.dbg, 0, n * sizeof(long), st));
        if (atoi(es) == 2) { const long magic = 77; DCHK(hipMemcpyAsync(a.dbg + 8 + 7, &magic, sizeof(long), hipMemcpyHostToDevice, st)); }
    }
    DCHK(I.dalloc(&a.lz, 2 * C * Nv)); DCHK(I.dalloc(&a.LT, C * Nv * Nv)); DCHK(I.dalloc(&a.cov, C * Nv * Nv)); DCHK(I.dalloc(&a.mu, C * Nv)); DCHK(I.dalloc(&a.sigma, C));
    DCHK(I.dalloc(&a.mults, CD * (size_t)a.desc.per + 1)); DCHK(I.dalloc(&a.pairs, 2 * CD)); DCHK(I.dalloc(&a.nh, CD)); DCHK(I.dalloc(&a.nn, CD));
    DCHK(I.dalloc(&a.noise, CD * (size_t)a.desc.stride));
    DCHK(hipMemsetAsync(a.counters, 0, 8 * sizeof(long), st));
    DCHK(hipMemsetAsync(a.moved, 0, C * sizeof(int), st));
    DCHK(hipMemsetAsync(a.Pmove, 0, C * sizeof(double), st));
    a.samples = nullptr; a.stats = nullptr;
    // Cholesky workspace: LDS when (Nv^2 + Nv) doubles fit beside the iteration's own LDS, else global scratch
    I.lds_base = (Np + 2 * Nv + 1) * sizeof(double) + unpack_lds_bytes() + 32;
    I.lds_adapt = (Nv * Nv + Nv) * sizeof(double);
    a.chol_in_lds = (I.lds_base + I.lds_adapt <= 156 * 1024) ? 1 : 0;
    if (!a.chol_in_lds) { DCHK(I.dalloc(&I.adapt_scratch, C * (Nv * Nv + Nv))); I.lds_adapt = 0; }
    if (I.lds_base + I.lds_adapt > 64 * 1024) {
        DCHK(hipFuncSetAttribute((const void *)k_iterate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(I.lds_base + I.lds_adapt)));
        DCHK(hipFuncSetAttribute((const void *)k_iterate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(I.lds_base + I.lds_adapt)));
    }
    if (I.lds_base > 64 * 1024) DCHK(hipFuncSetAttribute((const void *)k_spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)I.lds_base));
    // polynomial tables Pslm/Qlm: computed ON the device (its own double arithmetic), read through a uniform pointer
    mt::PolyTab *d_tab;
    DCHK(I.dalloc(&d_tab, 1));
    hipLaunchKernelGGL(k_fill_poly, dim3(1), dim3(64), 0, st, d_tab);
    DCHK(hipGetLastError());
    a.desc.poly = d_tab;
    for (int i = 0; i < 64; i++) { DCHK(hipEventCreate(&I.ev[i][0])); DCHK(hipEventCreate(&I.ev[i][1])); I.n_ev = i + 1; }
    {
        if (const char *ep = getenv("TAMCMC_PRE_LZ")) I.pre_lz = atoi(ep) != 0;
        const char *eg = getenv("TAMCMC_CHAIN_GROUPS");
        int G = in.chain_groups > 0 ? in.chain_groups : eg ? atoi(eg) : (in.C >= 8 ? 2 : 1);
        if (G < 1) G = 1;
        if (G > 4) G = 4;
        if (G > in.C) G = in.C;
        I.G = G;
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
    DCHK(hipStreamSynchronize(st));
    return TAMCMC_OK;
}

int DevSampler::upload_state(const double *vars, const double *params, const double *logL, const double *logPr,
                             const double *logPost, const double *init_logL) {
    Impl &I = *impl;
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
        if (build_mode_table(a.desc.model_id, params, I.h_plength.data(), c->hx.data(), c->Nx, tab.data(), a.desc.per, &n, nz.data(), &nh, &nn) == TAMCMC_OK && n <= a.desc.per)
            I.tile_rot = pick_tile_rot(tab.data(), n, a.desc.x_first, a.desc.step, tb, (int)((c->Nx + tb - 1) / tb));
    }
    return TAMCMC_OK;
}

int DevSampler::upload_proposal(int m, const double *L_rowmajor, const double *cov, const double *mu, double sigma) {
    Impl &I = *impl;
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
// moved C (as double) | counters 4 (as double: exact below 2^53)]
__global__ void __launch_bounds__(256) k_pack_state(const DevSamplerArgs a, const int P, double *out) {
    const size_t C = (size_t)a.C, Np = (size_t)a.desc.Np, Nv = (size_t)a.Nv;
    const size_t n_v = C * Nv, n_p = C * Np, total = n_v + n_p + 5 * C + 4;
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
            else v = (double)a.counters[r - 5 * C];
        }
        out[i] = v;
    }
}

// One small kernel + ONE copy into pinned memory (eight copies into pageable memory cost ~150 us per tamcmc_sampler_run call).
int DevSampler::download_state(double *vars, double *params, double *logL, double *logPr, double *logPost, double *Pmove,
                               int *moved, long *counters) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t C = (size_t)a.C, Np = (size_t)a.desc.Np, Nv = (size_t)a.Nv;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    const size_t n_v = C * Nv, n_p = C * Np, total = n_v + n_p + 5 * C + 4;
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
// Stretches without adaptation run as speculative rounds (k_spec); the others one iteration per round (k_iterate).
int DevSampler::run(long it0, long n_iter, const char *learn, double *samples, double *stats) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    if (n_iter <= 0) return TAMCMC_OK;
    DCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t C = (size_t)a.C, Nv = (size_t)a.Nv;
    const int D = a.D;
    const int tb = tile_bins(c->wgs, c->K);
    a.ntiles = (a.desc.Nx + tb - 1) / tb;
    DCHK(c->d_part.reserve(C * (size_t)D * (size_t)a.ntiles * 2));
    a.partials = c->d_part.p;
    a.tile_bins = tb;
    a.bg = nullptr;
    if (c->precision == TAMCMC_PRECISION_FAST) {
        DCHK(c->d_bg.reserve(C * (size_t)D * (size_t)a.ntiles * 8));
        a.bg = c->d_bg.p;
    }
    if (samples && I.smp_cap < (size_t)n_iter * C * Nv) {
        DCHK(I.dalloc(&a.samples, (size_t)n_iter * C * Nv));  // (older, smaller buffers are released with the sampler)
        I.smp_cap = (size_t)n_iter * C * Nv;
    }
    if (stats && I.stat_cap < (size_t)n_iter * C * 3) {
        DCHK(I.dalloc(&a.stats, (size_t)n_iter * C * 3));
        I.stat_cap = (size_t)n_iter * C * 3;
    }
    DevSamplerArgs args = a;
    if (!samples) args.samples = nullptr;
    if (!stats) args.stats = nullptr;
    // chain groups [goff[g], goff[g+1])
    const int G = I.G;
    int goff[5];
    for (int g = 0; g <= G; g++) goff[g] = (int)(((long)a.C * g) / G);
    auto group_of = [&](int chain) { int g = 0; while (g + 1 < G && chain >= goff[g + 1]) g++; return g; };
    LoglikeArgs la[4], las;
    auto fill_la = [&](LoglikeArgs &l, int first_slot, int nslots, bool stamps) {
        l.x = c->dx.p; l.y = c->dy.p; l.logx = c->dlogx.p; l.Nx = a.desc.Nx; l.B = nslots; l.ntiles = a.ntiles;
        l.x0 = a.desc.x_first; l.step = a.desc.step;
        l.mults = a.mults; l.offsets = a.pairs + 2 * first_slot; l.noise = a.noise + (size_t)first_slot * a.desc.stride; l.noise_stride = a.desc.stride;
        l.nharvey = a.nh + first_slot; l.nnoise = a.nn + first_slot; l.partials = a.partials + (size_t)first_slot * a.ntiles * 2; l.model = nullptr;
        l.dbg = (a.dbg && stamps) ? a.dbg + 8 : nullptr;
        l.tile_rot = I.tile_rot;
        l.bg_poly = a.bg ? a.bg + (size_t)first_slot * a.ntiles * 8 : nullptr;
    };
    for (int g = 0; g < G; g++) fill_la(la[g], goff[g], goff[g + 1] - goff[g], g == 0);
    fill_la(las, 0, a.C * D, true);

    int used_ev = 0;
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

    // ---- one iteration per round over [ia, ib): k_iterate settles iteration it-1 and proposes iteration it
    auto lockstep = [&](long ia, long ib) -> int {
        // the extra streams start after everything already enqueued on the context stream
        DCHK(hipEventRecord(I.ev_fork, st));
        for (int g = 1; g < G; g++) DCHK(hipStreamWaitEvent(I.gst[g], I.ev_fork, 0));
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
                                       I.adapt_scratch, goff[g], cnt, pre_flags);
                else  // settle the last iteration of this stretch (MH test, swap, record, adaptation); nothing is proposed
                    hipLaunchKernelGGL(k_iterate<false>, dim3(cnt), dim3(TB), lds, I.gst[g], args, it, P, pending, rec, learn_p, I.adapt_scratch,
                                       goff[g], cnt, 0);
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
            if (i < ib) {
                for (int g = 0; g < G; g++) {
                    const bool timed = g == 0 && c->timing && ((i - ia) % ev_every == 0) && used_ev < I.n_ev;
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
        DCHK(hipStreamSynchronize(st));
        // (with chain groups every launch carries C/G evaluations and overlaps the other groups' kernels)
        return drain_events((double)len * G, len * (long)a.C);
    };

    // ---- speculative rounds over [ia, ib) (no adaptation inside): every chain ends having completed iteration ib-1
    auto speculative = [&](long ia, long ib) -> int {
        const long it_a = it0 + ia, it_b = it0 + ib;
        std::vector<long> h_done(C);
        std::vector<int> h_np(C), h_ph(C);
        long cand0 = 0, cand1 = 0;
        DCHK(hipMemcpyAsync(&cand0, a.counters + 4, sizeof(long), hipMemcpyDeviceToHost, st));
        int first = 1;
        long remaining = ib - ia, rounds_total = 0;
        // expected iterations per round at the target acceptance rate: 1 + q + q^2 + ... (q = rejection probability)
        double rate = 0, q = 1.0;
        for (int k = 0; k < D; k++) { rate += q; q *= (1.0 - a.target_acceptance); }
        for (int batch = 0; batch < 1000000; batch++) {
            long rounds = (long)((double)remaining / rate) + 2;
            const long ev_every = rounds > 16 ? rounds / 16 : 1;
            for (long r = 0; r < rounds; r++) {
                hipLaunchKernelGGL(k_spec, dim3(a.C * D), dim3(TB), I.lds_base, st, args, P, first, it_a, it_b, it0);
                first = 0;
                P ^= 1;
                const bool timed = c->timing && (r % ev_every == 0) && used_ev < I.n_ev;
                if (timed) DCHK(hipEventRecord(I.ev[used_ev][0], st));
                DCHK(launch_loglike(las, c->precision, c->wgs, c->K, false, st));
                if (timed) { DCHK(hipEventRecord(I.ev[used_ev][1], st)); used_ev++; }
            }
            rounds_total += rounds;
            DCHK(hipMemcpyAsync(h_done.data(), a.sp_done + (size_t)P * C, C * sizeof(long), hipMemcpyDeviceToHost, st));
            DCHK(hipMemcpyAsync(h_np.data(), a.sp_nprop + (size_t)P * C, C * sizeof(int), hipMemcpyDeviceToHost, st));
            DCHK(hipMemcpyAsync(h_ph.data(), a.sp_phase + (size_t)P * C, C * sizeof(int), hipMemcpyDeviceToHost, st));
            DCHK(hipMemcpyAsync(&cand1, a.counters + 4, sizeof(long), hipMemcpyDeviceToHost, st));
            DCHK(hipStreamSynchronize(st));
            DCHK(hipGetLastError());
            int rc = drain_events((double)rounds, 0);
            if (rc) return rc;
            long min_done = it_b;
            bool settled = true;
            for (size_t m = 0; m < C; m++) {
                if (h_done[m] < min_done) min_done = h_done[m];
                if (h_np[m] != 0 || h_ph[m] != 0) settled = false;
            }
            if (min_done >= it_b && settled) break;
            remaining = it_b - min_done;
            if (remaining < 1) remaining = 1;
        }
        n_eval += cand1 - cand0;
        (void)rounds_total;
        return TAMCMC_OK;
    };

    // ---- split [0, n_iter) into stretches: quiet ones (no adaptation, at least MIN_SPEC long) run speculatively
    const long MIN_SPEC = 8;  // shorter quiet stretches are not worth the drain rounds
    auto quiet_end = [&](long from) { long q2 = from; while (q2 < n_iter && !(learn && learn[q2])) q2++; return q2; };
    long i = 0;
    while (i < n_iter) {
        const long jn = quiet_end(i);
        if (D > 1 && jn - i >= MIN_SPEC) {
            int rc = speculative(i, jn);
            if (rc) return rc;
            i = jn;
            continue;
        }
        long k = i;  // one iteration per round up to the start of the next long quiet stretch
        for (;;) {
            const long q2 = quiet_end(k);
            if (D > 1 && q2 - k >= MIN_SPEC && k > i) break;
            k = q2;
            while (k < n_iter && learn && learn[k]) k++;
            if (k >= n_iter) break;
        }
        int rc = lockstep(i, k);
        if (rc) return rc;
        i = k;
    }
    {  // iteration counter as the one-iteration engine leaves it
        const long itn = it0 + n_iter;
        DCHK(hipMemcpyAsync(a.counters, &itn, sizeof(long), hipMemcpyHostToDevice, st));
    }
    I.parity = P;
    if (a.dbg) {  // phase stamps of the last proposal workgroup of chain 0 / the middle tile (100 MHz wall clock)
        long h[16];
        DCHK(hipMemcpyAsync(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost, st));
        DCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[k_iterate stamps us] settle %.2f | rng+matvec %.2f | scatter+prior %.2f (constraints %.2f) | unpack %.2f (visibilities %.2f, rows %.2f)\n",
                (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[4] - h[2]) * 0.01, (h[3] - h[2]) * 0.01, (h[6] - h[4]) * 0.01, (h[5] - h[4]) * 0.01,
                (h[7] - h[5]) * 0.01);
        const long *k = h + 8;
        if (k[7] == 77 && a.ntiles <= 4096) {  // timeline of the last k_loglike launch of chain group 0
            const int nb = (G > 1 ? goff[1] : a.C), nw = nb * a.ntiles;
            std::vector<long> w(2 * (size_t)nw);
            DCHK(hipMemcpy(w.data(), a.dbg + 16, w.size() * sizeof(long), hipMemcpyDeviceToHost));
            long t0 = w[0], t1 = w[1];
            for (int q = 0; q < nw; q++) { if (w[2 * q] && w[2 * q] < t0) t0 = w[2 * q]; if (w[2 * q + 1] > t1) t1 = w[2 * q + 1]; }
            std::vector<double> dur, start;
            for (int q = 0; q < nw; q++) if (w[2 * q]) { dur.push_back((w[2 * q + 1] - w[2 * q]) * 0.01); start.push_back((w[2 * q] - t0) * 0.01); }
            std::sort(dur.begin(), dur.end()); std::sort(start.begin(), start.end());
            auto pct = [](const std::vector<double> &v, double p) { return v.empty() ? 0.0 : v[(size_t)(p * (v.size() - 1))]; };
            fprintf(stderr, "[k_loglike timeline us] %zu workgroups, first start -> last end %.2f | start p50 %.2f p90 %.2f max %.2f | duration p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
                    dur.size(), (t1 - t0) * 0.01, pct(start, .5), pct(start, .9), pct(start, 1.), pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1.));
        }
        fprintf(stderr, "[k_loglike stamps us, middle tile] prologue %.2f | staging %.2f | near %.2f | far+reduce %.2f | horner %.2f | epilogue %.2f\n",
                (k[1] - k[0]) * 0.01, (k[2] - k[1]) * 0.01, (k[3] - k[2]) * 0.01, (k[4] - k[3]) * 0.01, (k[5] - k[4]) * 0.01, (k[6] - k[5]) * 0.01);
    }
    DCHK(hipGetLastError());
    if (samples) DCHK(hipMemcpyAsync(samples, a.samples, (size_t)n_iter * C * Nv * 8, hipMemcpyDeviceToHost, st));
    if (stats) DCHK(hipMemcpyAsync(stats, a.stats, (size_t)n_iter * C * 3 * 8, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    c->kernel_ms += kernel_ms;
    c->launches += n_launch;
    c->evals += n_eval;
    return TAMCMC_OK;
}

}  // namespace tamcmc
