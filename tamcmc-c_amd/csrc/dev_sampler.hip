// dev_sampler.hip -- device-resident MCMC iteration (sampler-side algebra on the device).
//
// The host-driven loop (host_mala.cpp) spends ~3/4 of a step on the host (proposal, priors, table build, copies,
// one sync per step).  Here one MCMC iteration of ALL tempered chains is three kernels on the context's stream,
// with no host round trip and no copy in between:
//   k_propose_unpack  (one workgroup per chain)  z ~ N(0,I) (Philox, same streams as the host engine),
//                     x' = x + L z, params', log-prior (terms in parallel), params' -> multiplet table + noise row
//                     written straight into the likelihood kernel's input block
//                     (MALA.cpp:339-369 new_prop_values, model_def.cpp:484-492, priors_calc.cpp, models.cpp unpackers)
//   k_loglike         (kernels.hip)               the hot kernel, unchanged
//   k_accept_swap     (one workgroup)             per-chain partial sums -> tempered logL, MH accept (MALA.cpp:490-551),
//                     adjacent-pair parallel-tempering swap (MALA.cpp:397-461), sample/stat record (outputs.cpp buffers)
//   k_adapt           (one workgroup per chain, learning phases only) Robbins-Monro update of mu, Sigma, sigma
//                     (MALA.cpp:296-319) and Cholesky of (Sigma+eps2 I) sigma (MALA.cpp:348-350)
// The host only enqueues launches and fetches the recorded samples once per run() call.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"
#include "dev_sampler.h"
#include "kernels.h"
#include "mode_tables_impl.h"
#include "priors_impl.h"
#include "rng.h"

namespace tamcmc {

namespace {

__global__ void k_fill_poly(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
}

constexpr int PB = 128;  // threads of k_propose_unpack

__device__ __forceinline__ double block_sum(double v, double *s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double s = s_red[0];
    for (int w = 1; w < nw; w++) s = s + s_red[w];
    return s;
}

__global__ void __launch_bounds__(PB) k_propose_unpack(const DevSamplerArgs a, const long it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double *s_params = (double *)s_raw;          // [Np]
    double *s_vars = s_params + a.Np;            // [Nv]
    double *s_z = s_vars + a.Nv;                 // [Nv + 1]
    double *s_red = s_z + a.Nv + 1;              // [4]
    mt::Shared *S = (mt::Shared *)(s_red + 4);   // shared scalars of the unpack
    __shared__ int s_status;
    __shared__ double s_c, s_dnu;

    const int m = blockIdx.x, tid = threadIdx.x;
    const int Np = a.Np, Nv = a.Nv;
    // ---- proposal: x' = x + L z (MALA.cpp:348-355), L = chol((Sigma+eps2) sigma), stored transposed ----
    for (int k2 = tid; 2 * k2 < Nv; k2 += PB) {
        double z0, z1;
        rng_normal2(a.seed, RNG_PROPOSAL, (uint32_t)m, (uint64_t)it, (uint32_t)k2, z0, z1);
        s_z[2 * k2] = z0;
        s_z[2 * k2 + 1] = z1;
    }
    for (int i = tid; i < Np; i += PB) s_params[i] = a.params_cur[(size_t)m * Np + i];
    if (tid == 0) s_status = TAMCMC_OK;
    __syncthreads();
    const double *LT = a.LT + (size_t)m * Nv * Nv;
    for (int i = tid; i < Nv; i += PB) {
        double s = 0;
        for (int k = 0; k <= i; k++) s = s + LT[(size_t)k * Nv + i] * s_z[k];
        const double v = a.vars_cur[(size_t)m * Nv + i] + 0.0 + s;
        s_vars[i] = v;
        a.vars_prop[(size_t)m * Nv + i] = v;
    }
    __syncthreads();
    for (int k = tid; k < Nv; k += PB) s_params[a.index_to_relax[k]] = s_vars[k];  // update_params_with_vars
    __syncthreads();
    for (int i = tid; i < Np; i += PB) a.params_prop[(size_t)m * Np + i] = s_params[i];

    // ---- log-prior: hard constraints by one lane, additive terms one per lane, tree-summed ----
    if (tid == 0) {
        int st = TAMCMC_OK;
        mt::xreal c;
        if (a.prior_class == 2) {
            c = pr::ms_global_constraints(s_params, a.plength, a.priors_switch, a.extra, &st);
            double fit[2];
            mt::linfit_index(s_params + a.plength[0] + a.plength[1], a.plength[2], fit);
            s_dnu = fit[0];
        } else if (a.prior_class == 3) {
            c = pr::local_constraints(s_params, a.plength, a.priors_switch, a.extra);
        } else {
            c = pr::neg_inf();
            st = TAMCMC_ERR_BAD_MODEL;
        }
        s_c = c;
        if (st != TAMCMC_OK) s_status = st;
    }
    __syncthreads();
    double logPr;
    {
        const int n_extra = (a.prior_class == 2) ? pr::ms_global_extra_terms(a.plength, a.extra) : 0;
        double f = 0;
        int st = TAMCMC_OK;
        for (int t = tid; t < Np + n_extra; t += PB) {
            if (t < Np) f = f + pr::generic_prior_term(s_params, Np, a.priors, a.priors_switch, t, &st);
            else f = f + pr::ms_global_extra_term(s_params, a.plength, a.extra, s_dnu, t - Np);
        }
        if (st != TAMCMC_OK) s_status = st;
        f = block_sum(f, s_red);
        logPr = (s_c != 0) ? s_c : f;
    }

    // ---- params' -> multiplet table (skipped when the prior is -inf: model_def.cpp:472,476-480) ----
    const int per = a.per;
    const bool live = (logPr != -INFINITY) && !isnan(logPr);
    if (live) {
        if (tid == 0) {
            mt::shared_scalars_base(a.model_id, s_params, a.plength, *S);
        }
        __syncthreads();
        // m-visibilities: one lane per Wigner element d^l_{i,0}, i=0..l, l=1..3 (9 lanes) + the centre elements
        if (tid < 12) {
            int l, i;
            if (tid < 2) { l = 1; i = tid; } else if (tid < 5) { l = 2; i = tid - 2; } else if (tid < 9) { l = 3; i = tid - 5; }
            else { l = tid - 8; i = -1; }
            if (S->need_ratio[l]) {
                const double PI = 3.141592653589793238462643;
                const double ang = PI * S->inc / 180.;
                if (i >= 0) S->ratios[l][l + i] = mt::wigner_d(l, i, 0, ang);
                else S->centre[l] = mt::wigner_d(l, 0, 0, -ang);
            }
        }
        __syncthreads();
        if (tid >= 1 && tid <= 3 && S->need_ratio[tid]) {  // mirror, centre overwrite, square (function_rot.cpp:25-41)
            const int l = tid;
            double *V = S->ratios[l];
            for (int i = -l; i <= 0; i++) V[l + i] = V[l - i] * pow(-1.0, (double)i);
            V[l] = S->centre[l] * pow(-1.0, 0.);
            for (int i = 0; i <= 2 * l; i++) V[i] = V[i] * V[i];
        }
        __syncthreads();
        for (int idx = tid; idx < per; idx += PB) {
            tamcmc_multiplet r;
            const int st = mt::build_multiplet(a.model_id, *(const mt::PolyTab *)a.poly, s_params, *S, idx, a.x_first, a.x_last, a.Nx, a.step, &r);
            if (st) s_status = st;
            else a.mults[(size_t)m * per + idx] = r;
        }
        for (int i = tid; i < S->L.Nnoise; i += PB) a.noise[(size_t)m * a.stride + i] = fabs(s_params[S->L.o_noise + i]);
    }
    __syncthreads();
    if (tid == 0) {
        const bool ok = live && (s_status == TAMCMC_OK);
        a.pairs[2 * m] = m * per;
        a.pairs[2 * m + 1] = ok ? (m + 1) * per : m * per;
        a.nh[m] = ok ? S->nharvey : 0;
        a.nn[m] = ok ? S->L.Nnoise : 1;
        if (!ok) a.noise[(size_t)m * a.stride] = 1.0;  // placeholder row; the chain is rejected in k_accept_swap
        a.logPr_prop[m] = logPr;
        a.status_prop[m] = s_status;
    }
}

// One workgroup: finalize + accept + swap + record for every chain.
__global__ void __launch_bounds__(256) k_accept_swap(const DevSamplerArgs a, const long it, const long rec) {
    __shared__ int s_acc[TAMCMC_MAX_CHAINS];
    __shared__ int s_swapA;
    __shared__ double s_swapvals[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, Nv = a.Nv, Np = a.Np;
    // ---- per chain: partial sums -> tempered logL; Metropolis-Hastings test (MALA.cpp:490-551) ----
    for (int m = wave; m < C; m += 4) {
        double s1 = 0, s2 = 0;
        for (int t = lane; t < a.ntiles; t += 64) {
            const double *p = a.partials + ((size_t)m * a.ntiles + t) * 2;
            s1 = s1 + p[0];
            s2 = s2 + p[1];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s1 = s1 + __shfl_down(s1, off, 64);
            s2 = s2 + __shfl_down(s2, off, 64);
        }
        if (lane == 0) {
            const double S = s1 + s2;
            double logL = (-(double)a.pl * S) / a.Tcoefs[m];  // call_likelihood, model_def.cpp:399-401
            const double logPr = a.logPr_prop[m];
            double logPost;
            if (a.status_prop[m] != TAMCMC_OK) logL = NAN;
            if (logPr == -INFINITY || isnan(logPr)) { logL = a.init_logL[m]; logPost = -INFINITY; }
            else logPost = logL + logPr;
            double u, u1;
            rng_uniform2(a.seed, RNG_ACCEPT, (uint32_t)m, (uint64_t)it, 0, u, u1);
            double r;
            if (!isnan(logL)) {
                if (logPost == -INFINITY) r = 0.;
                else {
                    const double e = exp(logPost - a.logPost_cur[m]);
                    r = fmin(1.0, e);
                    if (isnan(r)) r = 0.;
                }
            } else r = 0.;
            const int acc = (u <= r) ? 1 : 0;
            s_acc[m] = acc;
            if (acc) {
                a.logL_cur[m] = logL;
                a.logPr_cur[m] = logPr;
                a.logPost_cur[m] = logPost;
            }
            a.moved[m] = acc;
            a.Pmove[m] = r;
            if (m == 0 && acc) a.counters[1] += 1;
        }
    }
    __syncthreads();
    for (int m = 0; m < C; m++) {
        if (s_acc[m]) {
            for (int i = tid; i < Nv; i += 256) a.vars_cur[(size_t)m * Nv + i] = a.vars_prop[(size_t)m * Nv + i];
            for (int i = tid; i < Np; i += 256) a.params_cur[(size_t)m * Np + i] = a.params_prop[(size_t)m * Np + i];
        }
    }
    __syncthreads();
    // ---- parallel tempering: adjacent pair, tempered log-likelihoods (MALA.cpp:397-461) ----
    const bool do_swap = a.dN_mixing > 0 && (it % a.dN_mixing == 0) && it != 0 && C > 1;
    if (tid == 0) {
        s_swapA = -1;
        if (do_swap) {
            double u, u2;
            rng_uniform2(a.seed, RNG_SWAP, 0, (uint64_t)it, 0, u, u2);
            int A = (int)(u2 * (double)(C - 1));
            if (A > C - 2) A = C - 2;
            const int B = A + 1;
            const double LA = a.logL_cur[A], LB = a.logL_cur[B];
            const double LA_TB = LA * a.Tcoefs[A] / a.Tcoefs[B];
            const double LB_TA = LB * a.Tcoefs[B] / a.Tcoefs[A];
            const double e = exp(LA_TB + LB_TA - LA - LB);
            const double rT = fmin(1.0, e);
            a.counters[2] += 1;
            if (u <= rT) {
                s_swapA = A;
                s_swapvals[0] = LB_TA;
                s_swapvals[1] = LA_TB;
                a.counters[3] += 1;
            }
        }
    }
    __syncthreads();
    if (s_swapA >= 0) {
        const int A = s_swapA, B = A + 1;
        for (int i = tid; i < Nv; i += 256) {
            const double t = a.vars_cur[(size_t)A * Nv + i];
            a.vars_cur[(size_t)A * Nv + i] = a.vars_cur[(size_t)B * Nv + i];
            a.vars_cur[(size_t)B * Nv + i] = t;
        }
        for (int i = tid; i < Np; i += 256) {
            const double t = a.params_cur[(size_t)A * Np + i];
            a.params_cur[(size_t)A * Np + i] = a.params_cur[(size_t)B * Np + i];
            a.params_cur[(size_t)B * Np + i] = t;
        }
        if (tid == 0) {
            const double prA = a.logPr_cur[A], prB = a.logPr_cur[B];
            a.logL_cur[A] = s_swapvals[0];
            a.logPr_cur[A] = prB;
            a.logPost_cur[A] = s_swapvals[0] + prB;
            a.logL_cur[B] = s_swapvals[1];
            a.logPr_cur[B] = prA;
            a.logPost_cur[B] = s_swapvals[1] + prA;
            const int mv = a.moved[A]; a.moved[A] = a.moved[B]; a.moved[B] = mv;
            const double pm = a.Pmove[A]; a.Pmove[A] = a.Pmove[B]; a.Pmove[B] = pm;
        }
    }
    __syncthreads();
    // ---- record (update_buffer_params / update_buffer_stat_criteria, MALA.cpp:708-710) ----
    if (a.samples && rec >= 0)
        for (int i = tid; i < C * Nv; i += 256) a.samples[(size_t)rec * C * Nv + i] = a.vars_cur[i];
    if (a.stats && rec >= 0)
        for (int m = tid; m < C; m += 256) {
            double *r = a.stats + ((size_t)rec * C + m) * 3;
            r[0] = a.logL_cur[m];
            r[1] = a.logPr_cur[m];
            r[2] = a.logPost_cur[m];
        }
    if (tid == 0) a.counters[0] = it + 1;
}

// Robbins-Monro adaptation + Cholesky, one workgroup per chain (learning phases only).
__global__ void __launch_bounds__(256) k_adapt(const DevSamplerArgs a, const long it, double *scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_red[4];
    __shared__ double s_scal[2];
    const int m = blockIdx.x, tid = threadIdx.x, Nv = a.Nv;
    double *A = a.chol_in_lds ? (double *)s_raw : scratch + (size_t)m * Nv * Nv;
    double *d = a.chol_in_lds ? (double *)s_raw + (size_t)Nv * Nv : scratch + (size_t)a.C * Nv * Nv + (size_t)m * Nv;
    const double g = a.c0 / (1. + (double)it);
    double *mu = a.mu + (size_t)m * Nv;
    double *cov = a.cov + (size_t)m * Nv * Nv;
    const double *vars = a.vars_cur + (size_t)m * Nv;
    // mu (MALA.cpp:307-308) with the norm clip p3
    double n2 = 0;
    for (int k = tid; k < Nv; k += 256) {
        const double v = mu[k] + g * (vars[k] - mu[k]);
        d[k] = v;
        n2 += v * v;
    }
    n2 = block_sum(n2, s_red);
    {
        const double nrm = sqrt(n2);
        const double sc = (nrm <= a.A1) ? 1.0 : a.A1 / nrm;
        for (int k = tid; k < Nv; k += 256) {
            const double v = (sc == 1.0) ? d[k] : d[k] * sc;
            mu[k] = v;
            d[k] = vars[k] - v;  // deviation from the UPDATED mu (MALA.cpp:311)
        }
    }
    __syncthreads();
    // covariance (MALA.cpp:311-313) with the Frobenius clip p2
    n2 = 0;
    for (int e = tid; e < Nv * Nv; e += 256) {
        const int i = e / Nv, j = e - i * Nv;
        const double v = cov[e] + g * (d[i] * d[j] - cov[e]);
        cov[e] = v;
        n2 += v * v;
    }
    n2 = block_sum(n2, s_red);
    if (tid == 0) {
        const double nrm = sqrt(n2);
        s_scal[0] = (nrm <= a.A1) ? 1.0 : a.A1 / nrm;
        // sigma (MALA.cpp:316-317) with the clip p1
        double v1 = a.sigma[m] + g * (a.Pmove[m] - a.target_acceptance);
        if (v1 < a.epsilon1) v1 = a.epsilon1;
        if (v1 > a.A1) v1 = a.A1;
        a.sigma[m] = v1;
        s_scal[1] = v1;
    }
    __syncthreads();
    const double sc = s_scal[0], sig = s_scal[1];
    for (int e = tid; e < Nv * Nv; e += 256) {
        const int i = e / Nv, j = e - i * Nv;
        double v = cov[e];
        if (sc != 1.0) { v = v * sc; cov[e] = v; }
        A[e] = (v + (i == j ? a.epsi2 : 0.0)) * sig;  // (covarmat + epsilon2) * sigma (MALA.cpp:348)
    }
    __syncthreads();
    // right-looking Cholesky in place (lower triangle of A)
    for (int j = 0; j < Nv; j++) {
        if (tid == 0) A[(size_t)j * Nv + j] = sqrt(A[(size_t)j * Nv + j]);
        __syncthreads();
        const double djj = A[(size_t)j * Nv + j];
        for (int i = j + 1 + tid; i < Nv; i += 256) A[(size_t)i * Nv + j] = A[(size_t)i * Nv + j] / djj;
        __syncthreads();
        const int rem = Nv - j - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = j + 1 + e / rem, k = j + 1 + e % rem;
            if (k <= i) A[(size_t)i * Nv + k] = A[(size_t)i * Nv + k] - A[(size_t)i * Nv + j] * A[(size_t)k * Nv + j];
        }
        __syncthreads();
    }
    double *LT = a.LT + (size_t)m * Nv * Nv;
    for (int e = tid; e < Nv * Nv; e += 256) {
        const int i = e / Nv, k = e - i * Nv;
        LT[(size_t)k * Nv + i] = (k <= i) ? A[e] : 0.0;
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
    size_t lds_propose = 0, lds_adapt = 0;

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
    for (int i = 0; i < impl->n_ev; i++) { (void)hipEventDestroy(impl->ev[i][0]); (void)hipEventDestroy(impl->ev[i][1]); }
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
    a.model_id = in.model_id; a.prior_class = in.prior_class; a.C = in.C; a.Np = in.Np; a.Nv = in.Nv;
    a.per = mt::count_multiplets(in.model_id, in.plength);
    if (a.per < 0) return TAMCMC_ERR_BAD_MODEL;
    a.stride = in.plength[8] > 0 ? in.plength[8] : 1;
    if ((a.stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    a.Nx = (int)c->Nx;
    a.x_first = c->hx[0]; a.x_last = c->hx[(size_t)c->Nx - 1]; a.step = c->hx[1] - c->hx[0];
    a.pl = (long)in.likelihood_params;
    a.seed = in.seed; a.dN_mixing = in.dN_mixing;
    a.c0 = in.c0; a.epsilon1 = in.epsilon1; a.epsi2 = in.epsi2; a.A1 = in.A1; a.target_acceptance = in.target_acceptance;
    const size_t C = (size_t)in.C, Np = (size_t)in.Np, Nv = (size_t)in.Nv;
    hipStream_t st = c->stream;
    int *d_pl, *d_idx, *d_sw;
    double *d_pr, *d_ex, *d_T;
    DCHK(I.dalloc(&d_pl, 11)); DCHK(I.dalloc(&d_idx, Nv)); DCHK(I.dalloc(&d_sw, Np));
    DCHK(I.dalloc(&d_pr, 4 * Np)); DCHK(I.dalloc(&d_ex, 10)); DCHK(I.dalloc(&d_T, C));
    DCHK(up(d_pl, in.plength, 11, st)); DCHK(up(d_idx, in.index_to_relax, Nv, st)); DCHK(up(d_sw, in.priors_switch, Np, st));
    DCHK(up(d_pr, in.priors, 4 * Np, st)); DCHK(up(d_ex, in.extra_priors, 10, st)); DCHK(up(d_T, in.Tcoefs, C, st));
    a.plength = d_pl; a.index_to_relax = d_idx; a.priors_switch = d_sw; a.priors = d_pr; a.extra = d_ex; a.Tcoefs = d_T;
    DCHK(I.dalloc(&a.vars_cur, C * Nv)); DCHK(I.dalloc(&a.params_cur, C * Np));
    DCHK(I.dalloc(&a.vars_prop, C * Nv)); DCHK(I.dalloc(&a.params_prop, C * Np));
    DCHK(I.dalloc(&a.logL_cur, C)); DCHK(I.dalloc(&a.logPr_cur, C)); DCHK(I.dalloc(&a.logPost_cur, C));
    DCHK(I.dalloc(&a.init_logL, C)); DCHK(I.dalloc(&a.logPr_prop, C)); DCHK(I.dalloc(&a.status_prop, C));
    DCHK(I.dalloc(&a.Pmove, C)); DCHK(I.dalloc(&a.moved, C)); DCHK(I.dalloc(&a.counters, 4));
    DCHK(I.dalloc(&a.LT, C * Nv * Nv)); DCHK(I.dalloc(&a.cov, C * Nv * Nv)); DCHK(I.dalloc(&a.mu, C * Nv)); DCHK(I.dalloc(&a.sigma, C));
    DCHK(I.dalloc(&a.mults, C * (size_t)a.per + 1)); DCHK(I.dalloc(&a.pairs, 2 * C)); DCHK(I.dalloc(&a.nh, C)); DCHK(I.dalloc(&a.nn, C));
    DCHK(I.dalloc(&a.noise, C * (size_t)a.stride));
    DCHK(hipMemsetAsync(a.counters, 0, 4 * sizeof(long), st));
    DCHK(hipMemsetAsync(a.moved, 0, C * sizeof(int), st));
    DCHK(hipMemsetAsync(a.Pmove, 0, C * sizeof(double), st));
    a.samples = nullptr; a.stats = nullptr;
    // Cholesky workspace: LDS when (Nv^2 + Nv) doubles fit in 160 KB, else global scratch
    I.lds_adapt = (Nv * Nv + Nv) * sizeof(double);
    a.chol_in_lds = I.lds_adapt <= 150 * 1024 ? 1 : 0;
    if (!a.chol_in_lds) { DCHK(I.dalloc(&I.adapt_scratch, C * Nv * Nv + C * Nv)); I.lds_adapt = 0; }
    I.lds_propose = (Np + 2 * Nv + 1 + 4) * sizeof(double) + sizeof(mt::Shared) + 64;
    if (I.lds_adapt > 64 * 1024)
        DCHK(hipFuncSetAttribute((const void *)k_adapt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)I.lds_adapt));
    // polynomial tables Pslm/Qlm: computed ON the device (its own double arithmetic), read through a uniform pointer
    mt::PolyTab *d_tab;
    DCHK(I.dalloc(&d_tab, 1));
    hipLaunchKernelGGL(k_fill_poly, dim3(1), dim3(64), 0, st, d_tab);
    DCHK(hipGetLastError());
    a.poly = d_tab;
    for (int i = 0; i < 64; i++) { DCHK(hipEventCreate(&I.ev[i][0])); DCHK(hipEventCreate(&I.ev[i][1])); I.n_ev = i + 1; }
    DCHK(hipStreamSynchronize(st));
    return TAMCMC_OK;
}

int DevSampler::upload_state(const double *vars, const double *params, const double *logL, const double *logPr,
                             const double *logPost, const double *init_logL) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t C = (size_t)a.C, Np = (size_t)a.Np, Nv = (size_t)a.Nv;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    DCHK(up(a.vars_cur, vars, C * Nv, st)); DCHK(up(a.params_cur, params, C * Np, st));
    DCHK(up(a.logL_cur, logL, C, st)); DCHK(up(a.logPr_cur, logPr, C, st)); DCHK(up(a.logPost_cur, logPost, C, st));
    DCHK(up(a.init_logL, init_logL, C, st));
    DCHK(hipStreamSynchronize(st));
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

int DevSampler::download_state(double *vars, double *params, double *logL, double *logPr, double *logPost, double *Pmove,
                               int *moved, long *counters) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    const size_t C = (size_t)a.C, Np = (size_t)a.Np, Nv = (size_t)a.Nv;
    hipStream_t st = c->stream;
    DCHK(hipSetDevice(c->device));
    auto down = [&](void *dst, const void *src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); };
    if (vars) DCHK(down(vars, a.vars_cur, C * Nv * 8));
    if (params) DCHK(down(params, a.params_cur, C * Np * 8));
    if (logL) DCHK(down(logL, a.logL_cur, C * 8));
    if (logPr) DCHK(down(logPr, a.logPr_cur, C * 8));
    if (logPost) DCHK(down(logPost, a.logPost_cur, C * 8));
    if (Pmove) DCHK(down(Pmove, a.Pmove, C * 8));
    if (moved) DCHK(down(moved, a.moved, C * sizeof(int)));
    if (counters) DCHK(down(counters, a.counters, 4 * sizeof(long)));
    DCHK(hipStreamSynchronize(st));
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
int DevSampler::run(long it0, long n_iter, const char *learn, double *samples, double *stats) {
    Impl &I = *impl;
    tamcmc_hip_ctx *c = I.ctx;
    DevSamplerArgs &a = I.a;
    if (n_iter <= 0) return TAMCMC_OK;
    DCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t C = (size_t)a.C, Nv = (size_t)a.Nv;
    const int tb = tile_bins(c->K);
    a.ntiles = (a.Nx + tb - 1) / tb;
    DCHK(c->d_part.reserve(C * (size_t)a.ntiles * 2));
    a.partials = c->d_part.p;
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
    LoglikeArgs la;
    la.x = c->dx.p; la.y = c->dy.p; la.logx = c->dlogx.p; la.Nx = a.Nx; la.B = a.C; la.ntiles = a.ntiles;
    la.mults = a.mults; la.offsets = a.pairs; la.noise = a.noise; la.noise_stride = a.stride;
    la.nharvey = a.nh; la.nnoise = a.nn; la.partials = a.partials; la.model = nullptr;
    const bool fast = c->precision == TAMCMC_PRECISION_FAST;
    int used_ev = 0;
    const long ev_every = n_iter > 64 ? n_iter / 64 : 1;
    for (long i = 0; i < n_iter; i++) {
        const long it = it0 + i;
        hipLaunchKernelGGL(k_propose_unpack, dim3(a.C), dim3(PB), I.lds_propose, st, args, it);
        const bool timed = c->timing && (i % ev_every == 0) && used_ev < I.n_ev;
        if (timed) DCHK(hipEventRecord(I.ev[used_ev][0], st));
        DCHK(launch_loglike(la, fast, c->K, false, st));
        if (timed) { DCHK(hipEventRecord(I.ev[used_ev][1], st)); used_ev++; }
        hipLaunchKernelGGL(k_accept_swap, dim3(1), dim3(256), 0, st, args, it, (samples || stats) ? i : (long)-1);
        if (learn && learn[i]) hipLaunchKernelGGL(k_adapt, dim3(a.C), dim3(256), I.lds_adapt, st, args, it, I.adapt_scratch);
    }
    DCHK(hipGetLastError());
    if (samples) DCHK(hipMemcpyAsync(samples, a.samples, (size_t)n_iter * C * Nv * 8, hipMemcpyDeviceToHost, st));
    if (stats) DCHK(hipMemcpyAsync(stats, a.stats, (size_t)n_iter * C * 3 * 8, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    if (used_ev) {
        double tot = 0;
        for (int e = 0; e < used_ev; e++) {
            float ms = 0;
            DCHK(hipEventElapsedTime(&ms, I.ev[e][0], I.ev[e][1]));
            tot += ms;
        }
        // extrapolate the sampled launches to all launches of this run (every launch has the same shape)
        c->kernel_ms += tot / used_ev * (double)n_iter;
        c->launches += n_iter;
        c->evals += n_iter * a.C;
    }
    return TAMCMC_OK;
}

}  // namespace tamcmc
