// dev_mala_impl.h -- the Langevin step on the device-resident engine (included by dev_sampler.hip, inside its anonymous namespace).
//
// use_drift = 1: what the reference leaves as stubs (MALA::D_MALA MALA.cpp:321-328, multinormal_logpdf :330-337, fatal at :496-500),
// same algorithm as the host engine (host_mala.cpp): preconditioned, optionally truncated drift (1/2) sigma (Sigma + eps2) grad with the
// forward-difference gradient of the tempered log-posterior, and the q(x|x') / q(x'|x) correction in the acceptance ratio.
// One iteration = the finite-difference batch of every proposal (fd_batch.hip: its base evaluation IS the proposal's
// generate_model; C x (Nvars+1) evaluations, windowed delta tables) between two small kernels, one workgroup per chain:
//   k_mala_test     gradient at the proposal from the batch's sums and priors -> drift there -> both proposal log-densities (two
//                   triangular solves with the Cholesky factor) -> MH test (MALA.cpp:490-551 with the correction) -> Robbins-Monro
//                   adaptation of the chain's proposal law when the schedule says so (MALA.cpp:296-319);
//   k_mala_settle   parallel-tempering swap on the post-test outcomes (MALA.cpp:397-461; the stored gradients follow the positions,
//                   their likelihood share re-tempered), settled state + records, then the next proposal x' = x + drift + L z.
// Nothing crosses PCIe inside an iteration; the host only enqueues.

struct MalaArgs {
    const double *S, *lpp, *lpm;  // sums / log-priors (forward, backward) of the last finite-difference batch (fd_batch.h)
    const int *st;                // ... and its per-evaluation status
    double *h;                    // [Nv] forward-difference steps of the NEXT batch (written by chain 0's workgroup)
    int E, windowed;
    double fd_step_rel, delta;
    double *grad_prop, *gradP_prop, *drift_cur;  // [C][Nv] gradient (and the prior's share) at the proposals; drift used by the proposals
    double *out;                  // [C][5] post-test outcome: acc, r, logL, logPr, logPost
};

// Gradient of the tempered log-posterior at the batch's base point of chain c (compute_gradients, host_mala.cpp; assembly of fd_run,
// fd_batch.hip): lane k -> g[k], gp[k] (the prior's share) in LDS.  base = the base point's parameter vector.  L0 / pr0 / st0 of the base
// evaluation are returned to every lane.
__device__ void mala_gradient(const DevSamplerArgs &a, const MalaArgs &M, int c, const double *base, double *g, double *gp, double &L0,
                              double &pr0, int &st0) {
    const int Nv = a.Nv, E = M.E, C = a.C;
    const double T = a.Tcoefs[c];
    auto scaled = [&](double S) { return (-(double)a.pl * S) / T; };  // call_likelihood, model_def.cpp:399-401
    st0 = M.st[(size_t)c * E];
    L0 = (st0 != TAMCMC_OK) ? (double)NAN : scaled(M.windowed ? M.S[c] : M.S[(size_t)c * E]);
    pr0 = M.lpp[(size_t)c * E];
    for (int k = threadIdx.x; k < Nv; k += blockDim.x) {
        const size_t e = (size_t)c * E + k + 1;
        const double x0 = base[a.index_to_relax[k]];
        volatile double xp = x0 + M.h[k];
        const double happ = xp - x0;  // the step actually applied (k_fd_unpack adds the same two doubles)
        double dl;
        if (M.st[e] != TAMCMC_OK || st0 != TAMCMC_OK) dl = NAN;
        else dl = M.windowed ? scaled(M.S[(size_t)C + e]) : scaled(M.S[e]) - L0;
        double gv = dl / happ;
        if (!isfinite(gv)) gv = 0.0;
        const double prp = M.lpp[e], prm = M.lpm[e];
        double gpv;
        if (isfinite(prp)) gpv = (prp - pr0) / happ;
        else gpv = isfinite(prm) ? (pr0 - prm) / happ : 0.0;  // forward point outside the support: backward difference, else flat
        gv += gpv;
        const bool ok = isfinite(gv);
        g[k] = ok ? gv : 0.0;
        gp[k] = ok ? gpv : 0.0;
    }
    __syncthreads();
}

// D_MALA (host_mala.cpp): drift = (1/2) sigma scale (Sigma + eps2) g, scale = delta/|g| when delta > 0 and |g| > delta; lane i -> out[i].
__device__ void mala_drift(const DevSamplerArgs &a, const MalaArgs &M, int m, const double *g, double *out, double *s_red) {
    const int Nv = a.Nv;
    double n2 = 0;
    for (int k = threadIdx.x; k < Nv; k += blockDim.x) n2 += g[k] * g[k];
    n2 = wg_sum(n2, s_red);
    const double nrm = sqrt(n2);
    const bool live = isfinite(nrm);
    double scale = 1.0;
    if (M.delta > 0 && nrm > M.delta) scale = M.delta / nrm;
    const double s = 0.5 * a.sigma[m] * scale;
    const double *cov = a.cov + (size_t)m * Nv * Nv;
    for (int i = threadIdx.x; i < Nv; i += blockDim.x) {
        double acc = 0;
        for (int j = 0; j < Nv; j++) acc += cov[(size_t)i * Nv + j] * g[j];
        acc += a.epsi2 * g[i];
        out[i] = live ? s * acc : 0.0;
    }
    __syncthreads();
}

// initial gradient: the batch ran on the chains' CURRENT positions (parity P)
__global__ void __launch_bounds__(TB) k_mala_ginit(const DevSamplerArgs a, const MalaArgs M, const int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double *g = (double *)s_raw, *gp = g + a.Nv;
    const int m = blockIdx.x, Nv = a.Nv, C = a.C;
    double L0, pr0;
    int st0;
    mala_gradient(a, M, m, a.params_cur + ((size_t)P * C + m) * a.desc.Np, g, gp, L0, pr0, st0);
    for (int k = threadIdx.x; k < Nv; k += TB) {
        a.grad_cur[((size_t)P * C + m) * Nv + k] = g[k];
        a.gradP_cur[((size_t)P * C + m) * Nv + k] = gp[k];
    }
}

// forward-difference steps from chain 0's running mean (host_mala.cpp::compute_gradients)
__global__ void k_mala_steps(const DevSamplerArgs a, const MalaArgs M) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.Nv; k += gridDim.x * blockDim.x) M.h[k] = M.fd_step_rel * fmax(fabs(a.mu[k]), 1e-3);
}

// The MH test of iteration `it` for chain m, after the finite-difference batch of the proposals (state of parity P).
__global__ void __launch_bounds__(TB) k_mala_test(const DevSamplerArgs a, const MalaArgs M, const long it, const int P, const int learn,
                                                  double *scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int Nv = a.Nv, Np = a.desc.Np, C = a.C, m = blockIdx.x, tid = threadIdx.x;
    double *g = (double *)s_raw;        // [Nv] gradient at the proposal
    double *gp = g + Nv;                // [Nv] its prior share
    double *dp = gp + Nv;               // [Nv] drift at the proposal
    double *sf = dp + Nv;               // [Nv] forward solve:  L w = (x' - x) - drift(x)
    double *sr = sf + Nv;               // [Nv] reverse solve:  L w = (x - x') - drift(x')
    double *s_red = sr + Nv;            // [8]
    double *s_A = (double *)(((uintptr_t)(s_red + 8) + 15) & ~(uintptr_t)15);  // adaptation workspace when it fits in LDS
    __shared__ double s_scal[2];
    __shared__ AcceptOut s_o;
    const double *prop_p = a.params_prop + (size_t)m * Np, *prop_v = a.vars_prop + (size_t)m * Nv;
    const double *cur_v = a.vars_cur + ((size_t)P * C + m) * Nv;
    double L0, pr0;
    int st0;
    mala_gradient(a, M, m, prop_p, g, gp, L0, pr0, st0);
    for (int k = tid; k < Nv; k += TB) { M.grad_prop[(size_t)m * Nv + k] = g[k]; M.gradP_prop[(size_t)m * Nv + k] = gp[k]; }
    mala_drift(a, M, m, g, dp, s_red);
    const double *dc = M.drift_cur + (size_t)m * Nv;
    for (int i = tid; i < Nv; i += TB) {
        const double d = prop_v[i] - cur_v[i];
        sf[i] = d - dc[i];
        sr[i] = -d - dp[i];
    }
    __syncthreads();
    // both triangular solves in one sweep, column by column: row i receives its subtractions in ascending k, like the row-wise
    // substitution of multinormal_logpdf (host_mala.cpp)
    const double *LT = a.LT + (size_t)m * Nv * Nv;  // LT[k*Nv + i] = L[i][k]
    if (a.chol_in_lds && Nv <= 128) {
        // The factor in LDS (the adaptation's work area: the launch always reserves it when it fits), the two right-hand sides in the
        // registers of ONE wave (lane i: rows i and i + 64), the solved component of a column handed on by a lane read: no barrier and no
        // memory round trip per column (93 columns x two barriers x an L2 read were 45 us of this kernel), and the pivots' reciprocals
        // taken once beforehand, all at a time (a product with 1/d instead of a division by d in the chain: 1 ulp).
        double *Ls = s_A, *inv_d = s_A + (size_t)Nv * Nv;
        for (int i = tid; i < Nv * Nv; i += TB) Ls[i] = LT[i];
        for (int k = tid; k < Nv; k += TB) inv_d[k] = 1.0 / LT[(size_t)k * Nv + k];
        __syncthreads();
        if (tid < 64) {
            const int lane = tid, hi = lane + 64;
            double f0 = lane < Nv ? sf[lane] : 0.0, f1 = hi < Nv ? sf[hi] : 0.0, r0 = lane < Nv ? sr[lane] : 0.0, r1 = hi < Nv ? sr[hi] : 0.0;
#pragma clang loop unroll(disable)
            for (int k = 0; k < Nv; k++) {
                const double *row = Ls + (size_t)k * Nv;
                const double id = inv_d[k];
                const double l0 = (lane > k && lane < Nv) ? row[lane] : 0.0, l1 = (hi > k && hi < Nv) ? row[hi] : 0.0;
                const double wf = (k < 64 ? __shfl(f0, k, 64) : __shfl(f1, k - 64, 64)) * id;
                const double wr = (k < 64 ? __shfl(r0, k, 64) : __shfl(r1, k - 64, 64)) * id;
                if (lane == (k & 63)) {
                    if (k < 64) { f0 = wf; r0 = wr; }
                    else { f1 = wf; r1 = wr; }
                }
                f0 = f0 - l0 * wf; r0 = r0 - l0 * wr;  // (rows above the column: l = 0, the value stays)
                f1 = f1 - l1 * wf; r1 = r1 - l1 * wr;
            }
            if (lane < Nv) { sf[lane] = f0; sr[lane] = r0; }
            if (hi < Nv) { sf[hi] = f1; sr[hi] = r1; }
        }
        __syncthreads();
    } else
        for (int k = 0; k < Nv; k++) {
            if (tid == 0) { const double d = LT[(size_t)k * Nv + k]; sf[k] = sf[k] / d; sr[k] = sr[k] / d; }
            __syncthreads();
            const double wf = sf[k], wr = sr[k];
            for (int i = k + 1 + tid; i < Nv; i += TB) {
                const double l = LT[(size_t)k * Nv + i];
                sf[i] = sf[i] - l * wf;
                sr[i] = sr[i] - l * wr;
            }
            __syncthreads();
        }
    double qf = 0, qr = 0;
    for (int i = tid; i < Nv; i += TB) { qf += sf[i] * sf[i]; qr += sr[i] * sr[i]; }
    qf = wg_sum(qf, s_red);
    qr = wg_sum(qr, s_red);
    if (tid == 0) {
        const double lq_fwd = -0.5 * qf, lq_rev = -0.5 * qr;
        double logL = L0, logPost;
        const double logPr = pr0;
        if (logPr != -INFINITY && !isnan(logPr)) logPost = logL + logPr;       // generate_model, model_def.cpp:466-482
        else { logL = a.init_logL[m]; logPost = -INFINITY; }
        double u, u1;
        rng_uniform2(a.seed, RNG_ACCEPT, (uint32_t)m, (uint64_t)it, 0, u, u1);
        double r;
        if (!isnan(logL)) {
            if (logPost == -INFINITY) r = 0.;
            else {
                const double e = exp(logPost - a.logPost_cur[P * C + m] + lq_rev - lq_fwd);
                r = fmin(1.0, e);
                if (isnan(r)) r = 0.;
            }
        } else r = 0.;
        AcceptOut o;
        o.acc = (u <= r) ? 1 : 0;
        o.r = r;
        if (o.acc) { o.logL = logL; o.logPr = logPr; o.logPost = logPost; }
        else { o.logL = a.logL_cur[P * C + m]; o.logPr = a.logPr_cur[P * C + m]; o.logPost = a.logPost_cur[P * C + m]; }
        s_o = o;
        double *w = M.out + (size_t)m * 5;
        w[0] = (double)o.acc; w[1] = o.r; w[2] = o.logL; w[3] = o.logPr; w[4] = o.logPost;
    }
    __syncthreads();
    if (learn) {  // MALA.cpp:656-667: the chain's own position after the test, before any swap
        const double *ov = s_o.acc ? prop_v : cur_v;
        double *Aw = a.chol_in_lds ? s_A : scratch + (size_t)m * ((size_t)Nv * Nv + Nv);
        adapt_chain(a, m, it, ov, s_o.r, Aw, Aw + (size_t)Nv * Nv, s_red, s_scal);
    }
}

// Settles iteration it-1 (swap, state, records) and proposes iteration `it`.
template <bool PROPOSE>
__global__ void __launch_bounds__(TB) k_mala_settle(const DevSamplerArgs a, const MalaArgs M, const long it, const int P, const int pending,
                                                    const long rec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int Nv = a.Nv, Np = a.desc.Np, C = a.C, m = blockIdx.x, tid = threadIdx.x, Q = P ^ 1;
    double *s_vars = (double *)s_raw;   // [Nv]
    double *s_par = s_vars + Nv;        // [Np]
    double *s_g = s_par + Np;           // [Nv] gradient at the settled position
    double *s_z = s_g + Nv;             // [Nv+1]
    double *s_d = s_z + Nv + 1;         // [Nv] drift
    double *s_red = s_d + Nv;           // [8]
    double *nv = a.vars_cur + ((size_t)Q * C + m) * Nv, *np_ = a.params_cur + ((size_t)Q * C + m) * Np;
    double *ng = a.grad_cur + ((size_t)Q * C + m) * Nv, *ngp = a.gradP_cur + ((size_t)Q * C + m) * Nv;
    if (pending) {
        const long itp = it - 1;
        auto outcome = [&](int j) {
            const double *w = M.out + (size_t)j * 5;
            AcceptOut o;
            o.acc = (int)w[0]; o.r = w[1]; o.logL = w[2]; o.logPr = w[3]; o.logPost = w[4];
            return o;
        };
        const AcceptOut own = outcome(m);
        AcceptOut mine = own, src_o = own;
        int src = m;
        if (is_swap_iter(a, itp)) {
            double u;
            const int A = swap_first(a, itp, &u);
            if (m == A || m == A + 1) {
                const int partner = (m == A) ? A + 1 : A;
                const AcceptOut po = outcome(partner);
                AcceptOut oA = (m == A) ? own : po, oB = (m == A) ? po : own;
                const int swapped = resolve_swap(a, A, u, oA, oB);
                if (swapped) { src = partner; mine = (m == A) ? oA : oB; src_o = po; }
                if (m == A && tid == 0) {
                    atomicAdd((unsigned long long *)&a.counters[2], 1ull);
                    if (swapped) atomicAdd((unsigned long long *)&a.counters[3], 1ull);
                }
            }
        }
        const double *sv = src_o.acc ? a.vars_prop + (size_t)src * Nv : a.vars_cur + ((size_t)P * C + src) * Nv;
        const double *sp = src_o.acc ? a.params_prop + (size_t)src * Np : a.params_cur + ((size_t)P * C + src) * Np;
        const double *sg = src_o.acc ? M.grad_prop + (size_t)src * Nv : a.grad_cur + ((size_t)P * C + src) * Nv;
        const double *sgp = src_o.acc ? M.gradP_prop + (size_t)src * Nv : a.gradP_cur + ((size_t)P * C + src) * Nv;
        const double tr = a.Tcoefs[src] / a.Tcoefs[m];  // a swapped position sits at a new temperature: likelihood share re-tempered
        for (int i = tid; i < Nv; i += TB) {
            const double v = sv[i];
            s_vars[i] = v; nv[i] = v;
            const double pg = sgp[i];
            const double gv = (src == m) ? sg[i] : (sg[i] - pg) * tr + pg;
            s_g[i] = gv; ng[i] = gv; ngp[i] = pg;
        }
        for (int i = tid; i < Np; i += TB) { const double v = sp[i]; s_par[i] = v; np_[i] = v; }
        if (tid == 0) {
            a.logL_cur[Q * C + m] = mine.logL;
            a.logPr_cur[Q * C + m] = mine.logPr;
            a.logPost_cur[Q * C + m] = mine.logPost;
            a.moved[m] = src_o.acc;
            a.Pmove[m] = src_o.r;
            if (m == 0 && src_o.acc) a.counters[1] += 1;
            a.counters[8 + m] += src_o.acc;
            if (m == 0) a.counters[0] = it;
            if (a.stats && rec >= 0) {
                double *r = a.stats + ((size_t)rec * C + m) * 3;
                r[0] = mine.logL; r[1] = mine.logPr; r[2] = mine.logPost;
            }
        }
        __syncthreads();
        if (a.samples && rec >= 0)
            for (int i = tid; i < Nv; i += TB) a.samples[((size_t)rec * C + m) * Nv + i] = s_vars[i];
    } else {
        const double *cv = a.vars_cur + ((size_t)P * C + m) * Nv, *cp = a.params_cur + ((size_t)P * C + m) * Np;
        const double *cg = a.grad_cur + ((size_t)P * C + m) * Nv, *cgp = a.gradP_cur + ((size_t)P * C + m) * Nv;
        for (int i = tid; i < Nv; i += TB) { s_vars[i] = cv[i]; nv[i] = cv[i]; s_g[i] = cg[i]; ng[i] = cg[i]; ngp[i] = cgp[i]; }
        for (int i = tid; i < Np; i += TB) { s_par[i] = cp[i]; np_[i] = cp[i]; }
        if (tid == 0) {
            a.logL_cur[Q * C + m] = a.logL_cur[P * C + m];
            a.logPr_cur[Q * C + m] = a.logPr_cur[P * C + m];
            a.logPost_cur[Q * C + m] = a.logPost_cur[P * C + m];
        }
        __syncthreads();
    }
    if (!PROPOSE) return;
    // ---- proposal of iteration `it`: x' = x + drift + L z (new_prop_values with the drift, host_mala.cpp)
    mala_drift(a, M, m, s_g, s_d, s_red);
    normals_into(a, m, it, s_z);
    __syncthreads();
    double *dc = M.drift_cur + (size_t)m * Nv;
    for (int i = tid; i < Nv; i += TB) {
        const double s = Lz_row(a, m, i, s_z);
        const double v = s_vars[i] + s_d[i] + s;
        dc[i] = s_d[i];
        a.vars_prop[(size_t)m * Nv + i] = v;
        s_par[a.index_to_relax[i]] = v;  // update_params_with_vars (distinct indices: no race)
    }
    __syncthreads();
    for (int i = tid; i < Np; i += TB) a.params_prop[(size_t)m * Np + i] = s_par[i];
    if (m == 0)
        for (int k = tid; k < Nv; k += TB) M.h[k] = M.fd_step_rel * fmax(fabs(a.mu[k]), 1e-3);
}
