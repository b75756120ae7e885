// fd_batch.hip -- finite-difference gradient batches built ON the device.
//
// The forward-difference gradient (the drift the reference leaves as a stub, MALA.cpp:321-337) needs, per chain,
// Nvars+1 evaluations.  Building those tables on the host costs far more than evaluating them (C3: 1880 tables,
// ~20 us each, against ~4 ms of likelihood kernel); here one workgroup per (chain, perturbed variable) perturbs the
// parameter vector, evaluates the log-prior (and the prior at the backward point, for the one-sided fallback at the
// edge of a prior's support) and writes its multiplet table straight into the likelihood kernel's input block:
//   k_fd_unpack (C*(Nvars+1) workgroups) -> k_loglike (one launch, B = C*(Nvars+1)) -> k_finalize.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"
#include "dev_unpack.h"
#include "kernels.h"

namespace tamcmc {
namespace {

constexpr int FB = 128;

__global__ void k_fill_poly_fd(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
}

struct FdArgs {
    ModelDesc desc;
    TablePtrs T;
    int C, E, Nv;
    const double *params;  // [C x Np]
    const int *idx;        // [Nv] index_to_relax
    const double *h;       // [Nv] steps
    double *logPr_plus;    // [C x E] log-prior at the evaluation point (e=0: the base point)
    double *logPr_minus;   // [C x E] log-prior at theta - h e_k (e>=1)
    int *status;           // [C x E]
};

__global__ void __launch_bounds__(FB) k_fd_unpack(const FdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int Np = a.desc.Np;
    double *s_params = (double *)s_raw;
    const UnpackLds U = carve_unpack_lds((unsigned char *)(s_params + Np));
    const int slot = blockIdx.x, c = slot / a.E, e = slot - c * a.E, tid = threadIdx.x;
    for (int i = tid; i < Np; i += FB) s_params[i] = a.params[(size_t)c * Np + i];
    unpack_begin(a.desc, U);
    double lp_minus = 0.0;
    const bool with_prior = a.desc.prior_class != 0;
    if (e > 0) {
        const int i = a.idx[e - 1];
        const double x0 = a.params[(size_t)c * Np + i], hh = a.h[e - 1];
        if (with_prior) {
            if (tid == 0) s_params[i] = x0 - hh;
            __syncthreads();
            lp_minus = wg_log_prior(a.desc, s_params, U, false);
            if (tid == 0) *U.reject = 0;
            __syncthreads();
        }
        if (tid == 0) s_params[i] = x0 + hh;
        __syncthreads();
    }
    double lp = 0.0;
    if (with_prior) lp = wg_log_prior(a.desc, s_params, U, true);
    else {
        if (tid == FB - 1) mt::shared_scalars_base(a.desc.model_id, s_params, a.desc.plength, *U.S);
        __syncthreads();
    }
    // the likelihood is evaluated whatever the prior says: the caller combines the parts
    if (tid == 0) *U.status = TAMCMC_OK;
    __syncthreads();
    wg_unpack(a.desc, s_params, U, slot, a.T, true);
    if (tid == 0) {
        a.logPr_plus[slot] = lp;
        a.logPr_minus[slot] = lp_minus;
        a.status[slot] = *U.status;
    }
}

}  // namespace
}  // namespace tamcmc

using namespace tamcmc;

static int fd_run(tamcmc_hip_ctx *c, int model_id, int prior_class, int C, const double *params, int64_t Nparams,
                  const int32_t *plength, const int32_t *index_to_relax, int Nvars, const double *hstep, const double *Tcoefs,
                  double p, const double *priors, const int32_t *priors_switch, const double *extra_priors, double *logL0,
                  double *logPr0, double *grad) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (c->Nx <= 0) return TAMCMC_ERR_NO_SPECTRUM;
    if (C < 0 || Nvars < 0 || !params || !plength || !index_to_relax || !hstep || !logL0 || !grad || Nparams < 1) return TAMCMC_ERR_BAD_ARG;
    if (prior_class != 0 && (!priors || !priors_switch || !extra_priors)) return TAMCMC_ERR_BAD_ARG;
    if (C == 0) return TAMCMC_OK;
    long psum = 0;
    for (int i = 0; i < 11; i++) psum += plength[i];
    if (psum != Nparams) return TAMCMC_ERR_BAD_ARG;
    for (int k = 0; k < Nvars; k++)
        if (index_to_relax[k] < 0 || index_to_relax[k] >= Nparams) return TAMCMC_ERR_BAD_ARG;
    const int per = mt::count_multiplets(model_id, plength);
    if (per < 0) return TAMCMC_ERR_BAD_MODEL;
    const int stride = plength[8] > 0 ? plength[8] : 1;
    if ((stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int E = Nvars + 1, B = C * E;
    const size_t Np = (size_t)Nparams, Nv = (size_t)Nvars;
    if (!c->poly_ready) {
        HIPCHK(c, c->d_poly.reserve(sizeof(mt::PolyTab)));
        hipLaunchKernelGGL(k_fill_poly_fd, dim3(1), dim3(64), 0, st, (mt::PolyTab *)c->d_poly.p);
        HIPCHK(c, hipGetLastError());
        c->poly_ready = true;
    }
    // ---- one pinned input block -> one H2D copy ----
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t o = 0;
    const size_t o_params = o; o = al(o + (size_t)C * Np * 8);
    const size_t o_h = o; o = al(o + Nv * 8);
    const size_t o_pr = o; o = al(o + 4 * Np * 8);
    const size_t o_ex = o; o = al(o + 10 * 8);
    const size_t o_pl = o; o = al(o + 11 * 4);
    const size_t o_idx = o; o = al(o + Nv * 4);
    const size_t o_sw = o; o = al(o + Np * 4);
    const size_t in_bytes = o;
    // device-only areas
    const size_t o_lpp = o; o = al(o + (size_t)B * 8);
    const size_t o_lpm = o; o = al(o + (size_t)B * 8);
    const size_t o_st = o; o = al(o + (size_t)B * 4);
    const size_t out_bytes = o - in_bytes;
    const StageLayout L(B, stride, (size_t)B * per);
    const size_t o_tab = o; o = al(o + L.bytes);
    HIPCHK(c, c->h_fd.reserve(in_bytes + out_bytes));
    HIPCHK(c, c->d_fd.reserve(o));
    unsigned char *hb = c->h_fd.p, *db = c->d_fd.p;
    std::memcpy(hb + o_params, params, (size_t)C * Np * 8);
    std::memcpy(hb + o_h, hstep, Nv * 8);
    if (prior_class != 0) {
        std::memcpy(hb + o_pr, priors, 4 * Np * 8);
        std::memcpy(hb + o_ex, extra_priors, 10 * 8);
        std::memcpy(hb + o_sw, priors_switch, Np * 4);
    }
    std::memcpy(hb + o_pl, plength, 11 * 4);
    std::memcpy(hb + o_idx, index_to_relax, Nv * 4);
    HIPCHK(c, hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, st));

    FdArgs fa;
    fa.desc.model_id = model_id; fa.desc.prior_class = prior_class; fa.desc.Np = (int)Nparams; fa.desc.per = per;
    fa.desc.stride = stride; fa.desc.Nx = (int)c->Nx;
    fa.desc.x_first = c->hx[0]; fa.desc.x_last = c->hx[(size_t)c->Nx - 1]; fa.desc.step = c->hx[1] - c->hx[0];
    fa.desc.plength = (const int *)(db + o_pl); fa.desc.priors_switch = (const int *)(db + o_sw);
    fa.desc.priors = (const double *)(db + o_pr); fa.desc.extra = (const double *)(db + o_ex); fa.desc.poly = c->d_poly.p;
    unsigned char *tb = db + o_tab;
    fa.T.mults = (tamcmc_multiplet *)(tb + L.off_mults); fa.T.pairs = (int *)(tb + L.off_pairs);
    fa.T.nh = (int *)(tb + L.off_nh); fa.T.nn = (int *)(tb + L.off_nn); fa.T.noise = (double *)(tb + L.off_noise);
    fa.C = C; fa.E = E; fa.Nv = Nvars;
    fa.params = (const double *)(db + o_params); fa.idx = (const int *)(db + o_idx); fa.h = (const double *)(db + o_h);
    fa.logPr_plus = (double *)(db + o_lpp); fa.logPr_minus = (double *)(db + o_lpm); fa.status = (int *)(db + o_st);
    const size_t lds = Np * 8 + unpack_lds_bytes() + 32;
    hipLaunchKernelGGL(k_fd_unpack, dim3(B), dim3(FB), lds, st, fa);
    HIPCHK(c, hipGetLastError());

    const int Nx = (int)c->Nx;
    const int tbins = tile_bins(c->wgs, c->K);
    const int ntiles = (Nx + tbins - 1) / tbins;
    HIPCHK(c, c->d_part.reserve((size_t)B * ntiles * 2));
    HIPCHK(c, c->d_S.reserve((size_t)B));
    HIPCHK(c, c->h_S.reserve((size_t)B));
    LoglikeArgs a;
    a.x = c->dx.p; a.y = c->dy.p; a.logx = c->dlogx.p; a.Nx = Nx; a.B = B; a.ntiles = ntiles;
    a.x0 = c->hx[0]; a.step = c->hx[1] - c->hx[0];
    a.mults = fa.T.mults; a.offsets = fa.T.pairs; a.noise = fa.T.noise; a.noise_stride = stride;
    a.nharvey = fa.T.nh; a.nnoise = fa.T.nn; a.partials = c->d_part.p; a.model = nullptr;
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, st));
    HIPCHK(c, launch_loglike(a, c->precision, c->wgs, c->K, false, st));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, launch_finalize(c->d_part.p, B, ntiles, c->d_S.p, st));
    HIPCHK(c, hipMemcpyAsync(c->h_S.p, c->d_S.p, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hb + in_bytes, db + in_bytes, out_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->timing) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->kernel_ms += ms;
        c->launches += 1;
        c->evals += B;
    }
    const double *lpp = (const double *)(hb + o_lpp), *lpm = (const double *)(hb + o_lpm);
    const int *stt = (const int *)(hb + o_st);
    const long pl = (long)p;
    int first_err = TAMCMC_OK;
    for (int ch = 0; ch < C; ch++) {
        const double T = Tcoefs ? Tcoefs[ch] : 1.0;
        auto logL_of = [&](int e) {
            const size_t s = (size_t)ch * E + e;
            if (stt[s] != TAMCMC_OK) { if (first_err == TAMCMC_OK) first_err = stt[s]; return (double)NAN; }
            long double f = c->h_S.p[s];
            f = -pl * f;
            return (double)(f / T);
        };
        const double L0 = logL_of(0);
        logL0[ch] = L0;
        if (logPr0) logPr0[ch] = lpp[(size_t)ch * E];
        const double pr0 = lpp[(size_t)ch * E];
        for (int k = 0; k < Nvars; k++) {
            const double x0 = params[(size_t)ch * Np + index_to_relax[k]];
            volatile double xp = x0 + hstep[k];
            const double happ = xp - x0;  // the step actually applied (the device adds the same two doubles)
            double g = (logL_of(k + 1) - L0) / happ;
            if (prior_class != 0) {
                if (!std::isfinite(g)) g = 0.0;
                const double prp = lpp[(size_t)ch * E + k + 1], prm = lpm[(size_t)ch * E + k + 1];
                double gp;
                if (std::isfinite(prp)) gp = (prp - pr0) / happ;
                else gp = std::isfinite(prm) ? (pr0 - prm) / happ : 0.0;  // forward point outside the support: backward, else flat
                g += gp;
            }
            grad[(size_t)ch * Nv + k] = g;
        }
    }
    return first_err;
}

extern "C" {

int tamcmc_hip_fd_gradient(tamcmc_hip_ctx *c, int model_id, int C, const double *params, int64_t Nparams,
                           const int32_t *plength, const int32_t *index_to_relax, int Nvars, const double *hstep,
                           const double *Tcoefs, double p, double *logL0, double *grad) {
    return fd_run(c, model_id, 0, C, params, Nparams, plength, index_to_relax, Nvars, hstep, Tcoefs, p, nullptr, nullptr, nullptr,
                  logL0, nullptr, grad);
}

int tamcmc_hip_fd_gradient_posterior(tamcmc_hip_ctx *c, int model_id, int prior_class, int C, const double *params,
                                     int64_t Nparams, const int32_t *plength, const int32_t *index_to_relax, int Nvars,
                                     const double *hstep, const double *Tcoefs, double p, const double *priors,
                                     const int32_t *priors_switch, const double *extra_priors, double *logL0, double *logPr0,
                                     double *grad) {
    if (prior_class != 2 && prior_class != 3) return TAMCMC_ERR_BAD_MODEL;
    return fd_run(c, model_id, prior_class, C, params, Nparams, plength, index_to_relax, Nvars, hstep, Tcoefs, p, priors,
                  priors_switch, extra_priors, logL0, logPr0, grad);
}

}  // extern "C"
