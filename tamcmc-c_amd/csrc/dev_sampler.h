// dev_sampler.h -- interface of the device-resident sampler engine (dev_sampler.hip).
#pragma once
#include <stdint.h>

#include "../../include/tamcmc_hip.h"
#if defined(__HIPCC__)
#include "dev_unpack.h"
#endif

#define TAMCMC_MAX_CHAINS 64  // the reference caps at 24 (MALA.cpp:580-587); BASELINE config 5 asks for 40

namespace tamcmc {

#if defined(__HIPCC__)
// Kernel argument block: device pointers + scalars (passed by value).
struct DevSamplerArgs {
    ModelDesc desc;       // model_id, prior_class, Np, per, stride, Nx, grid, plength/priors/extra/poly pointers
    int C, Nv, ntiles, chol_in_lds;
#ifdef TAMCMC_PROBE       // probe build only (tools/learn_probe.py): leave adapt_chain after phase N -- the chains are then WRONG, only times are read
    int probe = 0;
#endif
    int swap_rule;        // 1: chain B's stored logPosterior after a swap as MALA.cpp:433,444 execute it
    long pl;              // likelihood_params truncated to long (likelihoods.h:14)
    long dN_mixing;
    uint64_t seed;
    double c0, epsilon1, epsi2, A1, target_acceptance;
    const int *index_to_relax;
    const double *Tcoefs;
    // chain state
    double *vars_cur, *params_cur, *logL_cur, *logPr_cur, *logPost_cur, *init_logL;
    double *vars_prop, *params_prop, *logPr_prop;
    int *status_prop, *moved;
    double *Pmove;
    long *counters;       // [0] iteration, [1] accepted moves of chain 0, [2] swap attempts, [3] swaps accepted, [8 + m] recorded moves of chain m
    // proposal law
    double *LT, *cov, *mu, *sigma;   // LT = transposed Cholesky factor of (cov+eps2)*sigma
    double *lz;           // [2][C][Nv] L z of the NEXT iteration, computed ahead by spare workgroups while L is frozen
    double *grad_cur, *gradP_cur;  // Langevin step: [2][C][Nv] gradient of the tempered log-posterior at the chains' positions, and the prior's share
    // likelihood-kernel input block written by k_propose_unpack
    tamcmc_multiplet *mults;
    int *pairs, *nh, *nn;
    double *noise;
    double *partials;
    double *bg;           // [C x ntiles x 8] background series per (slot, tile), FAST far field only (else nullptr)
    int tile_bins;
    // records
    double *samples, *stats;
};
#endif

struct DevSamplerInit {
    int model_id, prior_class, C, Np, Nv;
    double likelihood_params;
    const int *plength, *index_to_relax, *priors_switch;
    const double *priors, *extra_priors, *Tcoefs;
    uint64_t seed;
    long dN_mixing;
    double c0, epsilon1, epsi2, A1, target_acceptance;
    int chain_groups = 0;  // 0 = default (2 from 8 chains on): stream groups of the lockstep scheme
    int swap_rule = 0;     // tamcmc_sampler_config.swap_rule
    int use_drift = 0;     // 1: Langevin step (dev_mala_impl.h)
    double delta = 0, fd_step_rel = 1e-7;  // drift truncation (0 = none), relative forward-difference step
};

class DevSampler {
    struct Impl;
    Impl *impl;

  public:
    DevSampler();
    ~DevSampler();
    DevSampler(const DevSampler &) = delete;
    DevSampler &operator=(const DevSampler &) = delete;
    int init(tamcmc_hip_ctx *ctx, const DevSamplerInit &in);
    int upload_state(const double *vars, const double *params, const double *logL, const double *logPr,
                     const double *logPost, const double *init_logL);
    int upload_proposal(int m, const double *L_rowmajor, const double *cov, const double *mu, double sigma);
    int download_state(double *vars, double *params, double *logL, double *logPr, double *logPost, double *Pmove,
                       int *moved, long *counters, long *moves_per_chain = nullptr);
    int download_proposal(int m, double *cov, double *mu, double *sigma);
    int download_gradient(double *grad, double *grad_prior);
    int download_last_proposal(double *vars_prop, double *grad_prop);  // use_drift: what the last iteration tested  // use_drift: [C x Nv] gradient the engine holds for the chains' positions
    // [0] Nvars [1] Nparams [2] adaptation workspace in LDS (1) / global scratch (0) [3] fused step available [4] chain groups
    // [5] iterations run fused [6] iterations run by the lockstep kernels [7] chains
    void info(long out[8]) const;
    int run(long it0, long n_iter, const char *learn, double *samples, double *stats);
    int run_mala(long it0, long n_iter, const char *learn, double *samples, double *stats);  // use_drift = 1 (dev_mala_impl.h)
};

}  // namespace tamcmc
