/*
 * tamcmc_sampler.h -- C ABI of the sampler that calls the hot path (host-side mirror of the reference's
 * MALA + Model_def classes, tamcmc/sources/MALA.cpp, tamcmc/sources/model_def.cpp).
 *
 * The reference drives the hot path from MALA::execute (MALA.cpp:555-747): per iteration, for every tempered chain,
 * propose -> Model_def::generate_model -> accept; then Robbins-Monro adaptation and a parallel-tempering swap.
 * tamcmc_sampler_run() is that loop with the chains batched into one device call per iteration.
 * The configuration struct carries what Config::setup (config.cpp:167-396) would have produced from the
 * .cfg/.model/.data files: Input_Data{inputs, relax, priors, plength, extra_priors} and the !MALA section.
 */
#ifndef TAMCMC_SAMPLER_H
#define TAMCMC_SAMPLER_H

#include <stdint.h>

#include "tamcmc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tamcmc_sampler tamcmc_sampler;

typedef struct tamcmc_sampler_config {
    /* modeling (config_default.cfg !Modeling; ids from Config/default/{models,priors,likelihoods}_ctrl.list) */
    int32_t model_id;        /* model_fct_name_switch: 3, 11, 23 */
    int32_t prior_class;     /* prior_fct_name_switch: 2 = io_MS_Global, 3 = io_local */
    int32_t likelihood_id;   /* 0 = chi(2,2p) */
    int32_t use_drift;       /* 0 = adaptive random-walk MH (the reference), 1 = Langevin drift with FD gradient */
    double likelihood_params;/* p */
    int64_t Nparams;
    const double *inputs;          /* [Nparams] initial parameter vector */
    const int32_t *relax;          /* [Nparams] 1 = free */
    const int32_t *plength;        /* [11] */
    const double *priors;          /* [4 x Nparams] row-major */
    const int32_t *priors_switch;  /* [Nparams] primitive prior ids (primepriors_ctrl.list) */
    const double *extra_priors;    /* [n_extra] */
    int32_t n_extra;
    /* !MALA section */
    int32_t Nchains;
    double lambda_temp, target_acceptance, c0, epsilon1, epsilon2, A1, delta, delta_x;
    const int64_t *Nt_learn;       /* [n_Nt_learn] */
    const int64_t *periods_learn;  /* [n_Nt_learn-1] */
    int32_t n_Nt_learn;
    int32_t engine;                /* 0 = host-driven loop (one batched device call per iteration),
                                      1 = device-resident iteration (proposal, priors, unpack, accept, swap, adaptation on the GPU;
                                          use_drift must be 0) */
    int64_t dN_mixing;
    const double *init_errors;     /* [Nvars] initial proposal standard deviations (errors_default.cfg), NULL -> 1 */
    /* additions of this build */
    uint64_t seed;                 /* counter-based RNG seed (the reference seeds libc rand() with time(NULL)) */
    double fd_step_rel;            /* forward-difference step = fd_step_rel * max(|theta_k|, 1e-3); 0 -> 1e-7 */
    int32_t chain_groups;          /* device engine, lockstep scheme (iterations with adaptation): the chains run as this many groups
                                      on separate HIP streams (one group's proposal kernel overlaps another's likelihood kernel).
                                      0 = default (2 from 8 chains on); use 1 when several stars share a GPU
                                      (tamcmc_sampler_run_packed): the co-resident stars already fill each other's gaps */
    int32_t swap_rule;             /* what chain B = A+1 stores as logPosterior after an accepted parallel-tempering swap:
                                      0 = logL_A(T_B) + logPrior_A, the posterior of the position it receives (default);
                                      1 = the reference as executed: MALA.cpp:433 overwrites logPrior[A] with B's before MALA.cpp:444
                                          reads it, so B stores logL_A(T_B) + its OWN OLD prior (that value enters B's next MH ratio,
                                          MALA.cpp:515, until B accepts a move) */
} tamcmc_sampler_config;

/* The context must already hold the spectrum (tamcmc_hip_set_spectrum). It is borrowed, not owned: the sampler uses its stream and
 * device buffers.  Destruction order is free: tamcmc_hip_destroy on a context that still has samplers defers the release to the
 * last tamcmc_sampler_destroy. */
int tamcmc_sampler_create(tamcmc_sampler **s, tamcmc_hip_ctx *ctx, const tamcmc_sampler_config *cfg);
void tamcmc_sampler_destroy(tamcmc_sampler *s);
int64_t tamcmc_sampler_nvars(const tamcmc_sampler *s);

/* ---- size limits of the sampler (the reference factors any size with Eigen's LLT, MALA.cpp:339-369, and caps Nchains at 24, :580-587) ----
 *   quantity                       limit                 beyond it
 *   Nchains                        <= 64                 tamcmc_sampler_create returns TAMCMC_ERR_BAD_ARG
 *   Harvey terms                   <= TAMCMC_MAX_HARVEY  TAMCMC_ERR_BAD_ARG
 *   Nvars, device engine:          none                  the adaptation's work matrix (Nvars^2 + Nvars doubles: covariance update, Cholesky
 *     adaptation workspace in LDS  while (Nvars^2 + Nvars) 8 B + (Nparams + 2 Nvars) 8 B + 4.3 KB <= 150 KB, i.e. Nvars <~ 135
 *                                                        factor) moves from LDS to a per-chain block of device memory -- same operations
 *                                                        in the same order, same factor bit for bit, several times slower per learning
 *                                                        iteration (TAMCMC_INFO_ADAPT_IN_LDS = 0); the Langevin engine's test kernel: Nvars <~ 139
 *   Nparams + 2 Nvars, device      <= 971                the fused one-launch iteration borrows the likelihood tile's 12 KB of LDS for its
 *     engine, fused step                                 candidate roles; longer vectors run every iteration on the lockstep kernels
 *                                                        (TAMCMC_INFO_FUSED_AVAILABLE = 0) -- same chains bit for bit
 *   red-giant models (ids 25/27)   lockstep kernels only; no Langevin step (use_drift = 1 -> TAMCMC_ERR_BAD_MODEL)
 * The host-driven engine has no size-dependent branches (host memory, column Cholesky).
 * tamcmc_sampler_get_info reports which side of each limit a sampler is on and how many iterations each scheme has run. */
#define TAMCMC_INFO_ENGINE 0          /* 0 host-driven, 1 device-resident */
#define TAMCMC_INFO_NVARS 1
#define TAMCMC_INFO_NPARAMS 2
#define TAMCMC_INFO_NCHAINS 3
#define TAMCMC_INFO_ADAPT_IN_LDS 4    /* device engine: 1 = adaptation workspace in LDS, 0 = in device memory; -1 host engine */
#define TAMCMC_INFO_FUSED_AVAILABLE 5 /* device engine: the fused one-launch iteration can be used for this star */
#define TAMCMC_INFO_CHAIN_GROUPS 6
#define TAMCMC_INFO_ITER_FUSED 7      /* iterations run as fused launches since creation */
#define TAMCMC_INFO_ITER_LOCKSTEP 8   /* iterations run by the lockstep kernels (adaptation, long vectors, red giants, Langevin) */
#define TAMCMC_SAMPLER_INFO_N 9
int tamcmc_sampler_get_info(const tamcmc_sampler *s, int64_t *info, int32_t n);

/* Advances all chains by n_iter iterations.  Optional outputs, one record per iteration after the swap step
 * (what update_buffer_params / update_buffer_stat_criteria record, MALA.cpp:708-710):
 *   samples : [n_iter x Nchains x Nvars]      stats : [n_iter x Nchains x 3] = logL (tempered), logPrior, logPosterior */
int tamcmc_sampler_run(tamcmc_sampler *s, int64_t n_iter, double *samples, double *stats);
/* S independent stars at once: tamcmc_sampler_run on every sampler, one host thread each (each sampler on its OWN context; the
 * contexts may share a GPU).  One star's MCMC iteration is two short dependent kernels and leaves most of an MI355X idle, so
 * co-resident stars raise the GPU's aggregate samples/s (~2x at 4 stars of the C3 size); each star's samples are bit-identical to a
 * run on its own (no shared state, random numbers addressed by (seed, chain, iteration)).  samples / stats: S pointers (or NULL
 * arrays / NULL entries).  Returns the first non-zero status. */
int tamcmc_sampler_run_packed(tamcmc_sampler *const *s, int32_t S, int64_t n_iter, double *const *samples, double *const *stats);

/* The random numbers iteration `iteration` consumes -- the counter-based streams (csrc/rng.h) that replace the reference's
 * rand() (MALA.cpp:62-63, random_JB.cpp:99-105,255), pure functions of (seed, chain, iteration): z [Nchains x Nvars] = the normals of
 * new_prop_values (MALA.cpp:352), u_accept [Nchains] = the comparators of update_position_MH (MALA.cpp:467,536), *u_swap / *ind_A = the
 * comparator and first chain of parallel_tempering (MALA.cpp:400,406).  Lets a caller replay or audit any step. Pointers may be NULL. */
int tamcmc_sampler_draws(const tamcmc_sampler *s, int64_t iteration, double *z, double *u_accept, double *u_swap, int32_t *ind_A);

/* Current state. Any pointer may be NULL.
 *   vars [Nchains x Nvars], logL/logPrior/logPost/Pmove/sigma [Nchains], counters [4] = iteration, accepted moves
 *   of chain 0, swap attempts, swaps accepted */
int tamcmc_sampler_get_state(const tamcmc_sampler *s, double *vars, double *logL, double *logPrior, double *logPost,
                             double *Pmove, double *sigma, int64_t *counters);
/* use_drift = 1: the gradient of the tempered log-posterior the sampler HOLDS for each chain's current position -- computed when that
 * position was proposed, carried through the accept step and, after a parallel-tempering swap, moved to the partner with its likelihood
 * share re-tempered (grad_prior = the prior's share).  grad / grad_prior [Nchains x Nvars], valid [Nchains] (0: will be recomputed before
 * its next use: start of a run, positions set from outside); any pointer may be NULL.  An audit entry like tamcmc_sampler_draws. */
int tamcmc_sampler_get_gradient(const tamcmc_sampler *s, double *grad, double *grad_prior, int32_t *valid);
/* What the LAST iteration tested -- the proposals vars_prop [Nchains x Nvars], their logL (tempered) / logPrior / logPosterior
 * stats_prop [Nchains x 3] and, with use_drift = 1, lq [Nchains x 2] = log q(x'|x), log q(x|x') without the constant the two share,
 * and grad_prop [Nchains x Nvars] = the gradient at the proposals.  Audit entry (tests replay an iteration piece by piece); any
 * pointer may be NULL.  Host-driven engine: everything.  Device-resident engine: use_drift = 1 only (else TAMCMC_ERR_BAD_ARG),
 * vars_prop and grad_prop; stats_prop and lq come back as NaN (those scalars are not kept). */
int tamcmc_sampler_get_last_test(const tamcmc_sampler *s, double *vars_prop, double *stats_prop, double *lq, double *grad_prop);
/* moves [Nchains]: for every chain the number of iterations since creation whose record carries moved = 1 -- the flag the reference
 * stores per iteration and chain (MALA.cpp:543-545; a swap exchanges the pair's flags, :436, :446) and its acceptance diagnostic counts
 * per buffer (Outputs::reject_rate / count_accepted_vals, outputs.cpp:1824-1858).  Differences between two calls / the iterations in
 * between = the acceptance rates of that buffer (tamcmc_outputs_write_acceptance). */
int tamcmc_sampler_get_move_counts(const tamcmc_sampler *s, int64_t *moves);
/* proposal law of chain m: mu [Nvars], covarmat [Nvars x Nvars] (restore file content, outputs.cpp:863-1025) */
int tamcmc_sampler_get_proposal(const tamcmc_sampler *s, int32_t m, double *mu, double *covarmat);
int tamcmc_sampler_set_proposal(tamcmc_sampler *s, int32_t m, const double *mu, const double *covarmat, double sigma);
/* Chain positions vars [Nchains x Nvars] from outside (restart): priors and likelihoods are re-evaluated on the device;
 * iteration >= 0 also sets the iteration counter (the learning schedule and gamma = c0/(1+i) depend on it). */
int tamcmc_sampler_set_state(tamcmc_sampler *s, const double *vars, int64_t iteration);

/* ---- checkpoint / resume: the reference's restore files <root>1.dat (positions), <root>2.dat (sigmas, mus), <root>3.dat
 * (covariance matrices) -- Outputs::write_buffer_restore outputs.cpp:863-1025, Config::read_restore_files config.cpp:1734-1990.
 * Written with 17 significant digits (the reference: 6); the *_mean blocks repeat the last values. */
int tamcmc_outputs_write_restore(const char *root, int32_t Nchains, int32_t Nvars, int64_t iteration, const char *const *names,
                                 const double *vars, const double *sigmas, const double *mus, const double *covarmats);
/* sizes always; arrays (may be NULL): vars [Nchains x Nvars], sigmas [Nchains], mus [Nchains x Nvars], covarmats [Nchains x Nvars^2] */
int tamcmc_outputs_read_restore(const char *root, int32_t *Nchains, int32_t *Nvars, int64_t *iteration, double *vars, double *sigmas,
                                double *mus, double *covarmats);
int tamcmc_sampler_write_restore(const tamcmc_sampler *s, const char *root, const char *const *var_names);
/* the three switches of the reference's !Outputs section: do_restore_variables, do_restore_proposal, do_restore_last_index */
int tamcmc_sampler_read_restore(tamcmc_sampler *s, const char *root, int32_t restore_variables, int32_t restore_proposal,
                                int32_t restore_last_index);

/* ---- the reference's on-disk sample formats (outputs.cpp:1231-1333, :1472-1550) and summary statistics ---- */
/* <root>params.hdr + <root>params_chain-<m>.bin: raw little-endian doubles [sample][var]; samples = [n x Nchains x Nvars]
 * exactly as tamcmc_sampler_run returns them.  names (may be NULL) = Nparams parameter names for the header; plength has n_plength
 * entries (11 for the spectrum models of this build, 10 for the reference's Gaussian-envelope models).
 * append != 0: a later buffer of the same run -- the .bin files grow and, as in the reference (outputs.cpp:1268), the header is
 * rewritten with the cumulative `! Nsamples_done` (rows already on disk + n). */
int tamcmc_outputs_write_params(const char *root, const double *samples, int64_t n, int32_t Nchains, int32_t Nvars,
                                int64_t Nsamples_total, const int32_t *relax, const int32_t *plength, int32_t n_plength, int64_t Nparams,
                                const double *inputs, const char *const *names, int32_t append);
/* <root>stat_criteria.hdr/.bin: per sample logLikelihood[0:Nchains], logPrior[0:Nchains], logPosterior[0:Nchains];
 * stats = [n x Nchains x 3] as tamcmc_sampler_run returns them; append as above (outputs.cpp:1502). */
int tamcmc_outputs_write_stat_criteria(const char *root, const double *stats, int64_t n, int32_t Nchains, int32_t append);
/* <file>: the acceptance diagnostic the reference appends one line to per buffer (Outputs::write_txt_acceptance, outputs.cpp:747-790):
 * x = (Ncopy + 0.5) Nbuffer + Nsamples_init (the buffer's average sample index, outputs.cpp:1838) followed by acceptance_rate[0:Nchains-1]
 * streamed as an Eigen row (6 significant digits, columns padded to the widest entry).  first != 0 starts the file with its header. */
int tamcmc_outputs_write_acceptance(const char *file, double xaxis, const double *rates, int32_t Nchains, int32_t first);
/* reads it back: *Nchains from the header, then up to max_rows lines into xaxis [max_rows] and rates [max_rows x Nchains] (either may be
 * NULL to query *n_rows) */
int tamcmc_outputs_read_acceptance(const char *file, int32_t *Nchains, int64_t max_rows, int64_t *n_rows, double *xaxis, double *rates);
/* reads back one chain (what tools/bin2txt_params.cpp does): the header's Nsamples_done rows; samples may be NULL to query the count */
int tamcmc_outputs_read_params(const char *root, int32_t chain, double *samples, int64_t max_samples, int64_t *n_read,
                               int32_t *Nchains, int32_t *Nvars);
/* mean, median, population standard deviation per variable (tools/quick_samples_stats.cpp:4-35 via bin2txt_params.cpp:165-168);
 * samples rows are row_stride doubles apart */
int tamcmc_params_summary(const double *samples, int64_t n, int32_t Nvars, int64_t row_stride, double *mean, double *median,
                          double *stddev);

/* Evidence diagnostic of the tempered ladder (Diagnostics::evidence_calc, diagnostics.cpp:980-1019; quad_interpol, interpol.cpp:46-101):
 * beta = 1/Tcoefs, L_beta[m] = mean recorded log-likelihood of chain m, both resampled to interp_factor*Nchains points, evidence =
 * average of the resampled L_beta.  logL[i*row_stride + m*col_stride]: for tamcmc_sampler_run's stats block pass row_stride =
 * 3*Nchains, col_stride = 3.  beta_interp / L_beta_interp ([interp_factor*Nchains]) may be NULL. */
int tamcmc_evidence_calc(const double *Tcoefs, int32_t Nchains, const double *logL, int64_t n, int64_t row_stride, int64_t col_stride,
                         int32_t interp_factor, double *beta, double *L_beta, double *beta_interp, double *L_beta_interp, double *evidence);
/* appends one line (sample count, L_beta, evidence) to the text file the reference's diagnostics keep (diagnostics.cpp:1021-1066) */
int tamcmc_outputs_write_evidence(const char *file, int64_t n_samples, int32_t Nchains, const double *beta, const double *L_beta,
                                  int32_t interp_factor, double evidence, int32_t first);

/* Host log-prior of one parameter vector = Model_def::call_prior (model_def.cpp:421-464) for the model classes
 * io_MS_Global (2) and io_local (3): long double arithmetic, the reference's term order.  *status (may be NULL) receives
 * TAMCMC_ERR_BAD_MODEL for prior ids / model families this build does not carry. */
double tamcmc_log_prior(int prior_class, const double *params, int64_t Nparams, const int32_t *plength, const double *priors,
                        const int32_t *priors_switch, const double *extra_priors, int32_t n_extra, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
