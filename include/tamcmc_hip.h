/*
 * tamcmc_hip.h -- C ABI of the MI355X (gfx950) hot path of TAMCMC:
 *   per-chain Lorentzian-sum model over the power-spectrum bins
 *   -> chi^2(2 d.o.f.) log-likelihood reduction
 *   -> finite-difference gradient for the Langevin proposal.
 *
 * Plain C: opaque context, plain pointers and sizes, int status codes
 * (never exit()).  All pointers are HOST pointers unless a name ends in _dev.
 * One context per host thread / GPU; a context is not thread-safe.
 *
 * The reference (OthmanB/TAMCMC-C, paths relative to its root) has no FFI:
 * its boundary is the in-process call
 *     Model_def::generate_model(Data*, m, Tcoefs)      tamcmc/sources/model_def.cpp:466-482
 * made once per chain per iteration from
 *     MALA::update_position_MH                         tamcmc/sources/MALA.cpp:486-488
 * inside `#pragma omp parallel for` over chains        tamcmc/sources/MALA.cpp:648-668.
 * Each entry point below names the reference interface it replaces.
 * INTEGRATION.md shows the reference-side binding.
 */
#ifndef TAMCMC_HIP_H
#define TAMCMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- status codes ---------------- */
#define TAMCMC_OK 0
#define TAMCMC_ERR_HIP (-1)          /* a HIP runtime call failed (tamcmc_hip_last_error has the text) */
#define TAMCMC_ERR_EMPTY_WINDOW (-2) /* set_imin_imax: imax-imin<=0 (reference exits, build_lorentzian.cpp:650-665) */
#define TAMCMC_ERR_NAN_WINDOW (-3)   /* NaN width/splitting: no window regime applies (build_lorentzian.cpp:597-634) */
#define TAMCMC_ERR_BAD_MODEL (-4)    /* model id without a device table builder (model_def.cpp:352-385) */
#define TAMCMC_ERR_BAD_ARG (-5)
#define TAMCMC_ERR_NO_SPECTRUM (-6)
#define TAMCMC_ERR_NO_DEVICE (-7)    /* no HIP device: the product path has NO CPU fallback */

/* ---------------- model ids (Config/default/models_ctrl.list) ---------------- */
#define TAMCMC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC 3 /* model_MS_Global_a1etaa3_HarveyLike_Classic, models.cpp:1943 */
#define TAMCMC_MODEL_MS_LOCAL_BASIC 11           /* model_MS_local_basic, models.cpp:3012 */
#define TAMCMC_MODEL_MS_GLOBAL_AJ 23             /* model_MS_Global_aj_HarveyLike, models.cpp:1195 */
#define TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4 25 /* model_RGB_asympt_aj_AppWidth_HarveyLike_v4, models.cpp:4684: only through
                                                    tamcmc_hip_loglike_params_batch (its table needs the device pre-step:
                                                    ARMM mixed-mode solver + zeta function, csrc/rgb_prestep.hip) */
#define TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4 27 /* model_RGB_asympt_aj_CteWidth_HarveyLike_v4, models.cpp:4334: same path, one
                                                    constant width for the l=0,2,3 modes */

/* ---------------- arithmetic modes ---------------- */
/* STRICT: per-bin operation order of the reference (IEEE divides, no FMA contraction): the model row is
 *         bit-identical to the CPU restatement when the Harvey pow() terms are inactive.
 * FAST  : same function, re-associated (common-denominator multiplet sum, reciprocal+Newton, exp/log Harvey) and,
 *         for multiplets far from a tile, summed as ONE degree-15 polynomial per tile (truncation <= 8^-16 of the far
 *         term); stated tolerance: |dM|/M <= 1e-12 per bin, |dlogL|/|logL| <= 1e-11. */
#define TAMCMC_PRECISION_STRICT 0
#define TAMCMC_PRECISION_FAST 1        /* far-field expansion per tile + direct near field */
#define TAMCMC_PRECISION_FAST_DIRECT 2 /* FAST arithmetic, every component evaluated per bin (no far field) */

#define TAMCMC_OPT_PRECISION 1   /* value: TAMCMC_PRECISION_* (default STRICT) */
#define TAMCMC_OPT_TIMING 2      /* value: 0/1 -- bracket the likelihood kernel with HIP events on the context stream */
#define TAMCMC_OPT_BINS_PER_THREAD 3 /* value: 1,2,4 (workgroup 256) or 4,8,16 (workgroup 64) -- tile = workgroup*value bins */
#define TAMCMC_OPT_FD_WINDOWED 5     /* value: 0/1 -- FAST modes: FD gradients from delta tables (only the multiplets a perturbation
                                        changes, on their windows, against the stored base model row); default 1 */
#define TAMCMC_OPT_WORKGROUP 4       /* value: 256 (four waves share a tile) or 64 (one wave per tile); resets bins per thread */
#define TAMCMC_OPT_STEP_SCHEME 6     /* device-resident sampler: 0 = automatic (fused launches wherever no adaptation separates two
                                        iterations -- one launch per iteration, or one per chain group on two streams once a launch no
                                        longer fits the GPU's resident waves -- lockstep kernels elsewhere), 1 = lockstep kernels only,
                                        2 = fused with one launch per iteration, 3 = fused with two chain groups whenever there are
                                        8 chains or more.  Same chains bit for bit in every case (tests/test_gpu_sampler.py); default 0 */
#define TAMCMC_OPT_ARMM_DENSE_SCAN 7 /* red-giant pre-step: 1 = walk the solver's whole grid like the reference (solver_mm.cpp:340-377)
                                        instead of the pole-structured scan that finds the same cells; default 0 */

/* One (n,l) multiplet: <=7 Lorentzian m-components on its truncation window.
 * This is the flat "mode table" row every Lorentzian model of the dispatch table reduces to
 * (build_lorentzian.cpp:131-161, :208-246).  152 bytes, no padding. */
typedef struct tamcmc_multiplet {
    int32_t l;      /* degree 0..3 -> 2l+1 components */
    int32_t i0;     /* first bin of the window (set_imin_imax, build_lorentzian.cpp:645-649) */
    int32_t i1;     /* one past the last bin */
    int32_t flags;  /* reserved, 0 */
    double fc;      /* central frequency nu_c (asymmetry reference, build_lorentzian.cpp:240) */
    double gamma;   /* width */
    double asym;    /* asymmetry coefficient (0 = symmetric Lorentzian) */
    double nu[7];   /* nu_nlm for m=-l..l */
    double hv[7];   /* H_l * V_m */
} tamcmc_multiplet;

typedef struct tamcmc_hip_ctx tamcmc_hip_ctx;

/* ---------------- context ---------------- */
int tamcmc_hip_create(tamcmc_hip_ctx **ctx, int device);
void tamcmc_hip_destroy(tamcmc_hip_ctx *ctx);
const char *tamcmc_hip_last_error(const tamcmc_hip_ctx *ctx);
int tamcmc_hip_set_option(tamcmc_hip_ctx *ctx, int option, int64_t value);
const char *tamcmc_hip_version(void);

/* Page-locked host memory for buffers the library copies results into (recorded samples and statistics of tamcmc_sampler_run): any
 * host pointer works there, a pinned one is filled by an asynchronous DMA instead of a staged copy.  NULL on failure. */
void *tamcmc_hip_host_alloc(size_t bytes);
void tamcmc_hip_host_free(void *p);

/* Replaces the shared read-only `Data{x,y,Nx}` (tamcmc/headers/data.h:23-34) every chain reads:
 * uploads the spectrum once; it stays resident in HBM. x must be a regular grid (build_lorentzian.cpp:645). */
int tamcmc_hip_set_spectrum(tamcmc_hip_ctx *ctx, const double *x, const double *y, int64_t Nx);

/* Replaces, for B parameter vectors at once, the per-bin work of
 *   call_model  (model_def.cpp:220-388 -> optimum_lorentzian_calc_* + harvey_like, noise_models.cpp:15-39)
 *   call_likelihood (model_def.cpp:390-419 -> likelihood_chi22p, likelihoods.cpp:17-28).
 * mults[offsets[b] .. offsets[b+1]) are evaluation b's multiplets in the reference's accumulation order;
 * noise + b*noise_stride = the nnoise[b] values |noise params| = [H0,tau0,p0, H1,tau1,p1, ..., N0] of evaluation b:
 * nharvey[b] Harvey triples are applied (noise_models.cpp:29-36), the white noise N0 is the LAST of the nnoise[b] entries;
 * Tcoefs[b] = temperature (NULL -> 1); p = likelihood_params truncated to long.
 * Out: logL[b] = -p * sum_i(y_i/M_i + ln M_i) / Tcoefs[b]; model (may be NULL) = B x Nx rows.
 * A non-finite model gives a NaN/inf logL that the caller rejects (MALA.cpp:490,522-524). */
int tamcmc_hip_loglike_batch(tamcmc_hip_ctx *ctx, int B, const tamcmc_multiplet *mults, const int32_t *offsets,
                             const double *noise, int noise_stride, const int32_t *nharvey, const int32_t *nnoise,
                             const double *Tcoefs, double p, double *logL, double *model);

/* Table builders: the host-side scalar part of the model functions
 *   VectorXd model_X(params, params_length, x, outparams)   tamcmc/headers/models.h:21-57
 * (parameter unpack, amplitude_ratio, lin_interpol, eta0, set_imin_imax) for ids 3, 11, 23.
 * Writes at most max_mults rows; *n_mults = rows needed.  noise_abs receives |noise params| (plength[8] values). */
int tamcmc_build_mode_table(int model_id, const double *params, const int32_t *plength, const double *x, int64_t Nx,
                            tamcmc_multiplet *mults, int max_mults, int *n_mults, double *noise_abs,
                            int *nharvey, int *nnoise);

/* model id + params level: table build on the host for each of the B vectors, then one batched device call.
 * This is the batched body of Model_def::generate_model without the prior (model_def.cpp:473-474).
 * status (may be NULL) receives the per-vector table status; vectors with a failed table get logL = NaN. */
int tamcmc_hip_loglike_params_batch(tamcmc_hip_ctx *ctx, int model_id, int B, const double *params, int64_t Nparams,
                                    const int32_t *plength, const double *Tcoefs, double p, double *logL,
                                    double *model, int32_t *status);

/* Red-giant tolerance (ids 25, 27; every arithmetic mode).  The reference evaluates the mixed-mode relation on its grids in double
 * (Eigen arrays, solver_mm.cpp:158-169, called at :356-358 and :382-383) and the intersection test in long double (:179-186, :389-404); the device pre-step uses double
 * throughout, the oracle long double throughout.  Stated: mixed-mode frequencies within 1e-10 muHz and zeta within 1e-9 of the oracle's,
 * model rows ||dM||_2 / ||M||_2 <= 1e-10, |dlogL| / |logL| <= 1e-11 (STRICT) / the FAST tolerance above (FAST).  Measured on MI355X
 * (tools/rgb_parity_probe.py, round 3): frequencies <= 3e-13 muHz (1.5e-11 at 2e5 bins with ~170 mixed modes), zeta <= 4e-13 (1.2e-10),
 * rows <= 2e-12 (1.1e-11), logL <= 1e-14.  A mixed mode is as narrow as 0.01 muHz, so a frequency error d nu moves single bins of its
 * profile by ~d nu / Gamma: the per-bin maximum is larger than the row norm (5e-12 typical, 5.5e-10 at the C5 size).
 *
 * Red-giant models (ids 25, 27): the l=1 mixed modes of ONE parameter vector as the device pre-step computes them for the table
 * (csrc/rgb_prestep.hip) -- what external/ARMM/do_solve.cpp:114-121 prints with the reference's solver:
 *   nu_m  = solve_mm_asymptotic_O2p / _O2from_l0 (solver_mm.cpp:470-760, chosen by the vector's model_type) + the spline bias,
 *   zeta  = ksi_fct2(nu_m, ..., "precise") (bump_DP.cpp:125-188),  h1_h0 = h_l_rgb(zeta, Hfactor) (bump_DP.cpp:235-254).
 * The solver's step is the spectrum's resolution x[2]-x[1] (models.cpp:4719).  Any of nu_m / zeta / h1_h0 ([max_modes]) may be NULL;
 * *n_modes = number of mixed modes found. */
int tamcmc_hip_rgb_mixed_modes(tamcmc_hip_ctx *ctx, int model_id, const double *params, int64_t Nparams, const int32_t *plength,
                               int max_modes, double *nu_m, double *zeta, double *h1_h0, int *n_modes);

/* Forward-difference gradient of the tempered logL (the drift MALA::D_MALA leaves as a stub, MALA.cpp:321-328):
 * for each of the C chains, Nvars+1 evaluations in ONE batched launch.
 * params: C x Nparams; index_to_relax: Nvars parameter indices (model_def.cpp:76-90); hstep: Nvars steps.
 * Out: logL0[C], grad[C x Nvars] = (logL(theta + h e_k) - logL(theta)) / h_applied.
 * Tolerance (FAST arithmetic, windowed differences: TAMCMC_OPT_FD_WINDOWED): the difference is formed term by term against the stored
 * base point -- per bin as the series in u = dM/M0 (five terms, closed form beyond |u| = 0.01), and on tiles where every changed
 * multiplet is in the far field from moments of the base point (first and second order in u; used where max|u| <= 1e-5, what is
 * omitted is below 1e-10 of the leading term).  Against the brute-force difference of two full evaluations it agrees to that
 * difference's own cancellation noise (~5e-15 Nx / h) + 1e-6 of the gradient's scale (tests/test_gpu_parity.py). */
int tamcmc_hip_fd_gradient(tamcmc_hip_ctx *ctx, int model_id, int C, const double *params, int64_t Nparams,
                           const int32_t *plength, const int32_t *index_to_relax, int Nvars, const double *hstep,
                           const double *Tcoefs, double p, double *logL0, double *grad);

/* Same batch, gradient of the tempered log-POSTERIOR: each of the C*(Nvars+1) workgroups also evaluates the log-prior of
 * its perturbed vector on the device (prior_class 2 = io_MS_Global, 3 = io_local; priors = 4 x Nparams row-major table,
 * priors_switch = primitive ids, extra_priors[10]: Input_Data of tamcmc/headers/data.h:51-62).  Where the forward point
 * leaves a prior's support the backward difference of the prior is used, else that prior term is flat.
 * Out: logL0[C] (tempered), logPr0[C] (may be NULL), grad[C x Nvars], grad_prior[C x Nvars] (may be NULL: the prior's
 * share of grad, so that a caller can re-temper the likelihood share after a parallel-tempering swap). */
int tamcmc_hip_fd_gradient_posterior(tamcmc_hip_ctx *ctx, int model_id, int prior_class, int C, const double *params,
                                     int64_t Nparams, const int32_t *plength, const int32_t *index_to_relax, int Nvars,
                                     const double *hstep, const double *Tcoefs, double p, const double *priors,
                                     const int32_t *priors_switch, const double *extra_priors, double *logL0, double *logPr0,
                                     double *grad, double *grad_prior);

/* Timing of the likelihood kernel measured with HIP events on the context's own stream
 * (enabled by TAMCMC_OPT_TIMING): totals since the last reset. */
int tamcmc_hip_get_kernel_stats(tamcmc_hip_ctx *ctx, double *kernel_ms_total, int64_t *launches,
                                int64_t *evaluations);
int tamcmc_hip_reset_kernel_stats(tamcmc_hip_ctx *ctx);
/* Windowed finite differences (TAMCMC_OPT_FD_WINDOWED, timing enabled): since the last reset, the bins inside the affected ranges of
 * the delta evaluations and the number of those evaluations -- the bytes such a launch really touches are 24 B per affected bin
 * (x, y, base model row), not 16 B x Nx per evaluation. */
int tamcmc_hip_get_fd_stats(tamcmc_hip_ctx *ctx, int64_t *affected_bins, int64_t *delta_evaluations);
/* ... and how many of those delta evaluations were "full tables": a perturbation that moves most multiplets (a splitting coefficient,
 * the asymmetry) is evaluated as the whole perturbed table minus the stored base model row instead of +new / -old row pairs
 * (FAST arithmetic only; same tolerance as the pair tables: the unchanged rows cancel exactly). */
int tamcmc_hip_get_fd_full_tables(tamcmc_hip_ctx *ctx, int64_t *full_table_evaluations);

#ifdef __cplusplus
}
#endif
#endif /* TAMCMC_HIP_H */
