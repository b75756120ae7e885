/*
 * tamcmc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A from-scratch plain-C restatement of the TAMCMC-C hot path
 *   Model_def::generate_model = prior -> call_model -> call_likelihood
 * used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / CPU baseline.  The product path (tamcmc-c_amd/) never
 * includes, links or calls anything declared here.
 *
 * Pinning status: the reference (/root/reference) cannot be compiled in the
 * authoring container (every hot-path TU needs Eigen3; build_lorentzian.h pulls
 * GSL and Boost) and its tests hold no stored numbers.  The oracle is pinned by
 *   (i)  tests/golden/acoefs_py.json : Pslm / nu_nlm values produced by the
 *        reference's own importable python helper test/lorentzian_test/acoefs.py
 *        (generator: tests/golden/make_acoefs_golden.py),
 *   (ii) analytic known-answer tests (tests/test_oracle_kat.py),
 *   (iii) the reference's own acceptance bound ||new-ref||_2 <= 1e-8
 *        (test/lorentzian_test/unit_tests/test_build_l_mode.cpp:104,134).
 * Everything not covered by (i)-(iii) is "parity unpinned by the reference".
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose meaning it restates.
 */
#ifndef TAMCMC_ORACLE_H
#define TAMCMC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define ORC_OK 0
#define ORC_ERR_EMPTY_WINDOW (-2)  /* build_lorentzian.cpp:650-665 (reference exits) */
#define ORC_ERR_NAN_WINDOW (-3)    /* SURVEY App. D caveat: NaN gamma/f_s leaves pvals unset */
#define ORC_ERR_BAD_MODEL (-4)     /* model_def.cpp:352-385 (reference exits) */
#define ORC_ERR_BAD_ARG (-5)

/* model ids = Config/default/models_ctrl.list */
#define ORC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC 3
#define ORC_MODEL_MS_LOCAL_BASIC 11
#define ORC_MODEL_MS_GLOBAL_AJ 23
#define ORC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4 25
#define ORC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4 27

/* ---- scalar helpers ---- */
long double orc_Pslm(int s, int l, int m);               /* acoefs.cpp:51-110 */
long double orc_Hslm_Ritzoller1991(int s, int l, int m); /* acoefs.cpp:19-49 */
double orc_Qlm(int l, int m);                            /* build_lorentzian.cpp:583-592 */
void orc_amplitude_ratio(int l, double beta_deg, double *V /*2l+1*/); /* function_rot.cpp:15-41 */
double orc_lin_interpol(const double *x, const double *y, long n, double x_int); /* interpol.cpp:13-43 */
void orc_linfit(const double *x, const double *y, long n, double out[2]);        /* linfit.cpp:17-35 */
double orc_eta0_from_dnu(double dnu);                    /* models.cpp:6073-6084 */
double orc_eta0_fct(const double *fl0, long n);          /* models.cpp:6065-6071 */
int orc_set_imin_imax(const double *x, long Nx, int l, double fc_l, double gamma_l, double f_s,
                      double c, double step, int ivals[2]); /* build_lorentzian.cpp:595-676 */

/* ---- multiplet builders on a window x_l[0..N) -> result[0..N) ---- */
void orc_build_l_mode_a1etaa3(const double *x_l, long N, double H_l, double fc_l, double f_s, double eta0,
                              double a3, double asym, double gamma_l, int l, const double *V,
                              double *result); /* build_lorentzian.cpp:131-161 */
void orc_build_l_mode_aj(const double *x_l, long N, double H_l, double fc_l, double a1, double a2, double a3,
                         double a4, double a5, double a6, double eta0, double asym, double gamma_l, int l,
                         const double *V, double *result); /* build_lorentzian.cpp:208-246 */
/* split-frequency helpers exposed for the golden-vector tests */
double orc_nu_nlm_aj(double fc_l, double a1, double a2, double a3, double a4, double a5, double a6,
                     double eta0, int l, int m); /* build_lorentzian.cpp:226-229 */
double orc_nu_nlm_a1etaa3(double fc_l, double f_s, double eta0, double a3, int l, int m); /* :145 */

/* windowed accumulate: model[i0..i1) += multiplet */
int orc_optimum_lorentzian_calc_a1etaa3(const double *x, double *model, long Nx, double H_l, double fc_l,
                                        double f_s, double eta0, double a3, double asym, double gamma_l,
                                        int l, const double *V, double step, double c); /* :441-458 */
int orc_optimum_lorentzian_calc_aj(const double *x, double *model, long Nx, double H_l, double fc_l, double a1,
                                   double a2, double a3, double a4, double a5, double a6, double eta0,
                                   double asym, double gamma_l, int l, const double *V, double step,
                                   double c); /* :502-522 + caller add models.cpp:1297-1298 */

/* ---- background and likelihood ---- */
void orc_harvey_like(const double *noise_params_abs, long Nnoise, const double *x, double *model, long Nx,
                     int Nharvey); /* noise_models.cpp:15-39 (in place: model += background) */
long double orc_likelihood_chi22p(const double *y, const double *model, long Nx, long p); /* likelihoods.cpp:17-28 */
/* same quantity with 80-bit accumulators: the "truth" used to size tolerances */
long double orc_likelihood_chi22p_ld(const double *y, const double *model, long Nx, long p);

/* ---- model functions: params/plength -> model[Nx] (models.h:21-57) ---- */
int orc_model_MS_Global_aj_HarveyLike(const double *params, const int *plength, const double *x, long Nx,
                                      double *model); /* models.cpp:1195-1408 */
int orc_model_MS_Global_a1etaa3_HarveyLike_Classic(const double *params, const int *plength, const double *x,
                                                   long Nx, double *model); /* models.cpp:1943-2121 */
int orc_model_MS_local_basic(const double *params, const int *plength, const double *x, long Nx,
                             double *model); /* models.cpp:3012-3195 */
int orc_call_model(int model_id, const double *params, const int *plength, const double *x, long Nx,
                   double *model); /* model_def.cpp:220-388 */

/* ---- red-giant model and its host pre-step (armm_oracle.c; parity unpinned, see that file's header) ---- */
typedef struct orc_eigensols {  /* Data_eigensols, external/ARMM/data_solver.h */
    long n_m, n_p, n_g;
    double *nu_m, *nu_p, *nu_g, *dnup, *dPg;
} orc_eigensols;
void orc_eigensols_free(orc_eigensols *e);
int orc_armm_solve_O2p(double Dnu_p, double epsilon, int el, double delta0l, double alpha_p, double nmax, double DPl, double alpha,
                       double q, double fmin, double fmax, double resol, orc_eigensols *out);      /* solver_mm.cpp:470-611 */
int orc_armm_solve_O2from_l0(const double *nu_l0, long n0, int el, double delta0l, double DPl, double alpha, double q, double resol,
                             double freq_min, double freq_max, orc_eigensols *out);               /* solver_mm.cpp:624-760 */
void orc_ksi_fct2_precise(const double *nu, long n, const double *nu_p, const double *Dnu_p, long Lp, const double *nu_g,
                          const double *DPl, long Lg, double q, double *ksi);                      /* bump_DP.cpp:125-188 */
double orc_spline_eval(const double *xn, const double *yn, long n, int type /*1 cubic, 2 Hermite*/, double x); /* spline.h:242-498 */
typedef struct orc_rgb_modes {  /* what the model function derives before it sums Lorentzians (models.cpp:4770-4913) */
    long N0, N1;
    double *fl0, *Wl0, *Hl0;                    /* radial modes */
    double *fl1, *Wl1, *Hl1, *a1_l1, *ksi;      /* l=1 mixed modes: frequency (+bias), width, height, splitting, zeta */
    double g[6];                                /* |width-law parameters| */
} orc_rgb_modes;
int orc_rgb_v4_modes(const double *params, const int *plength, double step, orc_rgb_modes *out);
int orc_rgb_v4_cte_modes(const double *params, const int *plength, double step, orc_rgb_modes *out);  /* constant-width variant */
void orc_rgb_modes_free(orc_rgb_modes *m);
int orc_model_RGB_asympt_aj_AppWidth_HarveyLike_v4(const double *params, const int *plength, const double *x, long Nx,
                                                   double *model);                                  /* models.cpp:4684-5079 */
int orc_model_RGB_asympt_aj_CteWidth_HarveyLike_v4(const double *params, const int *plength, const double *x, long Nx,
                                                   double *model);                                  /* models.cpp:4334-4682 */

/* call_likelihood (model_def.cpp:390-419): chi22p / Tcoef */
double orc_call_likelihood(const double *y, const double *model, long Nx, double likelihood_params, double Tcoef);

/* batched driver used for parity and as CPU baseline: for each b, model -> tempered logL.
 * OpenMP over b like MALA.cpp:648.  model_out may be NULL.  Returns first nonzero status. */
int orc_loglike_batch(int model_id, int B, const double *params /*B x Nparams*/, long Nparams,
                      const int *plength, const double *x, const double *y, long Nx, double likelihood_params,
                      const double *Tcoefs /*B*/, double *logL /*B*/, double *model_out /*B x Nx or NULL*/,
                      int *status /*B or NULL*/);

/* forward-difference gradient of the tempered logL wrt the relaxed parameters
 * (oracle of the path the reference leaves as a stub, MALA.cpp:321-337) */
int orc_fd_gradient(int model_id, const double *params, long Nparams, const int *plength,
                    const int *index_to_relax, int Nvars, const double *hstep /*Nvars*/, const double *x,
                    const double *y, long Nx, double likelihood_params, double Tcoef, double *logL0,
                    double *grad /*Nvars*/);

/* ---- evidence diagnostic (interpol.cpp:46-101; diagnostics.cpp:980-1019) ---- */
void orc_quad_interpol(const double *a, long n, long m, double *b);
double orc_evidence_calc(const double *Tcoefs, long Nchains, const double *Likelihoods /*[rows x Nchains]*/, long rows, int interpol_factor,
                         double *beta, double *L_beta, double *beta_interp, double *L_beta_interp);

/* ---- priors (stats_dictionary.cpp; priors_calc.cpp:725-870) ---- */
long double orc_logP_uniform(double b_min, double b_max, double x);                /* stats_dictionary.cpp:38-52 */
long double orc_logP_gaussian(double mean, double sigma, double x);                /* :98-105 */
long double orc_logP_jeffrey(double hmin, double hmax, double h);                  /* :127-143 */
long double orc_logP_uniform_abs(double b_min, double b_max, double x);            /* :56-70 */
long double orc_logP_jeffrey_abs(double hmin, double hmax, double h);              /* :149-165 */
long double orc_logP_gaussian_uniform(double b_min, double b_max, double sigma, double x);          /* GU  */
long double orc_logP_uniform_gaussian(double b_min, double b_max, double sigma, double x);          /* UG  */
long double orc_logP_gug(double b_min, double b_max, double sigma1, double sigma2, double x);       /* GUG */
/* apply_generic_priors over parameters [i0, i0+n): priors table is 4 x Nparams row-major */
long double orc_apply_generic_priors(const double *params, long i0, long n, const double *priors, long Nparams,
                                     const int *priors_names_switch);

/* model-class priors (priors_calc.cpp:27-317, :514-629) and call_prior (model_def.cpp:421-464) */
long double orc_priors_MS_Global(const double *params, const int *plength, const double *priors, const int *priors_names_switch,
                                 const double *extra_priors);
long double orc_priors_local(const double *params, const int *plength, const double *priors, const int *priors_names_switch,
                             const double *extra_priors);
long double orc_priors_asymptotic(const double *params, const int *plength, const double *priors, const int *priors_names_switch,
                                  const double *extra_priors);   /* priors_calc.cpp:319-512 */
double orc_call_prior(int prior_class, const double *params, const int *plength, const double *priors,
                      const int *priors_names_switch, const double *extra_priors);

/* ---- one sampler iteration with explicit random draws (sampler_oracle.c; MALA.cpp:135-176, :296-319, :339-369, :397-461, :463-553,
 *      :645-703; model_def.cpp:466-482) ---- */
typedef struct orc_sampler_star {  /* what Config::setup hands the sampler: the star, its model class, the !MALA scalars */
    int model_id, prior_class;
    long Nparams, Nvars, Nx, Nchains;
    const int *plength, *index_to_relax, *priors_switch;
    const double *priors, *extra_priors, *x, *y, *Tcoefs, *init_logL;
    double likelihood_params, epsilon1, epsi2, A1, target_acceptance, c0;
} orc_sampler_star;
long double orc_p1_fct(long double x, long double epsilon1, long double A1);
void orc_p2_fct(double *M, long n, double A1);
void orc_p3_fct(double *v, long n, double A1);
void orc_update_proposal(double *mu, double *covarmat, double *sigma, const double *vars, long Nvars, long double acceptance,
                         long double gamma, long double target_acceptance, long double epsilon1, long double A1);
int orc_new_prop_values(const double *covarmat, double sigma, double epsi2, const double *vars, const double *z, long Nvars, double *out,
                        double *L_out);
int orc_mh_accept(double logL_prop, double logPost_prop, double logPost_cur, double u, double *r_out);
int orc_parallel_tempering(double *logL, double *logPrior, double *logPost, double *vars, double *params, int *moved, double *Pmove,
                           const double *Tcoefs, long Nvars, long Nparams, int ind_A, double u, int literal_444, double *Pswap);
int orc_learn_at(long i, const long *Nt_learn, const long *periods_learn, long n_periods);
void orc_generate_model(const orc_sampler_star *S, const double *params, double Tcoef, double init_logL, double *model_scratch, double *logL,
                        double *logPr, double *logPost);
int orc_sampler_iteration(const orc_sampler_star *S, long i, int learn, int do_swap, int ind_A, double u_swap, int literal_444, const double *z,
                          const double *u_mh, double *params, double *vars, double *logL, double *logPrior, double *logPost, int *moved,
                          double *Pmove, double *mu, double *covarmat, double *sigma, int *swapped, double *prop_vars, double *prop_stats);

/* ---- the Langevin step (use_drift = 1; the reference's D_MALA / multinormal_logpdf are stubs, MALA.cpp:321-337): sampler_oracle.c ---- */
long double orc_mvn_logpdf(const double *v, const double *mean, const double *M, long n);
int orc_fd_gradient_posterior(const orc_sampler_star *S, const double *params, double Tcoef, const double *h, double *grad, double *gradP);
void orc_langevin_drift(const double *covarmat, double sigma, double epsi2, double delta, const double *grad, long Nvars, double *drift);
int orc_langevin_iteration(const orc_sampler_star *S, long i, int learn, int do_swap, int ind_A, double u_swap, int literal_444, const double *z,
                           const double *u_mh, double fd_step_rel, double delta, double *params, double *vars, double *logL, double *logPrior,
                           double *logPost, int *moved, double *Pmove, double *mu, double *covarmat, double *sigma, int *swapped,
                           double *prop_vars, double *prop_stats, double *diag, const int *chain_mask, const double *prop_given);

#ifdef __cplusplus
}
#endif
#endif
