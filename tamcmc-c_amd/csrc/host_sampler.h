// host_sampler.h -- host-side mirror of the reference's sampler-facing classes (product code, C++17, no Eigen).
//
//   Config      (the subset of tamcmc/headers/config.h the hot path reads: MALA.*, modeling.*, data.data, outputs.*)
//   Data        tamcmc/headers/data.h:23-34
//   Input_Data  tamcmc/headers/data.h:51-62
//   Model_def   tamcmc/headers/model_def.h:27-85   (same public state, same method names)
//   MALA        tamcmc/headers/MALA.h:26-69        (same method names; D_MALA / multinormal_logpdf implemented)
//
// Difference by design: Model_def gains generate_models_batch(): all chains' proposals are unpacked on the host
// and evaluated by ONE batched device call (tamcmc_hip_loglike_batch) instead of the reference's per-chain
// OpenMP fan-out (MALA.cpp:648-668).  generate_model(m) keeps the old per-chain call shape.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tamcmc_hip.h"

namespace tamcmc {

struct Matrix {  // dense row-major
    long rows = 0, cols = 0;
    std::vector<double> a;
    Matrix() = default;
    Matrix(long r, long c, double v = 0.0) : rows(r), cols(c), a((size_t)(r * c), v) {}
    double &operator()(long i, long j) { return a[(size_t)(i * cols + j)]; }
    double operator()(long i, long j) const { return a[(size_t)(i * cols + j)]; }
    double *row(long i) { return a.data() + (size_t)(i * cols); }
    const double *row(long i) const { return a.data() + (size_t)(i * cols); }
};

struct Data {  // data.h:23-34
    std::vector<double> x, y, sigma_y;
    long Nx = 0;
};

struct Input_Data {  // data.h:51-62
    std::string model_fullname;
    std::vector<std::string> inputs_names;
    std::vector<int> priors_names_switch;
    std::vector<double> inputs;
    std::vector<int> relax;
    Matrix priors;  // 4 x Nparams
    std::vector<int> plength;
    std::vector<double> extra_priors;
};

struct Config {
    struct {
        int Nchains = 5;                      // config_default.cfg:27
        double target_acceptance = 0.234, c0 = 10, epsilon1 = 1e-12, epsi2 = 1e-12, A1 = 1e14;
        double delta = 0, delta_x = 1e-10, lambda_temp = 3.5;
        std::vector<long> Nt_learn{1000, 1500, 100000};
        std::vector<long> periods_learn{1, 1};
        long dN_mixing = 1;
        int use_drift = 0;
        std::string proposal_type = "Random";
        std::vector<std::string> var_names_errors;  // errors_default.cfg
        std::vector<double> fraction_errors, offset_errors;
        // additions of this build (no reference counterpart)
        uint64_t seed = 20240229;             // reference: time(NULL), MALA.cpp:62
        double fd_step_rel = 1e-7;            // forward-difference step h_k = fd_step_rel * max(|theta_k|, 1e-3)
        int swap_rule = 0;                    // 1: chain B's logPosterior after a swap as MALA.cpp:433,444 execute it (see tamcmc_sampler.h)
    } MALA;
    struct {
        int model_fct_name_switch = 0, likelihood_fct_name_switch = 0, prior_fct_name_switch = 2;
        double likelihood_params = 1;
        Input_Data inputs;
    } modeling;
    struct { Data data; } data;
    struct { long Nsamples = 150, Nbuffer = 10000; } outputs;
};

// ---- priors (stats_dictionary.cpp, priors_calc.cpp) ----
long double logP_uniform(long double b_min, long double b_max, long double x);
long double logP_uniform_abs(long double b_min, long double b_max, long double x);
long double logP_gaussian(long double mean, long double sigma, long double x);
long double logP_jeffrey(long double hmin, long double hmax, long double h);
long double logP_jeffrey_abs(long double hmin, long double hmax, long double h);
long double logP_uniform_gaussian(long double b_min, long double b_max, long double sigma, long double x);
long double logP_gaussian_uniform(long double b_min, long double b_max, long double sigma, long double x);
long double logP_gaussian_uniform_gaussian(long double b_min, long double b_max, long double s1, long double s2, long double x);
long double apply_generic_priors(const double *params, long Nparams, const Matrix &priors_params,
                                 const std::vector<int> &priors_names_switch, int *status);
long double priors_MS_Global(const double *params, const std::vector<int> &plength, const Matrix &priors_params,
                             const std::vector<int> &priors_names_switch, const std::vector<double> &extra_priors, int *status);
long double priors_local(const double *params, const std::vector<int> &plength, const Matrix &priors_params,
                         const std::vector<int> &priors_names_switch, const std::vector<double> &extra_priors, int *status);
long double priors_asymptotic(const double *params, const std::vector<int> &plength, const Matrix &priors_params,
                         const std::vector<int> &priors_names_switch, const std::vector<double> &extra_priors, int *status);  // priors_calc.cpp:319-512

class Model_def {  // model_def.h:27-85
    std::vector<double> cons;
    long Ncons = 0, Nvars = 0, Nparams = 0, Nmodels = 0;
    std::vector<int> index_to_relax;
    int model_fct_name_switch = 0, likelihood_fct_name_switch = 0, prior_fct_name_switch = 0;
    std::vector<int> priors_params_names_switch;
    Matrix priors_params;
    double likelihood_params = 1;
    std::vector<int> relax, plength;
    std::vector<double> extra_priors;
    tamcmc_hip_ctx *ctx = nullptr;  // not owned

  public:
    Matrix params, vars;
    std::vector<double> logLikelihood, init_logLikelihood, logPrior, logPosterior, Pmove;
    std::vector<char> moved;
    bool swaped = false;
    long double Pswap = 0;
    std::vector<double> comparator_MH;
    long double comparator_PT = 0;
    int last_status = TAMCMC_OK;

    Model_def(Config *config, const std::vector<double> &Tcoefs, bool verbose, tamcmc_hip_ctx *ctx);
    long get_Nvars() const { return Nvars; }
    long get_Nparams() const { return Nparams; }
    const std::vector<int> &get_index_to_relax() const { return index_to_relax; }
    const std::vector<int> &get_plength() const { return plength; }
    int get_model_id() const { return model_fct_name_switch; }
    double get_likelihood_params() const { return likelihood_params; }
    tamcmc_hip_ctx *get_ctx() const { return ctx; }
    int get_prior_class() const { return prior_fct_name_switch; }
    const Matrix &get_priors() const { return priors_params; }
    const std::vector<int> &get_priors_switch() const { return priors_params_names_switch; }
    const std::vector<double> &get_extra_priors() const { return extra_priors; }

    std::vector<double> call_model(Data *data_struc, int m);                        // model_def.cpp:220
    void update_params_with_vars(long m);                                            // model_def.cpp:484
    long double call_prior(Data *data_struc, int m);                                 // model_def.cpp:421
    long double call_prior_params(const double *p);                                  // same switch on an explicit vector
    long double generate_model(Data *data_struc, long m, const std::vector<double> &Tcoefs);  // model_def.cpp:466
    // batched body: priors for every chain, then ONE device call for all chains with a finite prior
    int generate_models_batch(Data *data_struc, const std::vector<double> &Tcoefs);
};

class MALA {  // MALA.h:26-69
    uint64_t seed;
    long Nsamples, Nchains, Nvars;
    std::vector<long> Nt_learn, periods_learn;
    long dN_mixing;
    long double epsilon1, A1, delta, delta_x, c0, gamma, lambda_temp, target_acceptance;
    double epsi2;
    bool use_drift;
    double fd_step_rel;
    int swap_rule = 0;
    std::vector<Matrix> Lchol;          // cached factor of (covarmat+epsilon2)*sigma per chain
    std::vector<char> Lchol_valid;
    // Langevin state (use_drift): gradient of the tempered log-posterior at the current / proposed position
    Matrix grad_cur, grad_prop;
    Matrix gradP_cur, gradP_prop;       // the prior's share of those gradients (not tempered)
    std::vector<char> grad_valid;

  public:
    std::vector<double> sigma;
    Matrix mu;
    std::vector<double> Tcoefs;
    std::vector<Matrix> covarmat;
    long iteration = 0;                 // the reference's loop counter i (MALA.cpp:623)
    long Nswap_attempts = 0, Nswap_accepted = 0;

    explicit MALA(Config *cfg);
    void init_proposal(const std::vector<double> &vars, const std::vector<std::string> &var_names,
                       const std::vector<std::string> &s_inerror, const std::vector<double> &fracerr,
                       const std::vector<double> &offseterr);                       // MALA.cpp:246
    void update_proposal(const double *vars, long double acceptance, int m);        // MALA.cpp:296
    std::vector<double> D_MALA(const double *grad, int m);                           // MALA.cpp:321 (implemented here)
    long double multinormal_logpdf(const double *deltavars, const double *drift1, int m);  // MALA.cpp:330 (implemented)
    std::vector<double> new_prop_values(const double *vars, int m, const double *drift);   // MALA.cpp:339
    int parallel_tempering(Model_def *model);                                        // MALA.cpp:397
    long double p1_fct(long double x);
    void p2_fct(Matrix &x);
    void p3_fct(std::vector<double> &x);
    // one iteration of the loop body MALA.cpp:645-703 for ALL chains (propose all -> batched evaluate -> accept all)
    int step(Model_def *model_current, Model_def *model_propose, Data *data_struc, Config *cfg);
    int compute_gradients(Model_def *model, Data *data_struc, Matrix &grad_out, const std::vector<char> &which,
                          bool fill_state = false);
    const Matrix &factor(int m);
    bool learn_at(long i) const;  // MALA.cpp:656-667: is the proposal law updated after iteration i?
    uint64_t get_seed() const { return seed; }
    void invalidate(int m) { Lchol_valid[(size_t)m] = 0; grad_valid[(size_t)m] = 0; }
    // the gradient held for chain m's position (tamcmc_sampler_get_gradient): rows of grad_cur / gradP_cur, and whether it is current
    const double *held_gradient(int m) const { return grad_cur.a.data() + (size_t)m * (size_t)Nvars; }
    const double *held_gradient_prior(int m) const { return gradP_cur.a.data() + (size_t)m * (size_t)Nvars; }
    bool gradient_valid(int m) const { return grad_valid[(size_t)m] != 0; }
    // audit of the last Langevin test (tamcmc_sampler_get_last_test): log q(x'|x), log q(x|x') up to their common constant
    std::vector<double> last_lq_fwd, last_lq_rev;
    const double *proposal_gradient(int m) const { return grad_prop.a.data() + (size_t)m * (size_t)Nvars; }
};

}  // namespace tamcmc
