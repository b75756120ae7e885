// host_model_def.cpp -- Model_def: chain-indexed container + dispatch (mirror of tamcmc/sources/model_def.cpp).
// The per-bin work of call_model / call_likelihood goes to the device through the C ABI; there is no CPU path.
#include <cmath>
#include <omp.h>
#include <limits>

#include "host_sampler.h"
#include "mode_tables.h"

namespace tamcmc {

// model_def.cpp:25-190 -- same initialisation order: params rows = inputs, vars = relaxed subset,
// then model + logL + prior for every chain once (here: one batched device call).
Model_def::Model_def(Config *config, const std::vector<double> &Tcoefs, bool /*verbose*/, tamcmc_hip_ctx *c) : ctx(c) {
    Nmodels = config->MALA.Nchains;
    model_fct_name_switch = config->modeling.model_fct_name_switch;
    likelihood_fct_name_switch = config->modeling.likelihood_fct_name_switch;
    prior_fct_name_switch = config->modeling.prior_fct_name_switch;
    priors_params_names_switch = config->modeling.inputs.priors_names_switch;
    relax = config->modeling.inputs.relax;
    plength = config->modeling.inputs.plength;
    extra_priors = config->modeling.inputs.extra_priors;
    likelihood_params = config->modeling.likelihood_params;
    priors_params = config->modeling.inputs.priors;
    Nparams = 0;
    for (int v : plength) Nparams += v;
    params = Matrix(Nmodels, Nparams);
    for (long m = 0; m < Nmodels; m++)
        for (long i = 0; i < Nparams; i++) params(m, i) = config->modeling.inputs.inputs[(size_t)i];
    Pmove.assign((size_t)Nmodels, 0.0);
    moved.assign((size_t)Nmodels, 0);
    comparator_MH.assign((size_t)Nmodels, 0.0);
    Nvars = 0;
    Ncons = 0;
    for (long i = 0; i < Nparams; i++) {
        if (relax[(size_t)i] == 1) { index_to_relax.push_back((int)i); Nvars++; }
        else { cons.push_back(params(0, i)); Ncons++; }
    }
    vars = Matrix(Nmodels, Nvars);
    for (long m = 0; m < Nmodels; m++)
        for (long k = 0; k < Nvars; k++) vars(m, k) = params(m, index_to_relax[(size_t)k]);
    logLikelihood.assign((size_t)Nmodels, 0.0);
    logPrior.assign((size_t)Nmodels, 0.0);
    logPosterior.assign((size_t)Nmodels, 0.0);
    // model_def.cpp:142-150: the initial model/logL is evaluated whatever the prior says
    std::vector<int32_t> status((size_t)Nmodels);
    std::vector<int32_t> pl(plength.begin(), plength.end());
    if (likelihood_fct_name_switch != 0) { last_status = TAMCMC_ERR_BAD_MODEL; return; }
    last_status = tamcmc_hip_loglike_params_batch(ctx, model_fct_name_switch, (int)Nmodels, params.a.data(), Nparams,
                                                  pl.data(), Tcoefs.data(), likelihood_params, logLikelihood.data(),
                                                  nullptr, status.data());
    for (long m = 0; m < Nmodels; m++) {
        logPrior[(size_t)m] = (double)call_prior(&config->data.data, (int)m);
        logPosterior[(size_t)m] = logLikelihood[(size_t)m] + logPrior[(size_t)m];
    }
    init_logLikelihood = logLikelihood;
}

void Model_def::update_params_with_vars(long m) {
    for (size_t i = 0; i < index_to_relax.size(); i++) params(m, index_to_relax[i]) = vars(m, (long)i);
}

long double Model_def::call_prior_params(const double *p) {
    int st = TAMCMC_OK;
    long double r;
    switch (prior_fct_name_switch) {  // Config/default/priors_ctrl.list
    case 2: r = priors_MS_Global(p, plength, priors_params, priors_params_names_switch, extra_priors, &st); break;
    case 3: r = priors_local(p, plength, priors_params, priors_params_names_switch, extra_priors, &st); break;
    case 4: r = priors_asymptotic(p, plength, priors_params, priors_params_names_switch, extra_priors, &st); break;
    default: st = TAMCMC_ERR_BAD_MODEL; r = -std::numeric_limits<long double>::infinity(); break;
    }
    if (st != TAMCMC_OK) {
#pragma omp atomic write
        last_status = st;  // (chains evaluate their priors on OpenMP threads)
    }
    return r;
}

long double Model_def::call_prior(Data *, int m) { return call_prior_params(params.row(m)); }

std::vector<double> Model_def::call_model(Data *data_struc, int m) {
    std::vector<double> model((size_t)data_struc->Nx);
    std::vector<int32_t> pl(plength.begin(), plength.end());
    double l = 0;
    int32_t st = 0;
    last_status = tamcmc_hip_loglike_params_batch(ctx, model_fct_name_switch, 1, params.row(m), Nparams, pl.data(), nullptr,
                                                  likelihood_params, &l, model.data(), &st);
    return model;
}

// model_def.cpp:466-482, per-chain call shape (synchronous, one-evaluation batch)
long double Model_def::generate_model(Data *data_struc, long m, const std::vector<double> &Tcoefs) {
    logPrior[(size_t)m] = (double)call_prior(data_struc, (int)m);
    if (logPrior[(size_t)m] != -INFINITY) {
        std::vector<int32_t> pl(plength.begin(), plength.end());
        int32_t st = 0;
        double T = Tcoefs[(size_t)m];
        last_status = tamcmc_hip_loglike_params_batch(ctx, model_fct_name_switch, 1, params.row(m), Nparams, pl.data(), &T,
                                                      likelihood_params, &logLikelihood[(size_t)m], nullptr, &st);
        logPosterior[(size_t)m] = logLikelihood[(size_t)m] + logPrior[(size_t)m];
    } else {
        logLikelihood[(size_t)m] = init_logLikelihood[(size_t)m];
        logPosterior[(size_t)m] = -INFINITY;
    }
    return logPosterior[(size_t)m];
}

int Model_def::generate_models_batch(Data *data_struc, const std::vector<double> &Tcoefs) {
    std::vector<int> live;
    int nt = omp_get_max_threads();  // the priors of the chains are independent (long double, the reference's term order per chain)
    if (nt > 8) nt = 8;
    if (nt > Nmodels / 4) nt = Nmodels / 4 > 0 ? (int)(Nmodels / 4) : 1;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long m = 0; m < Nmodels; m++) logPrior[(size_t)m] = (double)call_prior(data_struc, (int)m);
    for (long m = 0; m < Nmodels; m++) {
        if (logPrior[(size_t)m] != -INFINITY) live.push_back((int)m);
        else {
            logLikelihood[(size_t)m] = init_logLikelihood[(size_t)m];
            logPosterior[(size_t)m] = -INFINITY;
        }
    }
    if (live.empty()) return TAMCMC_OK;
    const size_t B = live.size();
    std::vector<double> P(B * (size_t)Nparams), T(B), L(B);
    for (size_t b = 0; b < B; b++) {
        const double *src = params.row(live[b]);
        std::copy(src, src + Nparams, P.begin() + (long)(b * (size_t)Nparams));
        T[b] = Tcoefs[(size_t)live[b]];
    }
    std::vector<int32_t> pl(plength.begin(), plength.end()), status(B);
    int rc = tamcmc_hip_loglike_params_batch(ctx, model_fct_name_switch, (int)B, P.data(), Nparams, pl.data(), T.data(),
                                             likelihood_params, L.data(), nullptr, status.data());
    for (size_t b = 0; b < B; b++) {
        const size_t m = (size_t)live[b];
        logLikelihood[m] = L[b];  // NaN when the table could not be built: rejected by the caller (MALA.cpp:490)
        logPosterior[m] = logLikelihood[m] + logPrior[m];
    }
    if (rc == TAMCMC_ERR_EMPTY_WINDOW || rc == TAMCMC_ERR_NAN_WINDOW) rc = TAMCMC_OK;  // surfaced as NaN logL
    last_status = rc;
    return rc;
}

}  // namespace tamcmc
