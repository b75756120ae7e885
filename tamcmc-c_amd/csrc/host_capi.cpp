// host_capi.cpp -- C ABI of include/tamcmc_sampler.h over the host-side Model_def / MALA mirrors.
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>
#include <string>
#include <thread>

#include "../../include/tamcmc_sampler.h"
#include "dev_sampler.h"
#include "host_sampler.h"
#include "rng.h"

using namespace tamcmc;

extern "C" {
void tamcmc_hip_ctx_attach(tamcmc_hip_ctx *c);  // capi.hip (private)
void tamcmc_hip_ctx_detach(tamcmc_hip_ctx *c);
}

struct tamcmc_sampler {
    Config cfg;
    std::unique_ptr<MALA> mala;
    std::unique_ptr<Model_def> cur, prop;
    std::unique_ptr<DevSampler> dev;  // engine 1: the iteration runs on the GPU, the host objects mirror its state
    long accepted0 = 0;
    std::vector<int64_t> moves;     // per chain: iterations whose record carries moved = 1 (what the reference's acceptance diagnostic counts)
    tamcmc_hip_ctx *ctx = nullptr;  // borrowed
    bool attached = false;
    // engine 1: the host mirrors (cur, mala's proposal law) are refreshed from the device only when somebody looks at them
    mutable bool stale_state = false, stale_proposal = false;
    int sync_from_device(bool proposal_too);
    int refresh() const {
        if (!dev || (!stale_state && !stale_proposal)) return TAMCMC_OK;
        const int rc = const_cast<tamcmc_sampler *>(this)->sync_from_device(stale_proposal);
        if (rc == TAMCMC_OK) { stale_state = false; stale_proposal = false; }
        return rc;
    }
};

// pull the device engine's chain state (and, after learning, its proposal law) into the host mirrors
int tamcmc_sampler::sync_from_device(bool proposal_too) {
    const long Nc = cfg.MALA.Nchains;
    std::vector<int> moved((size_t)Nc);
    long counters[4];
    std::vector<long> mv((size_t)Nc);
    int rc = dev->download_state(cur->vars.a.data(), cur->params.a.data(), cur->logLikelihood.data(), cur->logPrior.data(),
                                 cur->logPosterior.data(), cur->Pmove.data(), moved.data(), counters, mv.data());
    if (rc) return rc;
    moves.assign(mv.begin(), mv.end());
    for (long m = 0; m < Nc; m++) cur->moved[(size_t)m] = (char)moved[(size_t)m];
    accepted0 = counters[1];
    mala->Nswap_attempts = counters[2];
    mala->Nswap_accepted = counters[3];
    if (proposal_too)
        for (long m = 0; m < Nc; m++) {
            rc = dev->download_proposal((int)m, mala->covarmat[(size_t)m].a.data(), mala->mu.row(m), &mala->sigma[(size_t)m]);
            if (rc) return rc;
            mala->invalidate((int)m);
        }
    return TAMCMC_OK;
}

extern "C" {

int tamcmc_sampler_create(tamcmc_sampler **out, tamcmc_hip_ctx *ctx, const tamcmc_sampler_config *c) {
    if (!out || !ctx || !c || !c->inputs || !c->relax || !c->plength || !c->priors || !c->priors_switch) return TAMCMC_ERR_BAD_ARG;
    if (c->Nchains < 1 || c->Nparams < 1) return TAMCMC_ERR_BAD_ARG;
    *out = nullptr;
    auto s = std::make_unique<tamcmc_sampler>();
    Config &g = s->cfg;
    g.modeling.model_fct_name_switch = c->model_id;
    g.modeling.prior_fct_name_switch = c->prior_class;
    g.modeling.likelihood_fct_name_switch = c->likelihood_id;
    g.modeling.likelihood_params = c->likelihood_params;
    Input_Data &in = g.modeling.inputs;
    const size_t Np = (size_t)c->Nparams;
    in.inputs.assign(c->inputs, c->inputs + Np);
    in.relax.assign(c->relax, c->relax + Np);
    in.plength.assign(c->plength, c->plength + 11);
    long psum = 0;
    for (int v : in.plength) psum += v;
    if (psum != c->Nparams) return TAMCMC_ERR_BAD_ARG;
    in.priors = Matrix(4, c->Nparams);
    std::memcpy(in.priors.a.data(), c->priors, 4 * Np * sizeof(double));
    in.priors_names_switch.assign(c->priors_switch, c->priors_switch + Np);
    in.extra_priors.assign(10, 0.0);
    for (int i = 0; i < c->n_extra && i < 10; i++) in.extra_priors[(size_t)i] = c->extra_priors[i];
    long nv = 0;
    for (size_t i = 0; i < Np; i++) {
        in.inputs_names.push_back("p" + std::to_string(i));
        if (in.relax[i] == 1) {
            g.MALA.var_names_errors.push_back(in.inputs_names.back());
            g.MALA.fraction_errors.push_back(0.0);
            g.MALA.offset_errors.push_back(c->init_errors ? c->init_errors[nv] : 1.0);
            nv++;
        }
    }
    if (nv < 1) return TAMCMC_ERR_BAD_ARG;
    g.MALA.Nchains = c->Nchains;
    g.MALA.lambda_temp = c->lambda_temp;
    g.MALA.target_acceptance = c->target_acceptance;
    g.MALA.c0 = c->c0;
    g.MALA.epsilon1 = c->epsilon1;
    g.MALA.epsi2 = c->epsilon2;
    g.MALA.A1 = c->A1;
    g.MALA.delta = c->delta;
    g.MALA.delta_x = c->delta_x;
    g.MALA.Nt_learn.assign(c->Nt_learn, c->Nt_learn + (c->Nt_learn ? c->n_Nt_learn : 0));
    g.MALA.periods_learn.assign(c->periods_learn, c->periods_learn + (c->periods_learn && c->n_Nt_learn > 0 ? c->n_Nt_learn - 1 : 0));
    g.MALA.dN_mixing = c->dN_mixing;
    g.MALA.use_drift = c->use_drift;
    g.MALA.seed = c->seed;
    g.MALA.fd_step_rel = c->fd_step_rel > 0 ? c->fd_step_rel : 1e-7;
    g.MALA.swap_rule = c->swap_rule == 1 ? 1 : 0;
    s->mala = std::make_unique<MALA>(&g);
    s->moves.assign((size_t)c->Nchains, 0);
    s->ctx = ctx;
    s->cur = std::make_unique<Model_def>(&g, s->mala->Tcoefs, false, ctx);
    if (s->cur->last_status != TAMCMC_OK) return s->cur->last_status;
    s->prop = std::make_unique<Model_def>(*s->cur);
    if (c->engine == 1) {
        s->dev = std::make_unique<DevSampler>();
        std::vector<int> idx(s->cur->get_index_to_relax());
        DevSamplerInit di;
        di.model_id = c->model_id; di.prior_class = c->prior_class; di.C = c->Nchains; di.Np = (int)c->Nparams; di.Nv = (int)nv;
        di.likelihood_params = c->likelihood_params;
        di.plength = in.plength.data(); di.index_to_relax = idx.data(); di.priors_switch = in.priors_names_switch.data();
        di.priors = in.priors.a.data(); di.extra_priors = in.extra_priors.data(); di.Tcoefs = s->mala->Tcoefs.data();
        di.seed = c->seed; di.dN_mixing = (long)c->dN_mixing; di.chain_groups = c->chain_groups;
        di.swap_rule = g.MALA.swap_rule;
        di.use_drift = c->use_drift ? 1 : 0; di.delta = c->delta; di.fd_step_rel = g.MALA.fd_step_rel;
        di.c0 = c->c0; di.epsilon1 = c->epsilon1; di.epsi2 = c->epsilon2; di.A1 = c->A1; di.target_acceptance = c->target_acceptance;
        int rc = s->dev->init(ctx, di);
        if (rc) return rc;
        rc = s->dev->upload_state(s->cur->vars.a.data(), s->cur->params.a.data(), s->cur->logLikelihood.data(),
                                  s->cur->logPrior.data(), s->cur->logPosterior.data(), s->cur->init_logLikelihood.data());
        if (rc) return rc;
        for (int m = 0; m < c->Nchains; m++) {
            rc = s->dev->upload_proposal(m, s->mala->factor(m).a.data(), s->mala->covarmat[(size_t)m].a.data(), s->mala->mu.row(m),
                                         s->mala->sigma[(size_t)m]);
            if (rc) return rc;
        }
    }
    tamcmc_hip_ctx_attach(ctx);
    s->attached = true;
    *out = s.release();
    return TAMCMC_OK;
}

void tamcmc_sampler_destroy(tamcmc_sampler *s) {
    if (!s) return;
    tamcmc_hip_ctx *ctx = s->attached ? s->ctx : nullptr;
    delete s;                       // (its device engine still uses the context's stream here)
    tamcmc_hip_ctx_detach(ctx);     // frees the context if tamcmc_hip_destroy was called while this sampler lived
}

int64_t tamcmc_sampler_nvars(const tamcmc_sampler *s) { return s ? s->cur->get_Nvars() : -1; }

int tamcmc_sampler_get_info(const tamcmc_sampler *s, int64_t *info, int32_t n) {
    if (!s || !info || n < 1) return TAMCMC_ERR_BAD_ARG;
    int64_t v[TAMCMC_SAMPLER_INFO_N] = {0};
    v[TAMCMC_INFO_ENGINE] = s->dev ? 1 : 0;
    v[TAMCMC_INFO_NVARS] = s->cur->get_Nvars();
    v[TAMCMC_INFO_NPARAMS] = s->cur->get_Nparams();
    v[TAMCMC_INFO_NCHAINS] = s->cfg.MALA.Nchains;
    v[TAMCMC_INFO_ADAPT_IN_LDS] = -1;
    if (s->dev) {
        long d[8];
        s->dev->info(d);
        v[TAMCMC_INFO_ADAPT_IN_LDS] = d[2];
        v[TAMCMC_INFO_FUSED_AVAILABLE] = d[3];
        v[TAMCMC_INFO_CHAIN_GROUPS] = d[4];
        v[TAMCMC_INFO_ITER_FUSED] = d[5];
        v[TAMCMC_INFO_ITER_LOCKSTEP] = d[6];
    }
    for (int32_t i = 0; i < n && i < TAMCMC_SAMPLER_INFO_N; i++) info[i] = v[i];
    return TAMCMC_OK;
}

int tamcmc_sampler_run(tamcmc_sampler *s, int64_t n_iter, double *samples, double *stats) {
    if (!s || n_iter < 0) return TAMCMC_ERR_BAD_ARG;
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    if (s->dev) {
        const long it0 = s->mala->iteration;
        std::vector<char> learn((size_t)n_iter);
        bool any = false;
        for (int64_t i = 0; i < n_iter; i++) { learn[(size_t)i] = s->mala->learn_at(it0 + i) ? 1 : 0; any = any || learn[(size_t)i]; }
        int rc = s->dev->run(it0, (long)n_iter, any ? learn.data() : nullptr, samples, stats);
        if (rc) return rc;
        s->mala->iteration = it0 + (long)n_iter;
        s->stale_state = true;                       // downloaded when a getter asks (tamcmc_sampler_get_state, ...)
        s->stale_proposal = s->stale_proposal || any;
        return TAMCMC_OK;
    }
    for (int64_t it = 0; it < n_iter; it++) {
        int rc = s->mala->step(s->cur.get(), s->prop.get(), &s->cfg.data.data, &s->cfg);
        if (rc) return rc;
        s->accepted0 += s->cur->moved[0] ? 1 : 0;
        for (long m = 0; m < Nc; m++) s->moves[(size_t)m] += s->cur->moved[(size_t)m] ? 1 : 0;
        if (samples) std::memcpy(samples + (size_t)it * Nc * Nv, s->cur->vars.a.data(), (size_t)(Nc * Nv) * sizeof(double));
        if (stats)
            for (long m = 0; m < Nc; m++) {
                double *r = stats + ((size_t)it * Nc + (size_t)m) * 3;
                r[0] = s->cur->logLikelihood[(size_t)m];
                r[1] = s->cur->logPrior[(size_t)m];
                r[2] = s->cur->logPosterior[(size_t)m];
            }
    }
    return TAMCMC_OK;
}

int tamcmc_sampler_run_packed(tamcmc_sampler *const *s, int32_t S, int64_t n_iter, double *const *samples, double *const *stats) {
    if (!s || S < 1 || n_iter < 0) return TAMCMC_ERR_BAD_ARG;
    for (int32_t k = 0; k < S; k++) {
        if (!s[k]) return TAMCMC_ERR_BAD_ARG;
        for (int32_t j = 0; j < k; j++)
            if (s[j] == s[k] || s[j]->ctx == s[k]->ctx) return TAMCMC_ERR_BAD_ARG;  // a context (its stream, its staging blocks) serves ONE sampler at a time
    }
    std::vector<int> rc((size_t)S, TAMCMC_OK);
    std::vector<std::thread> th;
    th.reserve((size_t)S);
    int32_t started = 1;
    try {  // (a thread that cannot be started must not unwind through the C boundary: its star runs on this thread instead)
        for (int32_t k = 1; k < S; k++) {
            th.emplace_back([&, k] { rc[(size_t)k] = tamcmc_sampler_run(s[k], n_iter, samples ? samples[k] : nullptr, stats ? stats[k] : nullptr); });
            started = k + 1;
        }
    } catch (...) {
    }
    rc[0] = tamcmc_sampler_run(s[0], n_iter, samples ? samples[0] : nullptr, stats ? stats[0] : nullptr);
    for (int32_t k = started; k < S; k++) rc[(size_t)k] = tamcmc_sampler_run(s[k], n_iter, samples ? samples[k] : nullptr, stats ? stats[k] : nullptr);
    for (auto &t : th) t.join();
    for (int32_t k = 0; k < S; k++)
        if (rc[(size_t)k]) return rc[(size_t)k];
    return TAMCMC_OK;
}

int tamcmc_sampler_draws(const tamcmc_sampler *s, int64_t iteration, double *z, double *u_accept, double *u_swap, int32_t *ind_A) {
    if (!s || iteration < 0) return TAMCMC_ERR_BAD_ARG;
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    const uint64_t seed = s->mala->get_seed();
    for (long m = 0; m < Nc; m++) {
        if (z)
            for (long k = 0; k < Nv; k += 2) {  // same addressing as MALA::new_prop_values / dev_sampler.hip::normals_into
                double z0, z1;
                rng_normal2(seed, RNG_PROPOSAL, (uint32_t)m, (uint64_t)iteration, (uint32_t)(k / 2), z0, z1);
                z[(size_t)(m * Nv + k)] = z0;
                if (k + 1 < Nv) z[(size_t)(m * Nv + k + 1)] = z1;
            }
        if (u_accept) {
            double u, unused;
            rng_uniform2(seed, RNG_ACCEPT, (uint32_t)m, (uint64_t)iteration, 0, u, unused);
            u_accept[m] = u;
        }
    }
    double u, u2;
    rng_uniform2(seed, RNG_SWAP, 0, (uint64_t)iteration, 0, u, u2);
    int a = (int)(u2 * (double)(Nc - 1));
    if (a > Nc - 2) a = (int)Nc - 2;
    if (u_swap) *u_swap = u;
    if (ind_A) *ind_A = Nc > 1 ? a : -1;
    return TAMCMC_OK;
}

int tamcmc_sampler_get_state(const tamcmc_sampler *s, double *vars, double *logL, double *logPrior, double *logPost,
                             double *Pmove, double *sigma, int64_t *counters) {
    if (!s) return TAMCMC_ERR_BAD_ARG;
    if (int rc = s->refresh()) return rc;
    const size_t Nc = (size_t)s->cfg.MALA.Nchains;
    if (vars) std::memcpy(vars, s->cur->vars.a.data(), s->cur->vars.a.size() * sizeof(double));
    if (logL) std::memcpy(logL, s->cur->logLikelihood.data(), Nc * sizeof(double));
    if (logPrior) std::memcpy(logPrior, s->cur->logPrior.data(), Nc * sizeof(double));
    if (logPost) std::memcpy(logPost, s->cur->logPosterior.data(), Nc * sizeof(double));
    if (Pmove) std::memcpy(Pmove, s->cur->Pmove.data(), Nc * sizeof(double));
    if (sigma) std::memcpy(sigma, s->mala->sigma.data(), Nc * sizeof(double));
    if (counters) {
        counters[0] = s->mala->iteration;
        counters[1] = s->accepted0;
        counters[2] = s->mala->Nswap_attempts;
        counters[3] = s->mala->Nswap_accepted;
    }
    return TAMCMC_OK;
}

int tamcmc_sampler_get_gradient(const tamcmc_sampler *s, double *grad, double *grad_prior, int32_t *valid) {
    if (!s || !s->cfg.MALA.use_drift) return TAMCMC_ERR_BAD_ARG;
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    if (s->dev) {
        const int rc = s->dev->download_gradient(grad, grad_prior);
        if (valid) for (long m = 0; m < Nc; m++) valid[m] = rc == TAMCMC_OK ? 1 : 0;
        return TAMCMC_OK;  // (no gradient yet: every chain flagged invalid)
    }
    for (long m = 0; m < Nc; m++) {
        if (valid) valid[m] = s->mala->gradient_valid((int)m) ? 1 : 0;
        if (grad) std::memcpy(grad + (size_t)m * Nv, s->mala->held_gradient((int)m), (size_t)Nv * sizeof(double));
        if (grad_prior) std::memcpy(grad_prior + (size_t)m * Nv, s->mala->held_gradient_prior((int)m), (size_t)Nv * sizeof(double));
    }
    return TAMCMC_OK;
}

int tamcmc_sampler_get_last_test(const tamcmc_sampler *s, double *vars_prop, double *stats_prop, double *lq, double *grad_prop) {
    if (!s) return TAMCMC_ERR_BAD_ARG;
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    if (s->dev) {  // device-resident Langevin engine: the proposals and their gradients; the scalars stay on the device (NaN here)
        for (long m = 0; m < Nc; m++) {
            if (stats_prop) stats_prop[3 * m] = stats_prop[3 * m + 1] = stats_prop[3 * m + 2] = NAN;
            if (lq) lq[2 * m] = lq[2 * m + 1] = NAN;
        }
        return s->dev->download_last_proposal(vars_prop, grad_prop);
    }
    for (long m = 0; m < Nc; m++) {
        if (vars_prop) std::memcpy(vars_prop + (size_t)m * Nv, s->prop->vars.row(m), (size_t)Nv * sizeof(double));
        if (stats_prop) {
            stats_prop[3 * m] = s->prop->logLikelihood[(size_t)m];
            stats_prop[3 * m + 1] = s->prop->logPrior[(size_t)m];
            stats_prop[3 * m + 2] = s->prop->logPosterior[(size_t)m];
        }
        if (lq) { lq[2 * m] = s->mala->last_lq_fwd[(size_t)m]; lq[2 * m + 1] = s->mala->last_lq_rev[(size_t)m]; }
        if (grad_prop && s->cfg.MALA.use_drift) std::memcpy(grad_prop + (size_t)m * Nv, s->mala->proposal_gradient((int)m), (size_t)Nv * sizeof(double));
    }
    return TAMCMC_OK;
}

int tamcmc_sampler_get_move_counts(const tamcmc_sampler *s, int64_t *moves) {
    if (!s || !moves) return TAMCMC_ERR_BAD_ARG;
    if (int rc = s->refresh()) return rc;
    for (size_t m = 0; m < (size_t)s->cfg.MALA.Nchains; m++) moves[m] = m < s->moves.size() ? s->moves[m] : 0;
    return TAMCMC_OK;
}

int tamcmc_sampler_get_proposal(const tamcmc_sampler *s, int32_t m, double *mu, double *covarmat) {
    if (!s || m < 0 || m >= s->cfg.MALA.Nchains) return TAMCMC_ERR_BAD_ARG;
    if (int rc = s->refresh()) return rc;
    const long Nv = s->cur->get_Nvars();
    if (mu) std::memcpy(mu, s->mala->mu.row(m), (size_t)Nv * sizeof(double));
    if (covarmat) std::memcpy(covarmat, s->mala->covarmat[(size_t)m].a.data(), (size_t)(Nv * Nv) * sizeof(double));
    return TAMCMC_OK;
}

int tamcmc_sampler_set_proposal(tamcmc_sampler *s, int32_t m, const double *mu, const double *covarmat, double sigma) {
    if (!s || m < 0 || m >= s->cfg.MALA.Nchains) return TAMCMC_ERR_BAD_ARG;
    if (int rc = s->refresh()) return rc;
    const long Nv = s->cur->get_Nvars();
    if (mu) std::memcpy(s->mala->mu.row(m), mu, (size_t)Nv * sizeof(double));
    if (covarmat) std::memcpy(s->mala->covarmat[(size_t)m].a.data(), covarmat, (size_t)(Nv * Nv) * sizeof(double));
    if (sigma > 0) s->mala->sigma[(size_t)m] = sigma;
    s->mala->invalidate(m);
    if (s->dev)
        return s->dev->upload_proposal(m, s->mala->factor(m).a.data(), s->mala->covarmat[(size_t)m].a.data(), s->mala->mu.row(m),
                                       s->mala->sigma[(size_t)m]);
    return TAMCMC_OK;
}

// Chain positions (and optionally the iteration counter) from outside: restart / resume (Config::read_restore_files +
// Model_def constructor path of the reference, config.cpp:1734-1990, MALA.cpp:100-131).  Re-evaluates prior and likelihood.
int tamcmc_sampler_set_state(tamcmc_sampler *s, const double *vars, int64_t iteration) {
    if (!s || !vars) return TAMCMC_ERR_BAD_ARG;
    if (int rc = s->refresh()) return rc;  // (the proposal law and the counters stay what the device holds)
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    for (long m = 0; m < Nc; m++) {
        std::memcpy(s->cur->vars.row(m), vars + (size_t)m * Nv, (size_t)Nv * sizeof(double));
        s->cur->update_params_with_vars(m);
        s->mala->invalidate((int)m);  // cached gradient of the old position
    }
    int rc = s->cur->generate_models_batch(&s->cfg.data.data, s->mala->Tcoefs);
    if (rc) return rc;
    if (iteration >= 0) s->mala->iteration = (long)iteration;
    if (s->dev) {
        rc = s->dev->upload_state(s->cur->vars.a.data(), s->cur->params.a.data(), s->cur->logLikelihood.data(), s->cur->logPrior.data(),
                                  s->cur->logPosterior.data(), s->cur->init_logLikelihood.data());
        if (rc) return rc;
    }
    return TAMCMC_OK;
}

int tamcmc_sampler_write_restore(const tamcmc_sampler *s, const char *root, const char *const *var_names) {
    if (!s || !root) return TAMCMC_ERR_BAD_ARG;
    if (int rc = s->refresh()) return rc;
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    std::vector<double> mus((size_t)(Nc * Nv)), covs((size_t)(Nc * Nv * Nv));
    for (long m = 0; m < Nc; m++) {
        std::memcpy(mus.data() + (size_t)(m * Nv), s->mala->mu.row(m), (size_t)Nv * sizeof(double));
        std::memcpy(covs.data() + (size_t)(m * Nv * Nv), s->mala->covarmat[(size_t)m].a.data(), (size_t)(Nv * Nv) * sizeof(double));
    }
    return tamcmc_outputs_write_restore(root, (int32_t)Nc, (int32_t)Nv, s->mala->iteration, var_names, s->cur->vars.a.data(),
                                        s->mala->sigma.data(), mus.data(), covs.data());
}

// do_restore_variables / do_restore_proposal / do_restore_last_index of the reference's !Outputs section (config.cpp:1780-1790)
int tamcmc_sampler_read_restore(tamcmc_sampler *s, const char *root, int32_t restore_variables, int32_t restore_proposal,
                                int32_t restore_last_index) {
    if (!s || !root) return TAMCMC_ERR_BAD_ARG;
    const long Nc = s->cfg.MALA.Nchains, Nv = s->cur->get_Nvars();
    int32_t nc = 0, nv = 0;
    int64_t it = 0;
    int rc = tamcmc_outputs_read_restore(root, &nc, &nv, &it, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (nc != Nc || nv != Nv) return TAMCMC_ERR_BAD_ARG;  // a restart needs the same chains and variables
    std::vector<double> vars((size_t)(Nc * Nv)), sig((size_t)Nc), mus((size_t)(Nc * Nv)), covs((size_t)(Nc * Nv * Nv));
    rc = tamcmc_outputs_read_restore(root, &nc, &nv, &it, vars.data(), sig.data(), mus.data(), covs.data());
    if (rc) return rc;
    if (restore_proposal)
        for (long m = 0; m < Nc; m++) {
            rc = tamcmc_sampler_set_proposal(s, (int32_t)m, mus.data() + (size_t)(m * Nv), covs.data() + (size_t)(m * Nv * Nv), sig[(size_t)m]);
            if (rc) return rc;
        }
    if (restore_variables) {
        rc = tamcmc_sampler_set_state(s, vars.data(), restore_last_index ? it : -1);
        if (rc) return rc;
    } else if (restore_last_index) s->mala->iteration = (long)it;
    return TAMCMC_OK;
}

double tamcmc_log_prior(int prior_class, const double *params, int64_t Nparams, const int32_t *plength, const double *priors,
                        const int32_t *priors_switch, const double *extra_priors, int32_t n_extra, int32_t *status) {
    if (status) *status = TAMCMC_OK;
    if (!params || !plength || !priors || !priors_switch || !extra_priors || Nparams < 1) {
        if (status) *status = TAMCMC_ERR_BAD_ARG;
        return NAN;
    }
    std::vector<int> pl(plength, plength + 11), sw(priors_switch, priors_switch + Nparams);
    std::vector<double> extra(10, 0.0);
    for (int i = 0; i < n_extra && i < 10; i++) extra[(size_t)i] = extra_priors[i];
    Matrix pp(4, Nparams);
    std::memcpy(pp.a.data(), priors, 4 * (size_t)Nparams * sizeof(double));
    int st = TAMCMC_OK;
    long double r;
    if (prior_class == 2) r = priors_MS_Global(params, pl, pp, sw, extra, &st);
    else if (prior_class == 3) r = priors_local(params, pl, pp, sw, extra, &st);
    else if (prior_class == 4) r = priors_asymptotic(params, pl, pp, sw, extra, &st);
    else { st = TAMCMC_ERR_BAD_MODEL; r = NAN; }
    if (status) *status = st;
    return (double)r;
}

}  // extern "C"
