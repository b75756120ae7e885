// host_mala.cpp -- MALA: adaptive Metropolis-Hastings (Atchade 2006) + parallel tempering + (new) Langevin drift.
// Mirror of tamcmc/sources/MALA.cpp with the chain loop restructured as
//     propose ALL chains -> ONE batched device evaluation -> accept ALL -> adapt -> swap
// instead of the reference's `#pragma omp parallel for` over chains (MALA.cpp:648-668).
// use_drift=1 implements what the reference leaves as stubs (D_MALA MALA.cpp:321-328, multinormal_logpdf :330-337,
// fatal at :496-500): a preconditioned, optionally truncated Langevin mean shift driven by the forward-difference
// gradient of the log-posterior, with the asymmetric-proposal correction in the acceptance ratio.
#include <algorithm>
#include <cmath>
#include <omp.h>
#include <cstring>
#include <limits>

#include "host_sampler.h"
#include "rng.h"

namespace tamcmc {

// in-place lower Cholesky (row-major); returns false if not positive definite
static bool cholesky_lower(Matrix &A) {
    const long n = A.rows;
    for (long j = 0; j < n; j++) {
        double d = A(j, j);
        for (long k = 0; k < j; k++) d -= A(j, k) * A(j, k);
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        A(j, j) = d;
        for (long i = j + 1; i < n; i++) {
            double s = A(i, j);
            const double *ri = A.row(i), *rj = A.row(j);
            for (long k = 0; k < j; k++) s -= ri[k] * rj[k];
            A(i, j) = s / d;
        }
        for (long k = j + 1; k < n; k++) A(j, k) = 0.0;
    }
    return true;
}

MALA::MALA(Config *cfg) {
    Nsamples = cfg->outputs.Nsamples;
    Nchains = cfg->MALA.Nchains;
    seed = cfg->MALA.seed;
    const Input_Data &in = cfg->modeling.inputs;
    std::vector<double> vars;
    std::vector<std::string> vars_names;
    for (size_t i = 0; i < in.inputs.size(); i++)
        if (in.relax[i] == 1) {
            vars.push_back(in.inputs[i]);
            vars_names.push_back(i < in.inputs_names.size() ? in.inputs_names[i] : std::string());
        }
    Nvars = (long)vars.size();
    epsilon1 = cfg->MALA.epsilon1;
    epsi2 = cfg->MALA.epsi2;
    A1 = cfg->MALA.A1;
    delta = cfg->MALA.delta;
    delta_x = cfg->MALA.delta_x;
    c0 = cfg->MALA.c0;
    lambda_temp = cfg->MALA.lambda_temp;
    use_drift = cfg->MALA.use_drift != 0;
    fd_step_rel = cfg->MALA.fd_step_rel;
    swap_rule = cfg->MALA.swap_rule;
    Nt_learn = cfg->MALA.Nt_learn;
    periods_learn = cfg->MALA.periods_learn;
    dN_mixing = cfg->MALA.dN_mixing;
    target_acceptance = cfg->MALA.target_acceptance;
    gamma = c0;
    Tcoefs.resize((size_t)Nchains);
    for (long m = 0; m < Nchains; m++) Tcoefs[(size_t)m] = std::pow((double)lambda_temp, (double)m);  // MALA.cpp:103
    init_proposal(vars, vars_names, cfg->MALA.var_names_errors, cfg->MALA.fraction_errors, cfg->MALA.offset_errors);
    Lchol.assign((size_t)Nchains, Matrix());
    Lchol_valid.assign((size_t)Nchains, 0);
    grad_cur = Matrix(Nchains, Nvars);
    grad_prop = Matrix(Nchains, Nvars);
    gradP_cur = Matrix(Nchains, Nvars);
    gradP_prop = Matrix(Nchains, Nvars);
    grad_valid.assign((size_t)Nchains, 0);
    last_lq_fwd.assign((size_t)Nchains, 0.0);
    last_lq_rev.assign((size_t)Nchains, 0.0);
}

// MALA.cpp:246-293: diagonal covariance from errors_default.cfg (error = value*fraction + offset, default 1),
// sigma_m = 2.38^2 T_m^0.2 / Nvars, mu = initial vars
void MALA::init_proposal(const std::vector<double> &vars, const std::vector<std::string> &var_names,
                         const std::vector<std::string> &s_inerror, const std::vector<double> &fracerr,
                         const std::vector<double> &offseterr) {
    std::vector<double> error((size_t)Nvars, 1.0);
    for (size_t i = 0; i < vars.size(); i++)
        for (size_t j = 0; j < s_inerror.size(); j++)
            if (var_names[i] == s_inerror[j]) error[i] = vars[i] * fracerr[j] + offseterr[j];
    covarmat.assign((size_t)Nchains, Matrix(Nvars, Nvars));
    sigma.assign((size_t)Nchains, 0.0);
    mu = Matrix(Nchains, Nvars);
    for (long m = 0; m < Nchains; m++) {
        for (long k = 0; k < Nvars; k++) covarmat[(size_t)m](k, k) = error[(size_t)k] * error[(size_t)k];
        sigma[(size_t)m] = std::pow(2.38, 2) * std::pow(Tcoefs[(size_t)m], 0.2) / (double)Nvars;
        for (long k = 0; k < Nvars; k++) mu(m, k) = vars[(size_t)k];
    }
}

long double MALA::p1_fct(long double x) {  // MALA.cpp:135-151
    if (x < epsilon1) return epsilon1;
    if (x > A1) return A1;
    return x;
}
void MALA::p2_fct(Matrix &x) {  // MALA.cpp:153-164 (Frobenius norm)
    long double n2 = 0;
    for (double v : x.a) n2 += (long double)v * v;
    const double nrm = (double)std::sqrt(n2);
    if (!(nrm <= A1)) for (double &v : x.a) v = (double)(v * A1 / nrm);
}
void MALA::p3_fct(std::vector<double> &x) {  // MALA.cpp:166-176
    long double n2 = 0;
    for (double v : x) n2 += (long double)v * v;
    const double nrm = (double)std::sqrt(n2);
    if (!(nrm <= A1)) for (double &v : x) v = (double)(v * A1 / nrm);
}

// MALA.cpp:296-319 -- Robbins-Monro updates of mu, covarmat, sigma with gain gamma = c0/(1+i)
void MALA::update_proposal(const double *vars, long double acceptance, int m) {
    const double g = (double)gamma;
    std::vector<double> v3((size_t)Nvars);
    for (long k = 0; k < Nvars; k++) v3[(size_t)k] = mu(m, k) + g * (vars[k] - mu(m, k));
    p3_fct(v3);
    for (long k = 0; k < Nvars; k++) mu(m, k) = v3[(size_t)k];
    Matrix &C = covarmat[(size_t)m];
    std::vector<double> d((size_t)Nvars);
    for (long k = 0; k < Nvars; k++) d[(size_t)k] = vars[k] - mu(m, k);  // with the UPDATED mu, as the reference
    for (long i = 0; i < Nvars; i++) {
        double *ci = C.row(i);
        const double di = d[(size_t)i];
        for (long j = 0; j < Nvars; j++) ci[j] = ci[j] + g * (di * d[(size_t)j] - ci[j]);
    }
    p2_fct(C);
    const long double v1 = sigma[(size_t)m] + gamma * (acceptance - target_acceptance);
    sigma[(size_t)m] = (double)p1_fct(v1);
    Lchol_valid[(size_t)m] = 0;
}

// cached Cholesky factor of (covarmat[m] + epsilon2) * sigma[m]  (MALA.cpp:348-350 recomputes it every call)
const Matrix &MALA::factor(int m) {
    if (!Lchol_valid[(size_t)m]) {
        Matrix T = covarmat[(size_t)m];
        for (long i = 0; i < Nvars; i++) T(i, i) += epsi2;
        for (double &v : T.a) v *= sigma[(size_t)m];
        // not positive definite (only while gamma = c0/(1+i) > 1): the previous factor stays -- same rule as the device engine
        // (dev_sampler.hip::adapt_chain); the reference hands Eigen's partially computed factor on (MALA.cpp:348-350)
        if (cholesky_lower(T) || Lchol[(size_t)m].a.empty()) Lchol[(size_t)m] = T;
        Lchol_valid[(size_t)m] = 1;
    }
    return Lchol[(size_t)m];
}

// Langevin drift: (1/2) * (covarmat+eps2)*sigma * D,  D = grad, truncated to norm <= delta when delta > 0
// (truncated MALA, Atchade 2006).  Returns zeros when use_drift = 0, like the reference's stub.
std::vector<double> MALA::D_MALA(const double *grad, int m) {
    std::vector<double> drift((size_t)Nvars, 0.0);
    if (!use_drift || !grad) return drift;
    long double n2 = 0;
    for (long k = 0; k < Nvars; k++) n2 += (long double)grad[k] * grad[k];
    const double nrm = (double)std::sqrt(n2);
    if (!std::isfinite(nrm)) return drift;
    double scale = 1.0;
    if (delta > 0 && nrm > (double)delta) scale = (double)delta / nrm;
    const Matrix &C = covarmat[(size_t)m];
    const double s = 0.5 * sigma[(size_t)m] * scale;
    for (long i = 0; i < Nvars; i++) {
        double acc = 0;
        const double *ci = C.row(i);
        for (long j = 0; j < Nvars; j++) acc += ci[j] * grad[j];
        acc += epsi2 * grad[i];
        drift[(size_t)i] = s * acc;
    }
    return drift;
}

// log N(deltavars; drift1, (covarmat+eps2) sigma) up to the constant that cancels in the MH ratio
long double MALA::multinormal_logpdf(const double *deltavars, const double *drift1, int m) {
    const Matrix &L = factor(m);
    std::vector<double> w((size_t)Nvars);
    long double q = 0;
    for (long i = 0; i < Nvars; i++) {
        double s = deltavars[i] - drift1[i];
        const double *li = L.row(i);
        for (long k = 0; k < i; k++) s -= li[k] * w[(size_t)k];
        w[(size_t)i] = s / li[i];
        q += (long double)w[(size_t)i] * w[(size_t)i];
    }
    return -0.5L * q;
}

// MALA.cpp:339-369: x' = x + drift + chol((covarmat+eps2) sigma) z ; redraw on a non-finite result
std::vector<double> MALA::new_prop_values(const double *vars, int m, const double *drift) {
    const Matrix &L = factor(m);
    std::vector<double> z((size_t)Nvars + 1), ran((size_t)Nvars);
    for (int attempt = 0; attempt < 16; attempt++) {
        for (long k = 0; k < Nvars; k += 2)
            rng_normal2(seed, RNG_PROPOSAL, (uint32_t)m, (uint64_t)iteration, (uint32_t)(k / 2 + 4096 * attempt), z[(size_t)k],
                        z[(size_t)k + 1]);
        bool ok = true;
        for (long i = 0; i < Nvars; i++) {
            double s = 0;
            const double *li = L.row(i);
            for (long k = 0; k <= i; k++) s += li[k] * z[(size_t)k];
            ran[(size_t)i] = vars[i] + (drift ? drift[i] : 0.0) + s;
            ok = ok && std::isfinite(ran[(size_t)i]);
        }
        if (ok) break;
    }
    return ran;
}

// MALA.cpp:397-461 -- adjacent-pair swap on the TEMPERED log-likelihoods; returns ind_A
int MALA::parallel_tempering(Model_def *model) {
    double u, u2;
    rng_uniform2(seed, RNG_SWAP, 0, (uint64_t)iteration, 0, u, u2);
    int ind_A = (int)(u2 * (double)(Nchains - 1));
    if (ind_A > Nchains - 2) ind_A = (int)Nchains - 2;
    const int ind_B = ind_A + 1;
    const long double LA = model->logLikelihood[(size_t)ind_A], LB = model->logLikelihood[(size_t)ind_B];
    const long double logL_A_TB = LA * Tcoefs[(size_t)ind_A] / Tcoefs[(size_t)ind_B];
    const long double logL_B_TA = LB * Tcoefs[(size_t)ind_B] / Tcoefs[(size_t)ind_A];
    const double e = std::exp((double)(logL_A_TB + logL_B_TA - LA - LB));
    const long double r_T = std::min(1.0, e);
    model->swaped = false;
    model->Pswap = 0;
    Nswap_attempts++;
    if (u <= r_T) {
        const long Np = model->get_Nparams(), Nv = model->get_Nvars();
        std::swap_ranges(model->params.row(ind_A), model->params.row(ind_A) + Np, model->params.row(ind_B));
        std::swap_ranges(model->vars.row(ind_A), model->vars.row(ind_A) + Nv, model->vars.row(ind_B));
        const double prA = model->logPrior[(size_t)ind_A], prB = model->logPrior[(size_t)ind_B];
        model->logLikelihood[(size_t)ind_A] = (double)logL_B_TA;
        model->logPrior[(size_t)ind_A] = prB;
        model->logPosterior[(size_t)ind_A] = (double)(logL_B_TA + prB);
        model->logLikelihood[(size_t)ind_B] = (double)logL_A_TB;
        model->logPrior[(size_t)ind_B] = prA;
        // MALA.cpp:444 reads logPrior[ind_A] after :433 has overwritten it with B's: swap_rule 1 reproduces that (B keeps its own old prior
        // in the stored posterior); the default stores the posterior of the position B receives
        model->logPosterior[(size_t)ind_B] = (double)(logL_A_TB + (swap_rule == 1 ? prB : prA));
        std::swap(model->moved[(size_t)ind_A], model->moved[(size_t)ind_B]);
        std::swap(model->Pmove[(size_t)ind_A], model->Pmove[(size_t)ind_B]);
        if (use_drift) {  // the stored gradient follows the position: grad = (likelihood share) T_old/T_new + (prior share)
            const double TA = Tcoefs[(size_t)ind_A], TB = Tcoefs[(size_t)ind_B];
            for (long k = 0; k < Nvars; k++) {
                const double gLA = grad_cur(ind_A, k) - gradP_cur(ind_A, k), gLB = grad_cur(ind_B, k) - gradP_cur(ind_B, k);
                const double pA = gradP_cur(ind_A, k), pB = gradP_cur(ind_B, k);
                gradP_cur(ind_A, k) = pB;
                gradP_cur(ind_B, k) = pA;
                grad_cur(ind_A, k) = gLB * TB / TA + pB;   // B's position now sits at temperature T_A
                grad_cur(ind_B, k) = gLA * TA / TB + pA;
            }
            std::swap(grad_valid[(size_t)ind_A], grad_valid[(size_t)ind_B]);
        }
        model->swaped = true;
        model->Pswap = r_T;
        Nswap_accepted++;
    }
    model->comparator_PT = u;
    return ind_A;
}

// Forward-difference gradient of the tempered log-posterior for the chains flagged in `which`:
// likelihood part in ONE batched device call (C x (Nvars+1) evaluations), prior part on the host.
int MALA::compute_gradients(Model_def *model, Data *, Matrix &grad_out, const std::vector<char> &which, bool fill_state) {
    std::vector<int> live;
    for (long m = 0; m < Nchains; m++)
        if (which[(size_t)m]) live.push_back((int)m);
    if (live.empty()) return TAMCMC_OK;
    const long Np = model->get_Nparams();
    const std::vector<int> &idx = model->get_index_to_relax();
    std::vector<int32_t> idx32(idx.begin(), idx.end()), pl(model->get_plength().begin(), model->get_plength().end());
    const size_t C = live.size();
    std::vector<double> P(C * (size_t)Np), T(C), L0(C), Pr0(C), G(C * (size_t)Nvars), GP(C * (size_t)Nvars), h((size_t)Nvars);
    for (long k = 0; k < Nvars; k++) h[(size_t)k] = fd_step_rel * std::max(std::abs(mu(0, k)), 1e-3);
    for (size_t c = 0; c < C; c++) {
        std::memcpy(&P[c * (size_t)Np], model->params.row(live[c]), (size_t)Np * sizeof(double));
        T[c] = Tcoefs[(size_t)live[c]];
    }
    // likelihood AND prior parts on the device: one workgroup per (chain, perturbed variable) builds its table and its prior
    std::vector<int32_t> sw32(model->get_priors_switch().begin(), model->get_priors_switch().end());
    std::vector<double> extra(model->get_extra_priors());
    extra.resize(10, 0.0);
    int rc = tamcmc_hip_fd_gradient_posterior(model->get_ctx(), model->get_model_id(), model->get_prior_class(), (int)C, P.data(), Np,
                                              pl.data(), idx32.data(), (int)Nvars, h.data(), T.data(), model->get_likelihood_params(),
                                              model->get_priors().a.data(), sw32.data(), extra.data(), L0.data(), Pr0.data(), G.data(), GP.data());
    if (rc == TAMCMC_ERR_EMPTY_WINDOW || rc == TAMCMC_ERR_NAN_WINDOW) rc = TAMCMC_OK;
    for (size_t c = 0; c < C; c++) {
        const int m = live[c];
        for (long k = 0; k < Nvars; k++) {
            const double g = G[c * (size_t)Nvars + (size_t)k], gp = GP[c * (size_t)Nvars + (size_t)k];
            grad_out(m, k) = std::isfinite(g) ? g : 0.0;
            Matrix &gpr = (&grad_out == &grad_cur) ? gradP_cur : gradP_prop;
            gpr(m, k) = std::isfinite(g) ? gp : 0.0;  // prior share, kept apart: the likelihood share is re-tempered on a swap
        }
        if (fill_state) {  // the batch's base evaluation IS generate_model(m): prior -> model -> tempered logL (model_def.cpp:466-482)
            model->logPrior[(size_t)m] = Pr0[c];
            if (Pr0[c] != -INFINITY && !std::isnan(Pr0[c])) {
                model->logLikelihood[(size_t)m] = L0[c];
                model->logPosterior[(size_t)m] = L0[c] + Pr0[c];
            } else {
                model->logLikelihood[(size_t)m] = model->init_logLikelihood[(size_t)m];
                model->logPosterior[(size_t)m] = -INFINITY;
            }
        }
    }
    return rc;
}

bool MALA::learn_at(long i) const {
    bool logic = false;
    long which = 0;
    for (size_t l = 0; l < periods_learn.size() && l + 1 < Nt_learn.size(); l++)
        if ((i >= Nt_learn[l]) && (i < Nt_learn[l + 1])) { logic = true; which = (long)l; }
    return logic && (i % periods_learn[(size_t)which]) == 0;
}

// One iteration i of MALA::execute's loop body (MALA.cpp:645-703) for all chains.
int MALA::step(Model_def *cur, Model_def *prop, Data *data, Config *) {
    const long i = iteration;
    gamma = c0 / (1. + i);
    int rc = TAMCMC_OK;
    // The per-chain linear algebra (drift, proposal, the two proposal densities, adaptation + Cholesky) touches per-chain state only
    // and its random numbers are addressed by (chain, iteration): the chains run on OpenMP threads like the reference's
    // `#pragma omp parallel for` over m (MALA.cpp:648), with the same results for any thread count.
    int nt = omp_get_max_threads();
    if (nt > 16) nt = 16;
    if (nt > Nchains) nt = (int)Nchains;
    if (nt < 1) nt = 1;
    std::vector<std::vector<double>> drift_cur((size_t)Nchains);
    if (use_drift) {
        std::vector<char> need((size_t)Nchains);
        for (long m = 0; m < Nchains; m++) need[(size_t)m] = !grad_valid[(size_t)m];
        rc = compute_gradients(cur, data, grad_cur, need);
        if (rc) return rc;
#pragma omp parallel for schedule(static) num_threads(nt)
        for (long m = 0; m < Nchains; m++) {
            grad_valid[(size_t)m] = 1;
            drift_cur[(size_t)m] = D_MALA(grad_cur.row(m), (int)m);
        }
    }
    // [1] propose every chain (MALA.cpp:481-487)
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long m = 0; m < Nchains; m++) {
        std::vector<double> v = new_prop_values(cur->vars.row(m), (int)m, use_drift ? drift_cur[(size_t)m].data() : nullptr);
        std::memcpy(prop->params.row(m), cur->params.row(m), (size_t)cur->get_Nparams() * sizeof(double));
        for (long k = 0; k < Nvars; k++) prop->vars(m, k) = v[(size_t)k];
        prop->update_params_with_vars(m);
    }
    // [2] ONE batched evaluation (MALA.cpp:488 for all chains)
    if (!use_drift) {
        rc = prop->generate_models_batch(data, Tcoefs);
        if (rc) return rc;
    } else {
        // Langevin: the finite-difference batch of every proposal -- its base evaluation is the proposal's own
        // prior / model / logL, the other Nvars give the gradient the reverse-move density needs
        std::vector<char> all((size_t)Nchains, 1);
        rc = compute_gradients(prop, data, grad_prop, all, true);
        if (rc) return rc;
    }
    // [3] accept / reject (MALA.cpp:490-551)
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long m = 0; m < Nchains; m++) {
        double u, u_unused;
        rng_uniform2(seed, RNG_ACCEPT, (uint32_t)m, (uint64_t)i, 0, u, u_unused);
        long double r;
        if (!std::isnan(prop->logLikelihood[(size_t)m])) {
            if (prop->logPosterior[(size_t)m] == -INFINITY) r = 0.;
            else {
                long double lq_fwd = 0, lq_rev = 0;
                if (use_drift) {
                    std::vector<double> d((size_t)Nvars), dr((size_t)Nvars);
                    for (long k = 0; k < Nvars; k++) {
                        d[(size_t)k] = prop->vars(m, k) - cur->vars(m, k);
                        dr[(size_t)k] = -d[(size_t)k];
                    }
                    const std::vector<double> drift_prop = D_MALA(grad_prop.row(m), (int)m);
                    lq_fwd = multinormal_logpdf(d.data(), drift_cur[(size_t)m].data(), (int)m);   // log q(x'|x)
                    lq_rev = multinormal_logpdf(dr.data(), drift_prop.data(), (int)m);            // log q(x|x')
                    last_lq_fwd[(size_t)m] = (double)lq_fwd;
                    last_lq_rev[(size_t)m] = (double)lq_rev;
                }
                const double e = std::exp((double)((long double)prop->logPosterior[(size_t)m] - cur->logPosterior[(size_t)m] +
                                                   lq_rev - lq_fwd));
                r = std::min(1.0, e);
                if (std::isnan((double)r)) r = 0.;  // reference: fatal (MALA.cpp:519-521); here: reject
            }
        } else r = 0.;  // NaN model: always rejected (MALA.cpp:522-524)
        if (u <= r) {
            std::memcpy(cur->params.row(m), prop->params.row(m), (size_t)cur->get_Nparams() * sizeof(double));
            std::memcpy(cur->vars.row(m), prop->vars.row(m), (size_t)Nvars * sizeof(double));
            cur->logLikelihood[(size_t)m] = prop->logLikelihood[(size_t)m];
            cur->logPrior[(size_t)m] = prop->logPrior[(size_t)m];
            cur->logPosterior[(size_t)m] = prop->logPosterior[(size_t)m];
            cur->moved[(size_t)m] = 1;
            if (use_drift) {
                std::memcpy(grad_cur.row(m), grad_prop.row(m), (size_t)Nvars * sizeof(double));
                std::memcpy(gradP_cur.row(m), gradP_prop.row(m), (size_t)Nvars * sizeof(double));
            }
        } else cur->moved[(size_t)m] = 0;
        cur->Pmove[(size_t)m] = (double)r;
        cur->comparator_MH[(size_t)m] = u;
        // [2'] learning (MALA.cpp:656-667)
        if (learn_at(i))
            update_proposal(cur->vars.row(m), cur->Pmove[(size_t)m], (int)m);  // position unchanged: gradient stays valid
    }
    // [3] parallel tempering (MALA.cpp:688-703)
    if (dN_mixing > 0 && i % dN_mixing == 0 && i != 0 && Nchains > 1) parallel_tempering(cur);
    else { cur->swaped = false; cur->Pswap = 0; }
    iteration = i + 1;
    return TAMCMC_OK;
}

}  // namespace tamcmc
