// fit_star.cpp -- a complete fit of one star driven from C++ through the C ABI only (include/tamcmc_*.h), the way a host that
// keeps the reference's Config / .model / .data surface would use the library: no Python, no torch.
//
//   fit_star <local|global|asymptotic> <star.model> <star.data> <sampler.cfg> <errors.cfg> <output root> [slice index]
//
//   .data + .model  -> spectrum cut to the model's range, parameter vector, prior table      (tamcmc_io_*,   config.cpp:167-396)
//   sampler.cfg     -> !MALA / !Modeling / !Outputs settings                                 (tamcmc_cfg_*,  config.cpp:1223-1732)
//   errors.cfg      -> initial proposal scales                                               (config.cpp:2096-2150)
//   run             -> learning phases + Nsamples recorded iterations on the device-resident engine (host-driven engine for the
//                      red-giant models, whose mixed-mode solver runs as a device pre-step), MALA::execute MALA.cpp:623-745
//   outputs         -> <root>params.hdr/.bin per chain, stat_criteria, restore files, evidence line, summary table
//                      (outputs.cpp:863-1025, :1231-1333, :1472-1550; diagnostics.cpp:980-1066; bin2txt_params.cpp:165-168)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/tamcmc_hip.h"
#include "../include/tamcmc_io.h"
#include "../include/tamcmc_sampler.h"

#define CHECK(call, what)                                                                          \
    do {                                                                                           \
        const int rc_ = (call);                                                                    \
        if (rc_ != 0) { std::fprintf(stderr, "fit_star: %s failed (%d): %s\n", what, rc_, err()); return 1; } \
    } while (0)

static tamcmc_hip_ctx *g_ctx = nullptr;
static const char *err() {
    const char *e = tamcmc_io_last_error();
    if (e && *e) return e;
    e = tamcmc_cfg_last_error();
    if (e && *e) return e;
    return g_ctx ? tamcmc_hip_last_error(g_ctx) : "";
}

int main(int argc, char **argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s <local|global|asymptotic> <star.model> <star.data> <sampler.cfg> <errors.cfg> <output root> [slice]\n", argv[0]);
        return 2;
    }
    const std::string dialect = argv[1], root = argv[6];
    const int slice = argc > 7 ? std::atoi(argv[7]) : 0;
    // ---- inputs
    double *tab = nullptr;
    int64_t nr = 0, nc = 0, i0 = 0, i1 = 0;
    CHECK(tamcmc_io_read_data(argv[3], &tab, &nr, &nc), "read_data");
    if (nr < 3 || nc < 2) { std::fprintf(stderr, "fit_star: the data file needs two columns and three rows\n"); return 1; }
    const double resol = tab[2 * nc] - tab[1 * nc];  // config.cpp:682, :720
    tamcmc_inputs *in = nullptr;
    if (dialect == "local") CHECK(tamcmc_io_load_model_local(argv[2], slice, resol, &in), "load_model_local");
    else if (dialect == "global") CHECK(tamcmc_io_load_model_global(argv[2], resol, &in), "load_model_global");
    else if (dialect == "asymptotic") CHECK(tamcmc_io_load_model_asymptotic(argv[2], resol, &in), "load_model_asymptotic");
    else { std::fprintf(stderr, "fit_star: unknown dialect %s\n", dialect.c_str()); return 2; }
    const int Np = tamcmc_inputs_nparams(in);
    std::vector<double> params((size_t)Np), priors((size_t)4 * Np);
    std::vector<int32_t> relax((size_t)Np), psw((size_t)Np);
    int32_t plength[11], model_id = 0, prior_class = 0;
    double extra[10], range[2];
    CHECK(tamcmc_inputs_get(in, params.data(), relax.data(), priors.data(), psw.data(), plength, extra, range, &model_id, &prior_class, nullptr, nullptr),
          "inputs_get");
    CHECK(tamcmc_io_select_range(tab, nr, nc, 0, range[0], range[1], &i0, &i1), "select_range");
    const int64_t Nx = i1 - i0;
    std::vector<double> x((size_t)Nx), y((size_t)Nx);
    for (int64_t i = 0; i < Nx; i++) { x[(size_t)i] = tab[(i0 + i) * nc]; y[(size_t)i] = tab[(i0 + i) * nc + 1]; }
    tamcmc_io_free(tab);
    std::vector<const char *> names((size_t)Np), vnames;
    std::vector<double> vvals;
    for (int i = 0; i < Np; i++) {
        names[(size_t)i] = tamcmc_inputs_name(in, i);
        if (relax[(size_t)i]) { vnames.push_back(names[(size_t)i]); vvals.push_back(params[(size_t)i]); }
    }
    const int64_t Nv = (int64_t)vnames.size();
    std::printf("%s: model %s (id %d), %d parameters (%lld free), %lld bins in [%g, %g]\n", argv[2], tamcmc_inputs_model_name(in), model_id, Np,
                (long long)Nv, (long long)Nx, range[0], range[1]);
    // ---- settings
    tamcmc_cfg *cfg = nullptr;
    CHECK(tamcmc_cfg_open(argv[4], &cfg), "cfg_open");
    tamcmc_sampler_config sc;
    std::memset(&sc, 0, sizeof sc);
    int64_t Nt_learn[16], periods[16], Nsamples = 0, Nbuffer = 0;
    CHECK(tamcmc_cfg_sampler(cfg, &sc, Nt_learn, periods, 16, &Nsamples, &Nbuffer), "cfg_sampler");
    tamcmc_cfg_free(cfg);
    std::vector<double> errs((size_t)Nv);
    CHECK(tamcmc_io_init_errors(argv[5], vnames.data(), vvals.data(), Nv, errs.data()), "init_errors");
    sc.model_id = model_id; sc.prior_class = prior_class;  // the .model file decides the model family, as the reference's io_* front ends do
    sc.Nparams = Np; sc.inputs = params.data(); sc.relax = relax.data(); sc.plength = plength;
    sc.priors = priors.data(); sc.priors_switch = psw.data(); sc.extra_priors = extra; sc.n_extra = 10;
    sc.init_errors = errs.data();
    sc.seed = 20240229;
    const bool red_giant = (model_id == TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4 || model_id == TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4);
    sc.engine = (sc.use_drift || red_giant) ? 0 : 1;
    // ---- device
    CHECK(tamcmc_hip_create(&g_ctx, 0), "hip_create (the product path has no CPU fallback)");
    CHECK(tamcmc_hip_set_option(g_ctx, TAMCMC_OPT_PRECISION, TAMCMC_PRECISION_FAST), "set_option");
    CHECK(tamcmc_hip_set_spectrum(g_ctx, x.data(), y.data(), Nx), "set_spectrum");
    tamcmc_sampler *s = nullptr;
    CHECK(tamcmc_sampler_create(&s, g_ctx, &sc), "sampler_create");
    // ---- learning, then the recorded samples (buffers of Nbuffer iterations appended to the output files)
    const int64_t n_learn = sc.n_Nt_learn > 0 ? Nt_learn[sc.n_Nt_learn - 1] : 0;
    CHECK(tamcmc_sampler_run(s, n_learn, nullptr, nullptr), "sampler_run (learning)");
    const int32_t C = sc.Nchains;
    const int64_t chunk = Nbuffer > 0 && Nbuffer < Nsamples ? Nbuffer : Nsamples;
    std::vector<double> smp((size_t)(chunk * C * Nv)), st((size_t)(chunk * C * 3)), all_stats, cold;
    for (int64_t done = 0; done < Nsamples; done += chunk) {
        const int64_t n = Nsamples - done < chunk ? Nsamples - done : chunk;
        CHECK(tamcmc_sampler_run(s, n, smp.data(), st.data()), "sampler_run");
        CHECK(tamcmc_outputs_write_params(root.c_str(), smp.data(), n, C, (int32_t)Nv, Nsamples, relax.data(), plength, 11, Np, params.data(), names.data(),
                                          done > 0), "write_params");
        CHECK(tamcmc_outputs_write_stat_criteria(root.c_str(), st.data(), n, C, done > 0), "write_stat_criteria");
        all_stats.insert(all_stats.end(), st.begin(), st.begin() + (size_t)(n * C * 3));
        for (int64_t i = 0; i < n; i++) cold.insert(cold.end(), smp.begin() + (size_t)(i * C * Nv), smp.begin() + (size_t)(i * C * Nv + Nv));
    }
    CHECK(tamcmc_sampler_write_restore(s, (root + "restore_").c_str(), vnames.data()), "write_restore");
    // ---- diagnostics: evidence of the ladder, summary of the coldest chain
    std::vector<double> T((size_t)C), beta((size_t)C), Lb((size_t)C);
    for (int32_t m = 0; m < C; m++) T[(size_t)m] = m == 0 ? 1.0 : T[(size_t)m - 1] * sc.lambda_temp;
    double evidence = 0;
    if (C > 1) {
        CHECK(tamcmc_evidence_calc(T.data(), C, all_stats.data(), Nsamples, 3 * C, 3, 1000, beta.data(), Lb.data(), nullptr, nullptr, &evidence), "evidence_calc");
        CHECK(tamcmc_outputs_write_evidence((root + "evidence.txt").c_str(), Nsamples, C, beta.data(), Lb.data(), 1000, evidence, 1), "write_evidence");
    }
    std::vector<double> mean((size_t)Nv), med((size_t)Nv), sd((size_t)Nv);
    CHECK(tamcmc_params_summary(cold.data(), Nsamples, (int32_t)Nv, Nv, mean.data(), med.data(), sd.data()), "params_summary");
    int64_t cnt[4];
    CHECK(tamcmc_sampler_get_state(s, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cnt), "get_state");
    std::printf("iterations %lld, accepted moves of the coldest chain %lld, swaps %lld / %lld, evidence figure %.6f\n", (long long)cnt[0],
                (long long)cnt[1], (long long)cnt[3], (long long)cnt[2], evidence);
    std::printf("%-40s %16s %16s %14s\n", "variable", "mean", "median", "stddev");
    for (int64_t v = 0; v < Nv; v++) std::printf("%-40s %16.8g %16.8g %14.6g\n", vnames[(size_t)v], mean[(size_t)v], med[(size_t)v], sd[(size_t)v]);
    tamcmc_sampler_destroy(s);
    tamcmc_hip_destroy(g_ctx);
    tamcmc_inputs_free(in);
    return 0;
}
