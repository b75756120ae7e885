// host_io.cpp -- input front end (include/tamcmc_io.h): `.data` reader, `.model` reader and the parameter-vector /
// prior-table builders of the local fit (model_MS_local_basic), of the global main-sequence fits
// (model_MS_Global_aj_HarveyLike, model_MS_Global_a1etaa3_HarveyLike_Classic) and of the red-giant fits
// (model_RGB_asympt_aj_AppWidth_HarveyLike_v4, ..._CteWidth_..., io_asymptotic.cpp).  Plain C++ (no device code).
//
// Restates, in its own structure (a table of parameter blocks instead of the reference's per-degree vectors):
//   Config::read_data_ascii_Ncols  tamcmc/sources/config.cpp:907-1060     Config::setup range cut  config.cpp:312-347
//   read_MCMC_file_local           tamcmc/sources/io_local.cpp:25-327      build_init_local         io_local.cpp:329-1176
//   set_noise_params_local         io_local.cpp:1178-1238                  IO_models::fill_param*   io_models.cpp:40-120
//   read_MCMC_file_MS_Global / build_init_MS_Global / set_noise_params / settings_aj_splittings
//                                  tamcmc/sources/io_ms_global.cpp:27-360, :362-1445, :1447-1536, :1718-1850
// The reference exits on malformed input; every such exit is a TAMCMC_IO_ERR_* code here.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/tamcmc_hip.h"
#include "../../include/tamcmc_io.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

std::string trim(const std::string &s) {
    const char *ws = " \t\r\n";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    const size_t b = s.find_last_not_of(ws);
    return s.substr(a, b - a + 1);
}
std::vector<std::string> split(const std::string &s, const char *delims) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        const size_t a = s.find_first_not_of(delims, i);
        if (a == std::string::npos) break;
        size_t b = s.find_first_of(delims, a);
        if (b == std::string::npos) b = s.size();
        out.push_back(s.substr(a, b - a));
        i = b;
    }
    return out;
}
// the reference converts through `istringstream >> long double` (config.cpp:1017) / stod-like helpers
bool to_double(const std::string &w, double *v) {
    std::istringstream is(w);
    long double t = 0;
    if (!(is >> t)) return false;
    *v = (double)t;
    return true;
}
bool to_flag(const std::string &w) {  // str_to_bool, string_handler.cpp:481-487
    std::istringstream is(trim(w));
    bool b = false;
    is >> b;
    return b;
}

// ---------------------------------------------------------------- parameter blocks (Input_Data of data.h:51-62)
struct Block {
    std::vector<std::string> names, prior_names;
    std::vector<double> inputs;
    std::vector<int> relax;
    std::vector<double> priors;  // 4 x n, row-major
    int n = 0;
    void init(int size) {  // IO_models::initialise_param, io_models.cpp:255-290
        n = size;
        names.assign((size_t)size, "Empty");
        prior_names.assign((size_t)size, "Fix");
        inputs.assign((size_t)size, 0.0);
        relax.assign((size_t)size, 0);
        priors.assign((size_t)4 * size, -9999.0);
    }
    double &pr(int k, int i) { return priors[(size_t)k * n + i]; }
    // IO_models::fill_param, io_models.cpp:40-75: vals[k + i0] -> priors(k, pos) unless the prior is "Fix"
    void fill(const std::string &name, const std::string &prior, double value, const std::vector<double> &vals, int pos, int i0) {
        names[(size_t)pos] = name;
        prior_names[(size_t)pos] = prior;
        inputs[(size_t)pos] = value;
        const bool fixed = (prior == "Fix");
        relax[(size_t)pos] = fixed ? 0 : 1;
        for (int k = 0; k < 4; k++) {
            const size_t q = (size_t)(k + i0);
            pr(k, pos) = fixed ? -9999.0 : (q < vals.size() ? vals[q] : -9999.0);
        }
    }
    // IO_models::fill_param_vect, io_models.cpp:77-96
    void fill_vect(const std::vector<double> &v, const std::vector<int> &rel, const std::string &name, const std::string &prior,
                   const std::vector<double> &vals, int pos, int i0_free) {
        for (size_t i = 0; i < v.size(); i++) fill(name, rel[i] ? prior : "Fix", v[i], vals, pos + (int)i, i0_free);
    }
    std::vector<double> prior_col(int i) {
        return {pr(0, i), pr(1, i), pr(2, i), pr(3, i)};
    }
};

// Config/default/primepriors_ctrl.list
int prior_id(const std::string &nm) {
    static const struct { const char *n; int id; } tab[] = {
        {"None", 0}, {"Fix", 0}, {"Uniform", 1}, {"Gaussian", 2}, {"multivar_Gaussian", 3}, {"Jeffreys", 4}, {"UG", 5}, {"GU", 6},
        {"GUG", 7}, {"Uniform_abs", 8}, {"Uniform_cos", 9}, {"Jeffreys_abs", 10}, {"Tabulated", 11}, {"Tabulated_2d", 12}, {"Auto", 13}};
    for (const auto &e : tab)
        if (nm == e.n) return e.id;
    return -1;
}

// harvey_like on a few points (noise_models.cpp:15-39): sum_k H_k/(1+(1e-3 tau_k x)^p_k) for tau_k != 0, + white noise
double harvey_at(const std::vector<double> &np, double x) {
    const int nh = ((int)np.size() - 1) / 3;
    double m = 0;
    for (int k = 0; k < nh; k++)
        if (np[(size_t)3 * k + 1] != 0) m += np[(size_t)3 * k] / (1. + std::pow(1e-3 * np[(size_t)3 * k + 1] * x, np[(size_t)3 * k + 2]));
    return m + np.back();
}

struct ModeLine { int l; double f; int rf, rH, rW; };
struct Common { std::string name, prior; std::vector<double> v; };
struct ModelFile {
    std::string id;
    double dnu = 0, c_l = 0, numax = -9999, err_numax = -9999;
    double range[2] = {0, 0};
    std::vector<std::vector<double>> hyper;     // "hyper priors" rows: value, then the numbers after the prior keyword
    std::vector<std::string> hyper_names;       // prior keyword of each row (absent in the one-column form)
    bool have_range = false;
    std::vector<ModeLine> modes;
    std::vector<std::vector<double>> eigen;     // l, nu, nu_min, nu_max, Gamma, H
    std::vector<double> noise;                  // 10 values, left-padded with -1
    std::vector<std::vector<double>> noise_s2;  // 10 rows (value, err-, err+), missing leading rows = -1
    std::vector<Common> common;
};

}  // namespace

struct tamcmc_inputs {
    Block all;
    int plength[11];
    double extra[10];
    double range[2];
    double dnu, c_l;
    int model_id, prior_class;
    std::string model_name;
};

namespace {

// read_MCMC_file_local (io_local.cpp:25-327) and read_MCMC_file_MS_Global (io_ms_global.cpp:27-360; same layout, ONE range):
// sections are delimited by COUNTING the lines that start with '#'
int read_model_file(const char *path, int slice_ind /* < 0: global fit, one range */, ModelFile &mf) {
    std::ifstream f(path);
    if (!f.is_open()) return fail(TAMCMC_IO_ERR_OPEN, std::string("cannot open ") + path);
    std::vector<std::string> L;
    for (std::string ln; std::getline(f, ln);) L.push_back(trim(ln));
    size_t ip = 0;
    auto have = [&]() { return ip < L.size(); };
    int hashes = 0, ranges = 0;
    // --- header + mode list: up to and including the third '#' line (:55-128)
    while (hashes < 3 && have()) {
        const std::string &ln = L[ip++];
        if (ln.empty()) continue;
        const char c0 = ln[0], c1 = ln.size() > 1 ? ln[1] : ' ';
        if (c0 == '#') {
            if (c1 == 'K') {
                const auto w = split(ln, "= \t");
                if (w.size() > 1) mf.id = w[1];
            }
            hashes++;
        } else if (c0 == '!') {
            const auto w = split(ln, " \t");
            double v = 0;
            if (w.size() < 2 || !to_double(w[1], &v)) return fail(TAMCMC_IO_ERR_SYNTAX, "'!' line without a value: " + ln);
            if (c1 == '!') mf.c_l = v;
            else if (c1 == 'n') {
                mf.numax = v;
                if (w.size() == 3 && !to_double(w[2], &mf.err_numax)) return fail(TAMCMC_IO_ERR_SYNTAX, "'!n' line: bad uncertainty: " + ln);
            } else mf.dnu = v;
        } else if (c0 == '*') {
            const auto w = split(ln, " \t");
            if (slice_ind < 0 && ranges > 0) return fail(TAMCMC_IO_ERR_SYNTAX, "a global fit takes ONE '*' frequency range (io_ms_global.cpp:93-104)");
            if (ranges == (slice_ind < 0 ? 0 : slice_ind)) {
                if (w.size() < 3 || !to_double(w[1], &mf.range[0]) || !to_double(w[2], &mf.range[1]))
                    return fail(TAMCMC_IO_ERR_SYNTAX, "bad '*' range line: " + ln);
                mf.have_range = true;
            }
            ranges++;
        } else {
            const auto w = split(ln, " \t");
            if (w.empty()) continue;
            if (w[0] != "p" && w[0] != "g" && w[0] != "co") return fail(TAMCMC_IO_ERR_SYNTAX, "mode type must be p, g or co: " + ln);
            ModeLine m;
            double lv = 0;
            if (w.size() < 3 || !to_double(w[1], &lv) || !to_double(w[2], &m.f)) return fail(TAMCMC_IO_ERR_SYNTAX, "bad mode line: " + ln);
            m.l = (int)lv;
            m.rf = w.size() >= 4 ? to_flag(w[3]) : 1;  // missing flags default to "free" (:97-113)
            m.rH = w.size() >= 5 ? to_flag(w[4]) : 1;
            m.rW = w.size() >= 6 ? to_flag(w[5]) : 1;
            mf.modes.push_back(m);
        }
    }
    if (!mf.have_range) return fail(TAMCMC_IO_ERR_SYNTAX, "no '*' frequency range for this slice index");
    if (have()) ip++;  // the reference reads one more line here and drops it (the "# Extra parameters" header, :124 then :143)
    // --- "hyper priors" (io_ms_global.cpp:160-207): lines up to the next '#' line: value [prior keyword, its numbers]; only the
    //     red-giant dialect (io_asymptotic.cpp) uses them (nodes of the frequency-bias spline)
    while (hashes < 4 && have()) {
        const std::string &ln = L[ip++];
        if (ln.empty()) continue;
        if (ln[0] == '#') { hashes++; continue; }
        const auto w = split(ln, " \t");
        std::vector<double> r;
        double v = 0;
        if (w.empty() || !to_double(w[0], &v)) continue;  // free text between the section headers of a file without hyper priors
        if (w.size() == 2) return fail(TAMCMC_IO_ERR_SYNTAX, "hyper prior: one column (value) or at least three (value, prior, numbers): " + ln);
        r.push_back(v);
        for (size_t k = 2; k < w.size(); k++) {
            if (!to_double(w[k], &v)) return fail(TAMCMC_IO_ERR_SYNTAX, "hyper prior: not a number: " + ln);
            r.push_back(v);
        }
        mf.hyper.push_back(r);
        if (w.size() > 1) mf.hyper_names.push_back(w[1]);
    }
    // --- eigen table (:182-203): rows of six numbers up to the next '#' line
    while (hashes < 5 && have()) {
        const std::string &ln = L[ip++];
        if (ln.empty()) continue;
        if (ln[0] == '#') { hashes++; continue; }
        const auto w = split(ln, " \t");
        if (w.size() != 6) return fail(TAMCMC_IO_ERR_SYNTAX, "eigen table row must have 6 columns: " + ln);
        std::vector<double> r(6);
        for (int k = 0; k < 6; k++)
            if (!to_double(w[(size_t)k], &r[(size_t)k])) return fail(TAMCMC_IO_ERR_SYNTAX, "eigen table: not a number: " + ln);
        mf.eigen.push_back(r);
    }
    // --- noise parameters (:205-230): up to 10 values, right-aligned, missing ones = -1
    std::vector<double> nz;
    while (hashes < 6 && have()) {
        const std::string &ln = L[ip++];
        if (ln.empty()) continue;
        if (ln[0] == '#') { hashes++; continue; }
        for (const auto &w : split(ln, " \t")) {
            double v = 0;
            if (!to_double(w, &v)) return fail(TAMCMC_IO_ERR_SYNTAX, "noise parameters: not a number: " + ln);
            nz.push_back(v);
        }
    }
    if (nz.size() > 10) return fail(TAMCMC_IO_ERR_SYNTAX, "more than 10 noise parameters");
    mf.noise.assign(10, -1.0);
    for (size_t k = 0; k < nz.size(); k++) mf.noise[10 - nz.size() + k] = nz[k];
    // --- noise information of the previous analysis step (:232-256): rows (value, err-, err+), right-aligned to 10 rows
    std::vector<std::vector<double>> s2;
    while (hashes < 7 && have()) {
        const std::string &ln = L[ip++];
        if (ln.empty()) continue;
        if (ln[0] == '#') { hashes++; continue; }
        const auto w = split(ln, " \t");
        std::vector<double> r(3, 0.0);
        for (size_t k = 0; k < 3 && k < w.size(); k++)
            if (!to_double(w[k], &r[k])) return fail(TAMCMC_IO_ERR_SYNTAX, "noise information: not a number: " + ln);
        s2.push_back(r);
    }
    if (s2.size() > 10) return fail(TAMCMC_IO_ERR_SYNTAX, "more than 10 rows of noise information");
    mf.noise_s2.assign(10, std::vector<double>(3, -1.0));
    for (size_t k = 0; k < s2.size(); k++) mf.noise_s2[10 - s2.size() + k] = s2[k];
    // --- controls and priors of the common parameters (:258-287): name, prior keyword, up to 5 numbers
    while (hashes < 9 && have()) {
        const std::string &ln = L[ip++];
        if (ln.empty()) continue;
        if (ln[0] == '#') { hashes++; continue; }
        const auto w = split(ln, " \t");
        if (w.size() < 2) return fail(TAMCMC_IO_ERR_SYNTAX, "common-parameter line needs a name and a prior keyword: " + ln);
        Common c;
        c.name = w[0];
        c.prior = w[1];
        c.v.assign(5, -9999.0);
        for (size_t k = 2; k < w.size() && k < 7; k++)
            if (!to_double(w[k], &c.v[k - 2])) c.v[k - 2] = 0.0;  // a non-numeric field (model_fullname's value is in c.prior)
        mf.common.push_back(c);
    }
    return TAMCMC_IO_OK;
}

int build_local(const ModelFile &mf, double resol, tamcmc_inputs &out) {
    const long double pi = 3.141592653589793238L;
    const double G = 6.667e-8, Dnu_sun = 135.1, R_sun = 6.96342e5, M_sun = 1.98855e30;
    const double rho_sun = (double)(M_sun * 1e3 / (4 * pi * std::pow(R_sun * 1e5, 3) / 3));
    const double rho = std::pow(mf.dnu / Dnu_sun, 2.) * rho_sun;
    const double Hmin = 1, Hmax = 10000;
    // ---- switches read before anything else (io_local.cpp:381-411)
    std::string model;
    int do_amp = 0;
    for (const auto &c : mf.common) {
        if (c.name == "model_fullname") model = c.prior;
        if (c.name == "fit_squareAmplitude_instead_Height") {
            if (c.prior != "bool") return fail(TAMCMC_IO_ERR_SYNTAX, "fit_squareAmplitude_instead_Height must be 'bool'");
            do_amp = c.v[0] != 0;
        }
    }
    if (model.empty()) return fail(TAMCMC_IO_ERR_SYNTAX, "the .model file has no model_fullname");
    if (model != "model_MS_local_basic") return fail(TAMCMC_IO_ERR_UNSUPPORTED, "model not covered by this loader: " + model);

    // ---- per degree: the eigen-table rows of that degree, each matched to ONE mode line within 1e-2 (:416-502), then only
    //      the modes strictly inside the slice's range (:504-557)
    std::vector<double> f, h, w, fmin, fmax;
    std::vector<int> rf, rh, rw;
    int Nf[4] = {0, 0, 0, 0};
    int lmax = 0;
    for (const auto &m : mf.modes) lmax = m.l > lmax ? m.l : lmax;
    if (lmax > 3) return fail(TAMCMC_IO_ERR_SYNTAX, "degrees above 3 are not supported");
    for (int el = 0; el <= lmax; el++) {
        bool listed = false;
        for (const auto &m : mf.modes) listed = listed || m.l == el;
        if (!listed) continue;
        for (const auto &e : mf.eigen) {
            if ((int)e[0] != el) continue;
            int match = -1, nmatch = 0;
            for (size_t k = 0; k < mf.modes.size(); k++)
                if (mf.modes[k].l == el && mf.modes[k].f > e[1] - 1e-2 && mf.modes[k].f < e[1] + 1e-2) { match = (int)k; nmatch++; }
            if (nmatch != 1) return fail(TAMCMC_IO_ERR_SYNTAX, "eigen-table frequency without a unique entry in the mode list");
            if (!(e[1] > mf.range[0] && e[1] < mf.range[1])) continue;
            f.push_back(e[1]); fmin.push_back(e[2]); fmax.push_back(e[3]); w.push_back(e[4]); h.push_back(e[5]);
            rf.push_back(mf.modes[(size_t)match].rf); rw.push_back(mf.modes[(size_t)match].rW); rh.push_back(mf.modes[(size_t)match].rH);
            Nf[el]++;
        }
    }
    const int Ntot = (int)f.size();
    if (Ntot == 0) return fail(TAMCMC_IO_ERR_EMPTY_RANGE, "no mode inside the slice's frequency range");
    if (do_amp)  // heights -> squared amplitudes pi*H*W (:569-583)
        for (int i = 0; i < Ntot; i++) h[(size_t)i] = (double)(pi * w[(size_t)i] * h[(size_t)i]);
    const std::string hname = do_amp ? "Amplitude_l" : "Height_l";

    Block height, width, freq, snlm, inc, noise;
    height.init(Ntot); width.init(Ntot); freq.init(Ntot); snlm.init(6); inc.init(1); noise.init(1);
    // ---- defaults (:586-673): Jeffreys on heights and widths, GUG on frequencies
    height.fill_vect(h, rh, hname, "Jeffreys", {Hmin, Hmax, -9999., -9999.}, 0, 0);
    const std::vector<double> wdef = {resol, mf.dnu > 0 ? mf.dnu / 3. : 20., -9999., -9999.};
    width.fill_vect(w, rw, "Width_l", "Jeffreys", wdef, 0, 0);
    for (int i = 0; i < Ntot; i++) {
        const double s = 0.01 * std::fabs(fmax[(size_t)i] - fmin[(size_t)i]);
        freq.fill("Frequency_l", rf[(size_t)i] ? "GUG" : "Fix", f[(size_t)i], {fmin[(size_t)i], fmax[(size_t)i], s, s}, i, 0);
    }
    double extra[4] = {0, 0, 0.2, 0};  // :700-705
    double trunc_c = -1;
    bool cosi = false, sini = false;
    auto no_auto = [&](const Common &c) { return c.prior == "Fix_Auto" ? fail(TAMCMC_IO_ERR_SYNTAX, c.name + " cannot be Fix_Auto") : 0; };
    // ---- the keywords of "# Controls and priors for common parameters" (:707-969)
    for (const auto &c : mf.common) {
        const std::string &n = c.name;
        if (n == "trunc_c") {
            if (c.prior != "Fix") return fail(TAMCMC_IO_ERR_SYNTAX, "trunc_c must be 'Fix'");
            trunc_c = c.v[0];
        } else if (n == "height" || n == "Height" || n == "amplitude" || n == "Amplitude") {
            if (c.prior == "Fix_Auto") {  // Jeffreys between input/Y and input*X per mode (:748-802)
                const bool amp = (n == "amplitude" || n == "Amplitude");
                for (int i = 0; i < Ntot; i++) {
                    const double s = amp ? (double)(pi * mf.dnu / 3.) : 1.0;
                    height.fill(hname, rh[(size_t)i] ? "Jeffreys" : "Fix", h[(size_t)i],
                                {s * h[(size_t)i] / c.v[0], s * h[(size_t)i] * c.v[1], -9999., -9999.}, i, 0);
                }
            } else height.fill_vect(h, rh, hname, c.prior, c.v, 0, 1);
        } else if (n == "width" || n == "Width") {
            if (c.prior == "Fix_Auto") width.fill_vect(w, rw, "Width_l", "Jeffreys", wdef, 0, 0);
            else width.fill_vect(w, rw, "Width_l", c.prior, c.v, 0, 0);  // (i0 = 0 for free widths, :845-852)
        } else if (n == "splitting_a1" || n == "Splitting_a1") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Splitting_a1", c.prior, c.v[0], c.v, 0, 1);
        } else if (n == "asphericity_eta" || n == "Asphericity_eta") {
            if (c.prior == "Fix_Auto") {  // the centrifugal term, always fixed (:887-905)
                snlm.names[1] = "Asphericity_eta0";
                snlm.prior_names[1] = "Fix";
                snlm.relax[1] = 0;
                snlm.inputs[1] = (c.v[0] == 1 && mf.dnu > 0) ? 3. / (4. * M_PI * rho * G) : 0.0;
            } else snlm.fill("Asphericity_eta", c.prior, c.v[0], c.v, 1, 1);
        } else if (n == "splitting_a3" || n == "Splitting_a3") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Splitting_a3", c.prior, c.v[0], c.v, 2, 1);
        } else if (n == "asymetry" || n == "Asymetry") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Lorentzian_asymetry", c.prior, c.v[0], c.v, 5, 1);
        } else if (n == "inclination" || n == "Inclination") {
            if (int rc = no_auto(c)) return rc;
            inc.fill("Inclination", c.prior, c.v[0] >= 90 ? 89.99999 : c.v[0], c.v, 0, 1);
        } else if (n == "sqrt(splitting_a1).cosi") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("sqrt(splitting_a1).cosi", c.prior, c.v[0], c.v, 3, 1);
            cosi = true;
        } else if (n == "sqrt(splitting_a1).sini") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("sqrt(splitting_a1).sini", c.prior, c.v[0], c.v, 4, 1);
            sini = true;
        }
        // freq_smoothness, Visibility_l*: irrelevant for a local fit (:709-720); model_fullname etc. handled above
    }
    if (cosi != sini) return fail(TAMCMC_IO_ERR_SYNTAX, "sqrt(splitting_a1).cosi and .sini must both appear");
    if (!cosi) {
        // Splitting_a1 + Inclination given: the model fits sqrt(a1) cos i and sqrt(a1) sin i (:978-1003)
        const double a1 = snlm.inputs[0], ang = (double)(inc.inputs[0] * pi / 180.);
        if (inc.prior_names[0] == "Fix" && snlm.prior_names[0] == "Fix") {
            snlm.fill("sqrt(splitting_a1).cosi", "Fix", std::sqrt(a1) * std::cos(ang), snlm.prior_col(0), 3, 0);
            snlm.fill("sqrt(splitting_a1).sini", "Fix", std::sqrt(a1) * std::sin(ang), snlm.prior_col(0), 4, 0);
        } else {
            snlm.pr(1, 0) = std::sqrt(snlm.pr(1, 0));  // upper bound of a1 -> upper bound of sqrt(a1)
            const std::string p = snlm.prior_names[0];
            snlm.fill("sqrt(splitting_a1).cosi", p, std::sqrt(a1) * std::cos(ang), snlm.prior_col(0), 3, 0);
            snlm.fill("sqrt(splitting_a1).sini", p, std::sqrt(a1) * std::sin(ang), snlm.prior_col(0), 4, 0);
        }
        if (snlm.inputs[3] < 1e-2) snlm.inputs[3] = 1e-2;
        if (snlm.inputs[4] < 1e-2) snlm.inputs[4] = 1e-2;
    }
    inc.fill("Empty", "Fix", 0, inc.prior_col(0), 0, 1);    // :1004 / :1057
    snlm.fill("Empty", "Fix", 0, snlm.prior_col(0), 0, 1);  // :1005 / :1059
    // ---- local white-noise level (set_noise_params_local, :1178-1238): mean of the background at the two ends of the range
    {
        std::vector<double> np;
        for (double v : mf.noise) {
            if (v == -1) np.push_back(0.0);
            else if (v >= 0) np.push_back(v);
        }
        if (np.size() < 1 || (np.size() - 1) % 3 != 0) return fail(TAMCMC_IO_ERR_SYNTAX, "noise parameters must be 3*Nharvey + 1 values");
        const double a = harvey_at(np, mf.range[0]), b = harvey_at(np, mf.range[1]);
        noise.names[0] = "White_Noise_N0";
        noise.prior_names[0] = "Uniform";
        noise.relax[0] = 1;
        noise.inputs[0] = (a + b) / 2.;
        noise.pr(0, 0) = (a < b ? a : b) * 0.5;
        noise.pr(1, 0) = (a > b ? a : b) * 1.5;
    }
    // ---- assemble (:1067-1118): heights, frequencies, splitting block, widths, noise, inclination, trunc_c, do_amp
    int *pl = out.plength;
    pl[0] = Ntot; pl[1] = 0; pl[2] = Nf[0]; pl[3] = Nf[1]; pl[4] = Nf[2]; pl[5] = Nf[3];
    pl[6] = 6; pl[7] = Ntot; pl[8] = 1; pl[9] = 1; pl[10] = 2;
    int N = 0;
    for (int k = 0; k < 11; k++) N += pl[k];
    Block &A = out.all;
    A.init(N);
    auto put = [&](Block &b, int pos) {  // IO_models::add_param, io_models.cpp:122-146
        for (int i = 0; i < b.n; i++) {
            A.names[(size_t)(pos + i)] = b.names[(size_t)i];
            A.prior_names[(size_t)(pos + i)] = b.prior_names[(size_t)i];
            A.inputs[(size_t)(pos + i)] = b.inputs[(size_t)i];
            A.relax[(size_t)(pos + i)] = b.relax[(size_t)i];
            for (int k = 0; k < 4; k++) A.pr(k, pos + i) = b.pr(k, i);
        }
    };
    int p0 = 0;
    put(height, p0); p0 += pl[0] + pl[1];
    put(freq, p0); p0 += pl[2] + pl[3] + pl[4] + pl[5];
    put(snlm, p0); p0 += pl[6];
    put(width, p0); p0 += pl[7];
    put(noise, p0); p0 += pl[8];
    put(inc, p0); p0 += pl[9];
    A.fill("Truncation parameter", "Fix", trunc_c > 0 ? trunc_c : 10000., {}, p0, 1);  // non-positive c -> full Lorentzian (:1108-1110)
    A.fill("Switch for fit of Amplitudes or Heights", "Fix", (double)do_amp, {}, p0 + 1, 1);
    for (int k = 0; k < 10; k++) out.extra[k] = k < 4 ? extra[k] : 0.0;
    out.range[0] = mf.range[0]; out.range[1] = mf.range[1];
    out.dnu = mf.dnu; out.c_l = mf.c_l;
    out.model_id = TAMCMC_MODEL_MS_LOCAL_BASIC;
    out.prior_class = 3;  // io_local, Config/default/priors_ctrl.list
    out.model_name = model;
    for (int i = 0; i < N; i++)
        if (prior_id(A.prior_names[(size_t)i]) < 0) return fail(TAMCMC_IO_ERR_SYNTAX, "unknown prior keyword: " + A.prior_names[(size_t)i]);
    return TAMCMC_IO_OK;
}

// set_noise_params (io_ms_global.cpp:1447-1536): first two Harvey profiles fixed, the third and the white noise Gaussian
void fill_noise_global(const ModelFile &mf, Block &noise) {
    static const char *nn[3] = {"Harvey-Noise_H", "Harvey-Noise_tc", "Harvey-Noise_p"};
    for (int k = 0; k < 9; k++) noise.names[(size_t)k] = nn[k % 3];
    noise.names[9] = "White_Noise_N0";
    for (int k = 0; k < 10; k++) {
        noise.inputs[(size_t)k] = mf.noise[(size_t)k];
        const bool free_k = k >= 6;
        noise.prior_names[(size_t)k] = free_k ? "Gaussian" : "Fix";
        noise.relax[(size_t)k] = free_k ? 1 : 0;
    }
    for (int g3 = 0; g3 < 3; g3++) {  // an absent / non-positive profile is switched off: (0, 0, 1) fixed
        const int b = 3 * g3;
        if (noise.inputs[(size_t)b] <= 0 || noise.inputs[(size_t)b + 1] <= 0 || noise.inputs[(size_t)b + 2] <= 0) {
            for (int k = 0; k < 3; k++) { noise.prior_names[(size_t)(b + k)] = "Fix"; noise.relax[(size_t)(b + k)] = 0; }
            noise.inputs[(size_t)b] = 0; noise.inputs[(size_t)b + 1] = 0; noise.inputs[(size_t)b + 2] = 1;
        }
    }
    const auto &S = mf.noise_s2;
    for (int k = 6; k <= 9; k++) noise.pr(0, k) = S[(size_t)k][0];
    noise.pr(1, 6) = (S[6][1] + S[6][2]) * 3. / 2;
    noise.pr(1, 7) = (S[7][1] + S[7][2]) * 3. / 2;
    noise.pr(1, 8) = (S[8][1] != 0) ? (S[8][1] + S[8][2]) * 3. / 2 : noise.pr(0, 8) * 0.1;
    noise.pr(1, 9) = noise.pr(0, 9) * 0.1;  // Gaussian white-noise prior
    const double floor_rel[4] = {0.05, 0.005, 0.05, 0.0005};
    for (int k = 6; k <= 9; k++)
        if (noise.pr(1, k) / noise.pr(0, k) <= floor_rel[k - 6] && noise.prior_names[(size_t)k] != "Fix") noise.pr(1, k) = noise.pr(0, k) * floor_rel[k - 6];
}

// build_init_MS_Global (io_ms_global.cpp:362-1445) for model_MS_Global_aj_HarveyLike and
// model_MS_Global_a1etaa3_HarveyLike_Classic, + set_noise_params (:1447-1536)
int build_global(const ModelFile &mf, double resol, tamcmc_inputs &out) {
    const long double pi = 3.141592653589793238L;
    const double Hmin = 1, Hmax = 10000;
    std::string model;
    int do_amp = 0;
    for (const auto &c : mf.common) {
        if (c.name == "model_fullname") model = c.prior;
        if (c.name == "fit_squareAmplitude_instead_Height") {
            if (c.prior != "bool") return fail(TAMCMC_IO_ERR_SYNTAX, "fit_squareAmplitude_instead_Height must be 'bool'");
            do_amp = c.v[0] != 0;
        }
    }
    if (model.empty()) return fail(TAMCMC_IO_ERR_SYNTAX, "the .model file has no model_fullname");
    const bool aj = (model == "model_MS_Global_aj_HarveyLike"), classic = (model == "model_MS_Global_a1etaa3_HarveyLike_Classic");
    if (!aj && !classic) return fail(TAMCMC_IO_ERR_UNSUPPORTED, "model not covered by this loader: " + model);
    double extra[10] = {1, 2., 1e6, 0.50, 0.20, 0.15, 0.05, 0.05, 0, -1};  // :403-413
    if (aj) extra[9] = 9;                                                    // :494-499
    int lmax = 0;
    for (const auto &m : mf.modes) lmax = m.l > lmax ? m.l : lmax;
    if (lmax > 3) return fail(TAMCMC_IO_ERR_SYNTAX, "degrees above 3 are not supported");
    // ---- frequencies of every degree, heights and widths of l=0 only (:530-580)
    std::vector<double> f, fmin, fmax, h, w;
    std::vector<int> rf, rh, rw;
    int Nf[4] = {0, 0, 0, 0};
    for (int el = 0; el <= lmax; el++)
        for (const auto &e : mf.eigen) {
            if ((int)e[0] != el) continue;
            int match = -1, nmatch = 0;
            for (size_t k = 0; k < mf.modes.size(); k++)
                if (mf.modes[k].l == el && mf.modes[k].f > e[1] - 1e-2 && mf.modes[k].f < e[1] + 1e-2) { match = (int)k; nmatch++; }
            if (nmatch != 1) return fail(TAMCMC_IO_ERR_SYNTAX, "eigen-table frequency without a unique entry in the mode list");
            f.push_back(e[1]); fmin.push_back(e[2]); fmax.push_back(e[3]);
            rf.push_back(mf.modes[(size_t)match].rf);
            if (el == 0) {
                w.push_back(e[4]); h.push_back(e[5]);
                rw.push_back(mf.modes[(size_t)match].rW); rh.push_back(mf.modes[(size_t)match].rH);
            }
            Nf[el]++;
        }
    const int Nh = (int)h.size(), Nfreq = (int)f.size();
    if (Nh == 0) return fail(TAMCMC_IO_ERR_EMPTY_RANGE, "no l=0 mode in the eigen table");
    if (do_amp)
        for (int i = 0; i < Nh; i++) h[(size_t)i] = (double)(pi * w[(size_t)i] * h[(size_t)i]);
    const std::string hname = do_amp ? "Amplitude_l0" : "Height_l0";
    Block height, width, freq, snlm, vis, inc, noise;
    height.init(Nh); width.init(Nh); freq.init(Nfreq); vis.init(lmax); inc.init(1); noise.init(10);
    // ---- defaults (:640-677)
    for (int i = 0; i < Nh; i++) height.fill(hname, rh[(size_t)i] ? "Jeffreys" : "Fix", h[(size_t)i], {Hmin, Hmax, -9999., -9999.}, i, 0);
    const std::vector<double> wdef = {resol, mf.dnu / 3., -9999., -9999.};
    auto wval = [&](int i) { return w[(size_t)i] < mf.dnu / 3. ? w[(size_t)i] : mf.dnu / 3.1; };
    for (int i = 0; i < Nh; i++) width.fill("Width_l0", rw[(size_t)i] ? "Jeffreys" : "Fix", wval(i), wdef, i, 0);
    for (int i = 0; i < Nfreq; i++)
        freq.fill("Frequency_l", rf[(size_t)i] ? "GUG" : "Fix", f[(size_t)i], {fmin[(size_t)i], fmax[(size_t)i], 0.01 * mf.dnu, 0.01 * mf.dnu}, i, 0);
    // ---- splitting block (:700-726): classic 6 slots; aj 12 a-coefficients + eta0 switch + asymmetry
    snlm.init(aj ? 14 : 6);
    if (aj) snlm.fill("eta0_switch", "Fix", 0, {}, 12, 0);
    double trunc_c = -1;
    bool cosi = false, sini = false;
    int aj_count = 0;
    auto no_auto = [&](const Common &c) { return c.prior == "Fix_Auto" ? fail(TAMCMC_IO_ERR_SYNTAX, c.name + " cannot be Fix_Auto") : 0; };
    static const char *aj_names[12] = {"a1_0", "a1_1", "a2_0", "a2_1", "a3_0", "a3_1", "a4_0", "a4_1", "a5_0", "a5_1", "a6_0", "a6_1"};
    for (const auto &c : mf.common) {
        const std::string &n = c.name;
        if (n == "freq_smoothness" || n == "Freq_smoothness") {
            if (c.prior != "bool") return fail(TAMCMC_IO_ERR_SYNTAX, "freq_smoothness must be 'bool'");
            extra[0] = c.v[0];
            extra[1] = c.v[1];
        } else if (n == "trunc_c") {
            if (c.prior == "Fix") trunc_c = c.v[0];
            else if (!to_double(c.prior, &trunc_c)) return fail(TAMCMC_IO_ERR_SYNTAX, "trunc_c must be 'Fix <value>' (or a bare value)");
        } else if (n == "Frequency" || n == "frequency") {
            if (c.prior != "GUG" && c.prior != "Uniform") return fail(TAMCMC_IO_ERR_SYNTAX, "Frequency prior must be GUG or Uniform");
            for (int i = 0; i < Nfreq; i++) {
                const std::vector<double> v = (c.prior == "GUG") ? std::vector<double>{fmin[(size_t)i], fmax[(size_t)i], c.v[3], c.v[4]}
                                                                 : std::vector<double>{fmin[(size_t)i], fmax[(size_t)i], -9999., -9999.};
                freq.fill("Frequency_l", rf[(size_t)i] ? c.prior : "Fix", f[(size_t)i], v, i, 0);
            }
        } else if (n == "height" || n == "Height" || n == "amplitude" || n == "Amplitude") {
            if (int rc = no_auto(c)) return rc;
            for (int i = 0; i < Nh; i++) height.fill(hname, rh[(size_t)i] ? c.prior : "Fix", h[(size_t)i], c.v, i, 0);  // values from the FIRST number (:880-886)
        } else if (n == "width" || n == "Width") {
            const bool au = (c.prior == "Fix_Auto");
            for (int i = 0; i < Nh; i++) width.fill("Width_l", rw[(size_t)i] ? (au ? "Jeffreys" : c.prior) : "Fix", wval(i), au ? wdef : c.v, i, 0);
        } else if (n == "splitting_a1" || n == "Splitting_a1") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Splitting_a1", c.prior, c.v[0], c.v, 0, 1);
        } else if (n == "asphericity_eta" || n == "Asphericity_eta") {  // kept as a fixed 0: the model computes eta0 itself (:956-962)
            snlm.names[1] = "Asphericity_eta";
            snlm.prior_names[1] = "Fix";
            snlm.relax[1] = 0;
            snlm.inputs[1] = 0;
        } else if (n == "splitting_a3" || n == "Splitting_a3") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Splitting_a3", c.prior, c.v[0], c.v, 2, 1);
        } else if (n == "asymetry" || n == "Asymetry") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Lorentzian_asymetry", c.prior, c.v[0], c.v, aj ? snlm.n - 1 : 5, 1);
        } else if (n == "visibility_l1" || n == "Visibility_l1" || n == "visibility_l2" || n == "Visibility_l2" || n == "visibility_l3" ||
                   n == "Visibility_l3") {
            if (int rc = no_auto(c)) return rc;
            const int k = n.back() - '0';
            if (lmax >= k) vis.fill(std::string("Visibility_l") + n.back(), c.prior, c.v[0], c.v, k - 1, 1);
        } else if (n == "inclination" || n == "Inclination") {
            if (int rc = no_auto(c)) return rc;
            inc.fill("Inclination", c.prior, c.v[0] >= 90 ? 89.99999 : c.v[0], c.v, 0, 1);
        } else if (n == "sqrt(splitting_a1).cosi" || n == "sqrt(splitting_a1).sini") {
            if (int rc = no_auto(c)) return rc;
            const bool is_cos = (n == "sqrt(splitting_a1).cosi");
            snlm.fill(n, c.prior, c.v[0], c.v, is_cos ? 3 : 4, 1);
            (is_cos ? cosi : sini) = true;
        } else if (aj) {  // settings_aj_splittings, :1718-1850
            for (int k = 0; k < 12; k++)
                if (n == aj_names[k]) {
                    if (c.prior == "Fix_Auto") return fail(TAMCMC_IO_ERR_SYNTAX, n + " cannot be Fix_Auto");
                    snlm.fill(n, c.prior, c.v[0], c.v, k, 1);
                    aj_count++;
                }
        }
    }
    if (aj && aj_count != 12) return fail(TAMCMC_IO_ERR_SYNTAX, "the aj model needs the 12 keywords a1_0 ... a6_1 (:1178-1181)");
    if (cosi != sini) return fail(TAMCMC_IO_ERR_SYNTAX, "sqrt(splitting_a1).cosi and .sini must both appear");
    if (cosi) {
        if (classic) return fail(TAMCMC_IO_ERR_SYNTAX, "the Classic model takes Splitting_a1 and Inclination (:1289-1294)");
        inc.fill("Empty", "Fix", 0, inc.prior_col(0), 0, 1);
        snlm.fill("Empty", "Fix", 0, snlm.prior_col(0), 0, 1);
    }
    fill_noise_global(mf, noise);
    // ---- assemble (:1327-1376): heights, visibilities, frequencies, splitting block, widths, noise, inclination, trunc_c, do_amp
    int *pl = out.plength;
    pl[0] = Nh; pl[1] = lmax; pl[2] = Nf[0]; pl[3] = Nf[1]; pl[4] = Nf[2]; pl[5] = Nf[3];
    pl[6] = snlm.n; pl[7] = Nh; pl[8] = 10; pl[9] = 1; pl[10] = 2;
    int N = 0;
    for (int k = 0; k < 11; k++) N += pl[k];
    Block &A = out.all;
    A.init(N);
    auto put = [&](Block &b, int pos) {
        for (int i = 0; i < b.n; i++) {
            A.names[(size_t)(pos + i)] = b.names[(size_t)i];
            A.prior_names[(size_t)(pos + i)] = b.prior_names[(size_t)i];
            A.inputs[(size_t)(pos + i)] = b.inputs[(size_t)i];
            A.relax[(size_t)(pos + i)] = b.relax[(size_t)i];
            for (int k = 0; k < 4; k++) A.pr(k, pos + i) = b.pr(k, i);
        }
    };
    int p0 = 0;
    put(height, p0); p0 += pl[0];
    put(vis, p0); p0 += pl[1];
    put(freq, p0); p0 += pl[2] + pl[3] + pl[4] + pl[5];
    put(snlm, p0); p0 += pl[6];
    put(width, p0); p0 += pl[7];
    put(noise, p0); p0 += pl[8];
    put(inc, p0); p0 += pl[9];
    A.fill("Truncation parameter", "Fix", trunc_c > 0 ? trunc_c : 10000., {}, p0, 1);
    A.fill("Switch for fit of Amplitudes or Heights", "Fix", (double)do_amp, {}, p0 + 1, 1);
    for (int k = 0; k < 10; k++) out.extra[k] = extra[k];
    out.range[0] = mf.range[0]; out.range[1] = mf.range[1];
    out.dnu = mf.dnu; out.c_l = mf.c_l;
    out.model_id = aj ? TAMCMC_MODEL_MS_GLOBAL_AJ : TAMCMC_MODEL_MS_GLOBAL_A1ETAA3_CLASSIC;
    out.prior_class = 2;  // io_MS_Global, Config/default/priors_ctrl.list
    out.model_name = model;
    for (int i = 0; i < N; i++)
        if (prior_id(A.prior_names[(size_t)i]) < 0) return fail(TAMCMC_IO_ERR_SYNTAX, "unknown prior keyword: " + A.prior_names[(size_t)i]);
    return TAMCMC_IO_OK;
}

// build_init_asymptotic (io_asymptotic.cpp:32-838, settings_aj_splittings_RGB :841-955) for the red-giant models
// model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (id 25) and model_RGB_asympt_aj_CteWidth_HarveyLike_v4 (id 27); width-law start values
// and priors: set_width_App2016_params_v2 (io_ms_global.cpp:1625-1720).  The l=1 modes are not listed: the mode list carries ONE
// l=1 placeholder line, whose slot becomes the block [delta01, DP1, alpha_g, q, sigma_Hl1, -, Wfactor, Hfactor, fref.., ferr..].
int build_asymptotic(const ModelFile &mf, double resol, tamcmc_inputs &out) {
    const long double pi = 3.141592653589793238L;
    const double Hmin = 1, Hmax = 10000;
    std::string model;
    int do_amp = 0;
    for (const auto &c : mf.common) {
        if (c.name == "model_fullname") model = c.prior;
        if (c.name == "fit_squareAmplitude_instead_Height") {
            if (c.prior != "bool") return fail(TAMCMC_IO_ERR_SYNTAX, "fit_squareAmplitude_instead_Height must be 'bool'");
            do_amp = c.v[0] != 0;
        }
    }
    if (model.empty()) return fail(TAMCMC_IO_ERR_SYNTAX, "the .model file has no model_fullname");
    const bool app = (model == "model_RGB_asympt_aj_AppWidth_HarveyLike_v4"), cte = (model == "model_RGB_asympt_aj_CteWidth_HarveyLike_v4");
    if (!app && !cte) return fail(TAMCMC_IO_ERR_UNSUPPORTED, "model not covered by this loader: " + model);
    double numax = mf.numax, err_numax = mf.err_numax;
    if (app && numax <= 0) return fail(TAMCMC_IO_ERR_SYNTAX, "the AppWidth model needs a positive numax ('!n' line; :100-109)");
    if (cte && numax != -9999 && numax <= 0) return fail(TAMCMC_IO_ERR_SYNTAX, "numax, when given, must be positive (:113-121)");
    if (numax > 0 && err_numax <= 0) err_numax = 0.05 * numax;  // :397-401
    int lmax = 0;
    for (const auto &m : mf.modes) lmax = m.l > lmax ? m.l : lmax;
    if (lmax > 3) return fail(TAMCMC_IO_ERR_SYNTAX, "degrees above 3 are not supported");
    // ---- eigen table by degree (:146-197): frequencies of every degree, heights and widths of l=0
    std::vector<double> f, fmin, fmax, h, w;
    std::vector<int> rf, rh, rw;
    int Nf[4] = {0, 0, 0, 0};
    for (int el = 0; el <= lmax; el++)
        for (const auto &e : mf.eigen) {
            if ((int)e[0] != el) continue;
            int match = -1, nmatch = 0;
            for (size_t k = 0; k < mf.modes.size(); k++)
                if (mf.modes[k].l == el && mf.modes[k].f > e[1] - 1e-2 && mf.modes[k].f < e[1] + 1e-2) { match = (int)k; nmatch++; }
            if (nmatch != 1) return fail(TAMCMC_IO_ERR_SYNTAX, "eigen-table frequency without a unique entry in the mode list");
            f.push_back(e[1]); fmin.push_back(e[2]); fmax.push_back(e[3]);
            rf.push_back(mf.modes[(size_t)match].rf);
            if (el == 0) {
                w.push_back(e[4]); h.push_back(e[5]);
                rw.push_back(mf.modes[(size_t)match].rW); rh.push_back(mf.modes[(size_t)match].rH);
            }
            Nf[el]++;
        }
    const int Nh = (int)h.size();
    if (Nh < 2) return fail(TAMCMC_IO_ERR_EMPTY_RANGE, "the red-giant models need at least two l=0 modes in the eigen table");
    // the slot of the l=1 block is opened when the walk over the listed frequencies meets the first l=1 entry (:327-339): without
    // one, the reference writes the l=2 frequencies over the block
    if (Nf[1] < 1) return fail(TAMCMC_IO_ERR_SYNTAX, "the mode list needs one l=1 placeholder line (its slot holds the mixed-mode parameters)");
    if (do_amp)
        for (int i = 0; i < Nh; i++) h[(size_t)i] = (double)(pi * w[(size_t)i] * h[(size_t)i]);
    const std::string hname = do_amp ? "Amplitude_l0_rgb" : "Height_l0_rgb";
    // ---- nodes of the bias spline (:251-283)
    const int Nferr = (int)mf.hyper.size();
    if (Nferr < 1) return fail(TAMCMC_IO_ERR_SYNTAX, "no hyper-prior rows: the v4 red-giant models take the nodes of the frequency-bias spline there");
    const size_t hcols = mf.hyper[0].size();
    if (hcols > 1 && (int)mf.hyper_names.size() != Nferr) return fail(TAMCMC_IO_ERR_SYNTAX, "hyper priors: every row needs a prior keyword");
    int Nfix = 0;
    for (int i = 0; i + 1 < Nferr; i++) {
        if (mf.hyper[(size_t)i + 1].size() != hcols) return fail(TAMCMC_IO_ERR_SYNTAX, "hyper priors: rows of different lengths");
        if (mf.hyper[(size_t)i + 1][0] < mf.hyper[(size_t)i][0]) return fail(TAMCMC_IO_ERR_SYNTAX, "hyper priors: the node frequencies must increase");
        if (hcols > 1 && mf.hyper_names[(size_t)i] == "Fix") Nfix++;
    }
    const int nnames = (int)mf.hyper_names.size();
    if (Nfix != nnames - 1 && Nfix != 0) return fail(TAMCMC_IO_ERR_SYNTAX, "hyper priors: either all 'Fix' or none (:262-266)");
    std::vector<double> fref((size_t)Nferr), ferr((size_t)Nferr, 0.0);
    for (int i = 0; i < Nferr; i++) {
        fref[(size_t)i] = mf.hyper[(size_t)i][0];
        if (hcols > 1) ferr[(size_t)i] = mf.hyper[(size_t)i][1];
    }
    const int Nmixed = 7 + 2 * Nferr + 1;
    Block height, width, freq, snlm, vis, inc, noise;
    height.init(Nh); vis.init(lmax); inc.init(1); noise.init(10); snlm.init(10);
    width.init(app ? 6 : 1);
    freq.init(Nf[0] + Nmixed + Nf[2] + Nf[3]);
    for (int i = 0; i < Nh; i++) height.fill(hname, rh[(size_t)i] ? "Jeffreys" : "Fix", h[(size_t)i], {Hmin, Hmax, -9999., -9999.}, i, 0);
    // l=0, then (after the block) l=2 and l=3 frequencies (:321-340)
    auto put_freqs = [&](const std::string &name, const std::string &prior, double s1, double s2) {
        int cpt = 0;
        for (int i = 0; i < (int)f.size(); i++) {
            if (i < Nf[0] || i >= Nf[0] + Nf[1]) {
                freq.fill(name, rf[(size_t)i] ? prior : "Fix", f[(size_t)i], {fmin[(size_t)i], fmax[(size_t)i], s1, s2}, cpt, 0);
                cpt++;
            } else if (i == Nf[0]) cpt += Nmixed;
        }
    };
    put_freqs("Frequency_RGB_l", "GUG", 0.0025 * mf.dnu, 0.0025 * mf.dnu);
    {   // spline nodes: fixed frequencies, free (or as the file says) bias values (:343-378)
        int cpt = Nf[0] + 7 + 1;
        for (int i = 0; i < Nferr; i++) freq.fill("fref_bias", "Fix", fref[(size_t)i], {}, cpt++, 0);
        if (hcols == 1) {
            for (int i = 0; i < Nferr; i++) {
                const double lo = (i == 0) ? -mf.dnu / 2 : -mf.dnu / 20, hi = (i == Nferr - 1 && Nferr > 1) ? mf.dnu / 2 : mf.dnu / 20;
                freq.fill("ferr_bias", "Uniform", ferr[(size_t)i], {lo, hi, -9999., -9999.}, cpt++, 0);
            }
        } else {
            for (int i = 0; i < Nferr; i++) {
                std::vector<double> v(4, -9999.);
                for (size_t k = 0; k + 2 < hcols && k < 4; k++) v[k] = mf.hyper[(size_t)i][2 + k];
                freq.fill("ferr_bias", mf.hyper_names[(size_t)i], ferr[(size_t)i], v, cpt++, 0);
            }
        }
    }
    double extra[10] = {1, 2., 0.2, 0, 3, 0, 0, 0, 0, 0};  // :413-424: smoothness on, 2 muHz, |a3/a1| <= 0.2, no height normalisation, v4 priors
    double trunc_c = -1, model_type = -1, bias_type = -1;
    int aj_count = 0;
    auto no_auto = [&](const Common &c) { return c.prior == "Fix_Auto" ? fail(TAMCMC_IO_ERR_SYNTAX, c.name + " cannot be Fix_Auto") : 0; };
    auto fixed_only = [&](const Common &c) { return c.prior != "Fix" ? fail(TAMCMC_IO_ERR_SYNTAX, c.name + " must be 'Fix <value>'") : 0; };
    static const struct { const char *a, *b, *c; int pos; } rot[8] = {
        {"rot_env", "Rot_env", "a1_env", 0}, {"rot_core", "Rot_core", "a1_core", 1}, {"a2_core", "", "", 2}, {"a2_env", "", "", 3},
        {"a3_env", "", "", 4}, {"a4_env", "", "", 5}, {"a5_env", "", "", 6}, {"a6_env", "", "", 7}};  // slots as the reference fills them (:846-925)
    for (const auto &c : mf.common) {
        const std::string &n = c.name;
        if (n == "freq_smoothness" || n == "Freq_smoothness") {
            if (c.prior != "bool") return fail(TAMCMC_IO_ERR_SYNTAX, "freq_smoothness must be 'bool'");
            extra[0] = c.v[0];
            extra[1] = c.v[1];
        } else if (n == "trunc_c") {
            if (int rc = fixed_only(c)) return rc;
            trunc_c = c.v[0];
        } else if (n == "model_type") {
            if (int rc = fixed_only(c)) return rc;
            model_type = c.v[0];
        } else if (n == "bias_type") {
            if (int rc = fixed_only(c)) return rc;
            bias_type = (Nfix != nnames - 1) ? c.v[0] : 0;  // all nodes fixed: no bias (:462-468)
        } else if (n == "Frequency" || n == "frequency") {
            if (c.prior != "GUG" && c.prior != "Uniform") return fail(TAMCMC_IO_ERR_SYNTAX, "Frequency prior must be GUG or Uniform");
            put_freqs("Frequency_l", c.prior, c.prior == "GUG" ? c.v[3] : -9999., c.prior == "GUG" ? c.v[4] : -9999.);
        } else if (n == "delta01") {
            if (c.prior == "Fix_Auto") freq.fill("delta01", "Uniform", 0.5 * mf.dnu / 100, {-mf.dnu / 100, mf.dnu / 100, -9999., -9999.}, Nf[0], 0);
            else freq.fill("delta01", c.prior, c.v[0], c.v, Nf[0], 1);
        } else if (n == "DP1" || n == "alpha_g" || n == "q" || n == "sigma_Hl1" || n == "Wfactor" || n == "Hfactor") {
            if (int rc = no_auto(c)) return rc;
            const int off = n == "DP1" ? 1 : n == "alpha_g" ? 2 : n == "q" ? 3 : n == "sigma_Hl1" ? 4 : n == "Wfactor" ? 6 : 7;
            freq.fill(n, c.prior, c.v[0], c.v, Nf[0] + off, 1);
        } else if (n == "height" || n == "Height" || n == "amplitude" || n == "Amplitude") {
            if (int rc = no_auto(c)) return rc;
            for (int i = 0; i < Nh; i++) height.fill(hname, rh[(size_t)i] ? c.prior : "Fix", h[(size_t)i], c.v, i, 0);
        } else if ((n == "width" || n == "Width") && cte) {  // one width: the mean of the listed l=0 widths (:633-648)
            if (c.prior != "Fix_Auto") return fail(TAMCMC_IO_ERR_SYNTAX, "the CteWidth model takes 'Width Fix_Auto'");
            double mean = 0;
            for (int i = 0; i < Nh; i++) mean = mean + w[(size_t)i] / Nh;
            width.fill("Width_l", "Jeffreys", mean, {resol, mf.dnu / 3., -9999., -9999.}, 0, 0);
        } else if (n == "asymetry" || n == "Asymetry") {
            if (int rc = no_auto(c)) return rc;
            snlm.fill("Lorentzian_asymetry", c.prior, c.v[0], c.v, 9, 1);
        } else if (n == "visibility_l1" || n == "Visibility_l1" || n == "visibility_l2" || n == "Visibility_l2" || n == "visibility_l3" ||
                   n == "Visibility_l3") {
            if (int rc = no_auto(c)) return rc;
            const int k = n.back() - '0';
            if (lmax >= k) vis.fill(std::string("Visibility_l") + n.back(), c.prior, c.v[0], c.v, k - 1, 1);
        } else if (n == "inclination" || n == "Inclination") {
            if (int rc = no_auto(c)) return rc;
            inc.fill("Inclination", c.prior, c.v[0] >= 90 ? 89.99999 : c.v[0], c.v, 0, 1);
        } else if (n == "asphericity_eta" || n == "Asphericity_eta") {
            snlm.names[8] = "eta0_switch"; snlm.prior_names[8] = "Fix"; snlm.relax[8] = 0; snlm.inputs[8] = 0;
        } else if (n == "eta0_switch") {
            if (int rc = fixed_only(c)) return rc;
            snlm.fill("eta0_switch", "Fix", c.v[0], c.v, 8, 1);
        } else {
            for (const auto &r : rot)
                if (n == r.a || n == r.b || n == r.c) {
                    if (int rc = no_auto(c)) return rc;
                    snlm.fill(r.a, c.prior, c.v[0], c.v, r.pos, 1);
                    aj_count++;
                }
        }
    }
    if (aj_count != 8) return fail(TAMCMC_IO_ERR_SYNTAX, "the red-giant models need the 8 keywords rot_env, rot_core, a2_core, a2_env, a3_env ... a6_env (:712-718)");
    if ((model_type == -1) != (bias_type == -1)) return fail(TAMCMC_IO_ERR_SYNTAX, "model_type and bias_type must both be given (:744-748)");
    if (model_type == -1) return fail(TAMCMC_IO_ERR_SYNTAX, "the v4 red-giant models read model_type, bias_type and the node count from the vector: both keywords are needed");
    if (cte && width.names[0] == "Empty") return fail(TAMCMC_IO_ERR_SYNTAX, "the CteWidth model needs the keyword 'Width Fix_Auto'");
    if (app) {  // set_width_App2016_params_v2, io_ms_global.cpp:1625-1720
        double o[6];
        o[0] = std::fabs(numax); o[1] = std::fabs(numax);
        o[2] = std::fabs(4. / 2150. * numax + (1. - 1000. * 4. / 2150.));
        o[3] = std::fabs(0.8 / 2150. * numax + (4.5 - 1000. * 0.8 / 2150.));
        o[4] = std::fabs(3400. / 2150. * numax + (1000. - 1000. * 3400. / 2150.));
        o[5] = std::fabs(2.8 / 2200. * numax + (1. - 2.8 / 2200. * 1.));
        if (numax < 800) o[3] = o[3] / 5;
        width.fill("width:Appourchaux_v2:numax", "Gaussian", o[0], {o[0], err_numax, -9999., -9999.}, 0, 0);
        width.fill("width:Appourchaux_v2:nudip", "Gaussian", o[1], {o[1], err_numax, -9999., -9999.}, 1, 0);
        width.fill("width:Appourchaux_v2:alpha", "Uniform", o[2], {0., 6., -9999., -9999.}, 2, 0);
        width.fill("width:Appourchaux_v2:Gamma_alpha", "Uniform", o[3], {0., 10., -9999., -9999.}, 3, 0);
        width.fill("width:Appourchaux_v2:Wdip", "Gaussian", o[4], {o[4], o[4] * 0.25, -9999., -9999.}, 4, 0);
        width.fill("width:Appourchaux_v2:DeltaGammadip", "Uniform", o[5], {0., 15., -9999., -9999.}, 5, 0);
    }
    fill_noise_global(mf, noise);
    // ---- assemble (:724-822)
    int *pl = out.plength;
    pl[0] = Nh; pl[1] = lmax; pl[2] = Nf[0]; pl[3] = Nmixed; pl[4] = Nf[2]; pl[5] = Nf[3];
    pl[6] = snlm.n; pl[7] = width.n; pl[8] = 10; pl[9] = 1; pl[10] = 6;
    int N = 0;
    for (int k = 0; k < 11; k++) N += pl[k];
    Block &A = out.all;
    A.init(N);
    auto put = [&](Block &b, int pos) {
        for (int i = 0; i < b.n; i++) {
            A.names[(size_t)(pos + i)] = b.names[(size_t)i];
            A.prior_names[(size_t)(pos + i)] = b.prior_names[(size_t)i];
            A.inputs[(size_t)(pos + i)] = b.inputs[(size_t)i];
            A.relax[(size_t)(pos + i)] = b.relax[(size_t)i];
            for (int k = 0; k < 4; k++) A.pr(k, pos + i) = b.pr(k, i);
        }
    };
    int p0 = 0;
    put(height, p0); p0 += pl[0];
    put(vis, p0); p0 += pl[1];
    put(freq, p0); p0 += pl[2] + pl[3] + pl[4] + pl[5];
    put(snlm, p0); p0 += pl[6];
    put(width, p0); p0 += pl[7];
    put(noise, p0); p0 += pl[8];
    put(inc, p0); p0 += pl[9];
    A.fill("Truncation parameter", "Fix", trunc_c > 0 ? trunc_c : 10000., {}, p0, 1);
    A.fill("Switch for fit of Amplitudes or Heights", "Fix", (double)do_amp, {}, p0 + 1, 1);
    A.fill("Maximum limit on random values generated by N(0,sigma_m)", "Fix", mf.dnu / 10., {}, p0 + 2, 1);
    A.fill("model type ", "Fix", model_type, {}, p0 + 3, 1);
    A.fill("bias type ", "Fix", bias_type, {}, p0 + 4, 1);
    A.fill("Nferr ", "Fix", (double)Nferr, {}, p0 + 5, 1);
    if (bias_type == 0 && Nfix != nnames - 1)  // no bias asked for: the node values are pinned to 0 (:825-833)
        for (int i = 0; i < N; i++)
            if (A.names[(size_t)i] == "ferr_bias") A.fill("ferr_bias", "Fix", 0, {}, i, 1);
    for (int k = 0; k < 10; k++) out.extra[k] = extra[k];
    out.range[0] = mf.range[0]; out.range[1] = mf.range[1];
    out.dnu = mf.dnu; out.c_l = mf.c_l;
    out.model_id = app ? 25 : 27;  // TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4 / _CTEWIDTH_V4
    out.prior_class = 4;           // io_asymptotic, Config/default/priors_ctrl.list
    out.model_name = model;
    for (int i = 0; i < N; i++)
        if (prior_id(A.prior_names[(size_t)i]) < 0) return fail(TAMCMC_IO_ERR_SYNTAX, "unknown prior keyword: " + A.prior_names[(size_t)i]);
    return TAMCMC_IO_OK;
}

}  // namespace

extern "C" {

const char *tamcmc_io_last_error(void) { return g_err.c_str(); }

void tamcmc_io_free(void *p) { std::free(p); }

int tamcmc_io_read_data(const char *path, double **table, int64_t *nrows, int64_t *ncols) {
    if (!path || !table || !nrows || !ncols) return fail(TAMCMC_IO_ERR_ARG, "null argument");
    std::ifstream f(path);
    if (!f.is_open()) return fail(TAMCMC_IO_ERR_OPEN, std::string("cannot open ") + path);
    std::vector<double> vals;
    size_t nc = 0, nr = 0;
    bool header_done = false, labels_seen = false, units_seen = false;
    for (std::string raw; std::getline(f, raw);) {
        const std::string ln = trim(raw);
        if (!header_done) {  // '#' lines, then at most one '!' line, then at most one '*' line (config.cpp:936-1003)
            if (!ln.empty() && ln[0] == '#' && !labels_seen && !units_seen) continue;
            if (!ln.empty() && ln[0] == '!' && !labels_seen && !units_seen) { labels_seen = true; continue; }
            if (!ln.empty() && ln[0] == '*' && !units_seen) { units_seen = true; continue; }
            header_done = true;
        }
        if (ln.empty()) continue;
        const auto w = split(ln, " \t");
        if (nr == 0) nc = w.size();
        for (size_t k = 0; k < nc; k++) {
            double v = std::nan("");
            if (k < w.size() && !to_double(w[k], &v)) v = std::nan("");
            vals.push_back(v);
        }
        nr++;
    }
    if (nr == 0 || nc == 0) return fail(TAMCMC_IO_ERR_SYNTAX, "no data rows");
    double *t = (double *)std::malloc(vals.size() * sizeof(double));
    if (!t) return fail(TAMCMC_IO_ERR_ARG, "out of memory");
    std::memcpy(t, vals.data(), vals.size() * sizeof(double));
    *table = t;
    *nrows = (int64_t)nr;
    *ncols = (int64_t)nc;
    return TAMCMC_IO_OK;
}

int tamcmc_io_select_range(const double *table, int64_t nrows, int64_t ncols, int x_col, double xmin, double xmax, int64_t *imin,
                           int64_t *imax) {
    if (!table || !imin || !imax || nrows < 1 || x_col < 0 || x_col >= ncols) return fail(TAMCMC_IO_ERR_ARG, "bad argument");
    int64_t a = 0;
    while (a < nrows && table[a * ncols + x_col] < xmin) a++;
    if (a >= nrows) return fail(TAMCMC_IO_ERR_EMPTY_RANGE, "the requested range starts beyond the data");
    int64_t b = a;
    while (b < nrows && table[b * ncols + x_col] < xmax) b++;
    *imin = a;
    *imax = b;
    return TAMCMC_IO_OK;
}

int tamcmc_io_load_model_local(const char *model_path, int slice_ind, double resol, tamcmc_inputs **out) {
    if (!model_path || !out || slice_ind < 0) return fail(TAMCMC_IO_ERR_ARG, "bad argument");
    ModelFile mf;
    int rc = read_model_file(model_path, slice_ind, mf);
    if (rc) return rc;
    tamcmc_inputs *in = new tamcmc_inputs();
    rc = build_local(mf, resol, *in);
    if (rc) { delete in; return rc; }
    *out = in;
    return TAMCMC_IO_OK;
}

int tamcmc_io_load_model_global(const char *model_path, double resol, tamcmc_inputs **out) {
    if (!model_path || !out) return fail(TAMCMC_IO_ERR_ARG, "bad argument");
    ModelFile mf;
    int rc = read_model_file(model_path, -1, mf);
    if (rc) return rc;
    tamcmc_inputs *in = new tamcmc_inputs();
    rc = build_global(mf, resol, *in);
    if (rc) { delete in; return rc; }
    *out = in;
    return TAMCMC_IO_OK;
}

int tamcmc_io_load_model_asymptotic(const char *model_path, double resol, tamcmc_inputs **out) {
    if (!model_path || !out) return fail(TAMCMC_IO_ERR_ARG, "bad argument");
    ModelFile mf;
    int rc = read_model_file(model_path, -1, mf);
    if (rc) return rc;
    tamcmc_inputs *in = new tamcmc_inputs();
    rc = build_asymptotic(mf, resol, *in);
    if (rc) { delete in; return rc; }
    *out = in;
    return TAMCMC_IO_OK;
}

void tamcmc_inputs_free(tamcmc_inputs *in) { delete in; }
int tamcmc_inputs_nparams(const tamcmc_inputs *in) { return in ? in->all.n : 0; }

int tamcmc_inputs_get(const tamcmc_inputs *in, double *params, int32_t *relax, double *priors, int32_t *priors_switch, int32_t *plength,
                      double *extra_priors, double *freq_range, int32_t *model_id, int32_t *prior_class, double *dnu, double *c_l) {
    if (!in) return fail(TAMCMC_IO_ERR_ARG, "null inputs");
    const int N = in->all.n;
    for (int i = 0; i < N; i++) {
        if (params) params[i] = in->all.inputs[(size_t)i];
        if (relax) relax[i] = in->all.relax[(size_t)i];
        if (priors_switch) priors_switch[i] = prior_id(in->all.prior_names[(size_t)i]);
    }
    if (priors) std::memcpy(priors, in->all.priors.data(), (size_t)4 * N * sizeof(double));
    if (plength) for (int k = 0; k < 11; k++) plength[k] = in->plength[k];
    if (extra_priors) for (int k = 0; k < 10; k++) extra_priors[k] = in->extra[k];
    if (freq_range) { freq_range[0] = in->range[0]; freq_range[1] = in->range[1]; }
    if (model_id) *model_id = in->model_id;
    if (prior_class) *prior_class = in->prior_class;
    if (dnu) *dnu = in->dnu;
    if (c_l) *c_l = in->c_l;
    return TAMCMC_IO_OK;
}
const char *tamcmc_inputs_name(const tamcmc_inputs *in, int i) { return (in && i >= 0 && i < in->all.n) ? in->all.names[(size_t)i].c_str() : ""; }
const char *tamcmc_inputs_prior_name(const tamcmc_inputs *in, int i) {
    return (in && i >= 0 && i < in->all.n) ? in->all.prior_names[(size_t)i].c_str() : "";
}
const char *tamcmc_inputs_model_name(const tamcmc_inputs *in) { return in ? in->model_name.c_str() : ""; }

}  // extern "C"
