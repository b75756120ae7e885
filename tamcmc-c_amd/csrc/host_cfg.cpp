// host_cfg.cpp -- the `.cfg` files of the kept surface (include/tamcmc_io.h): the group/key=value; comment dialect of
// Config/default/config_default.cfg (Config::format_line + read_cfg_file, tamcmc/sources/config.cpp:1062-1112, :1223-1732)
// and the three-column Config/default/errors_default.cfg (Config::read_defautlerrors :2096-2150, used by
// MALA::init_proposal, MALA.cpp:246-262).  Plain C++ (no device code).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/tamcmc_io.h"
#include "../../include/tamcmc_sampler.h"

namespace {
thread_local std::string g_cfg_err;
int cfail(int code, const std::string &m) { g_cfg_err = m; return code; }
std::string trim(const std::string &s) {
    const char *ws = " \t\r\n";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
}
// leading number of a string ("3.50 #1.70" -> 3.5), as the reference's stringstream conversions read it
bool lead_number(const std::string &s, double *v) {
    std::istringstream is(s);
    long double t = 0;
    if (!(is >> t)) return false;
    *v = (double)t;
    return true;
}
}  // namespace

struct tamcmc_cfg {
    std::map<std::string, std::map<std::string, std::string>> groups;  // group (without '!' and ':') -> key -> raw value
};

extern "C" {

const char *tamcmc_cfg_last_error(void) { return g_cfg_err.c_str(); }

int tamcmc_cfg_open(const char *path, tamcmc_cfg **out) {
    if (!path || !out) return cfail(TAMCMC_IO_ERR_ARG, "null argument");
    std::ifstream f(path);
    if (!f.is_open()) return cfail(TAMCMC_IO_ERR_OPEN, std::string("cannot open ") + path);
    tamcmc_cfg *c = new tamcmc_cfg();
    std::string group;
    for (std::string raw; std::getline(f, raw);) {
        const std::string ln = trim(raw);
        if (ln.empty() || ln[0] == '#') continue;
        if (ln == "/END" || ln == "/END;") break;
        if (ln[0] == '!') {  // group indicator up to ':' (format_line, config.cpp:1098-1108)
            const size_t p = ln.find(':');
            if (p == std::string::npos) { delete c; return cfail(TAMCMC_IO_ERR_SYNTAX, "group line without ':': " + ln); }
            group = trim(ln.substr(1, p - 1));
            continue;
        }
        const size_t semi = ln.find(';');  // the value ends at the first ';' -- the rest is a comment (config.cpp:1087-1094)
        if (semi == std::string::npos) { delete c; return cfail(TAMCMC_IO_ERR_SYNTAX, "line without the ';' terminator: " + ln); }
        const std::string body = trim(ln.substr(0, semi));
        const size_t eq = body.find('=');
        if (eq == std::string::npos) { delete c; return cfail(TAMCMC_IO_ERR_SYNTAX, "line without '=': " + ln); }
        c->groups[group][trim(body.substr(0, eq))] = trim(body.substr(eq + 1));
    }
    *out = c;
    return TAMCMC_IO_OK;
}
void tamcmc_cfg_free(tamcmc_cfg *c) { delete c; }

int tamcmc_cfg_string(const tamcmc_cfg *c, const char *group, const char *key, char *buf, int n) {
    if (!c || !group || !key || !buf || n < 1) return cfail(TAMCMC_IO_ERR_ARG, "bad argument");
    const auto g = c->groups.find(group);
    if (g == c->groups.end()) return cfail(TAMCMC_IO_ERR_SYNTAX, std::string("no group ") + group);
    const auto k = g->second.find(key);
    if (k == g->second.end()) return cfail(TAMCMC_IO_ERR_SYNTAX, std::string("no key ") + key);
    std::snprintf(buf, (size_t)n, "%s", k->second.c_str());
    return TAMCMC_IO_OK;
}

int tamcmc_cfg_numbers(const tamcmc_cfg *c, const char *group, const char *key, double *out, int max, int *n) {
    char buf[512];
    int rc = tamcmc_cfg_string(c, group, key, buf, sizeof buf);
    if (rc) return rc;
    if (!out || !n || max < 1) return cfail(TAMCMC_IO_ERR_ARG, "bad argument");
    int cnt = 0;
    std::string s(buf);
    size_t i = 0;
    while (i <= s.size() && cnt < max) {  // comma- or blank-separated list (str_to_arrint(word, " ,"), config.cpp:1291-1300)
        size_t j = s.find_first_of(", \t", i);
        if (j == std::string::npos) j = s.size();
        if (j > i) {
            double v;
            if (!lead_number(s.substr(i, j - i), &v)) break;  // a trailing comment without ';'
            out[cnt++] = v;
        }
        i = j + 1;
    }
    *n = cnt;
    return cnt > 0 ? TAMCMC_IO_OK : cfail(TAMCMC_IO_ERR_SYNTAX, std::string("no number in ") + key);
}

// The numeric fields of tamcmc_sampler_config from the !MALA / !Modeling groups (pointer fields other than Nt_learn and
// periods_learn are left untouched); Nsamples / Nbuffer from !Outputs.
int tamcmc_cfg_sampler(const tamcmc_cfg *c, tamcmc_sampler_config *out, int64_t *Nt_learn, int64_t *periods_learn, int max_learn,
                       int64_t *Nsamples, int64_t *Nbuffer) {
    if (!c || !out || !Nt_learn || !periods_learn || max_learn < 2) return cfail(TAMCMC_IO_ERR_ARG, "bad argument");
    double v[16];
    int n = 0;
    struct { const char *key; double *dst; } dbl[] = {
        {"target_acceptance", &out->target_acceptance}, {"c0", &out->c0}, {"epsilon1", &out->epsilon1}, {"epsilon2", &out->epsilon2},
        {"A1", &out->A1}, {"delta", &out->delta}, {"delta_x", &out->delta_x}, {"lambda_temp", &out->lambda_temp}};
    for (auto &e : dbl) {
        int rc = tamcmc_cfg_numbers(c, "MALA", e.key, v, 1, &n);
        if (rc) return rc;
        *e.dst = v[0];
    }
    int rc = tamcmc_cfg_numbers(c, "MALA", "use_drift", v, 1, &n);
    if (rc) return rc;
    out->use_drift = (int32_t)v[0];
    if ((rc = tamcmc_cfg_numbers(c, "MALA", "Nchains", v, 1, &n))) return rc;
    out->Nchains = (int32_t)v[0];
    if ((rc = tamcmc_cfg_numbers(c, "MALA", "dN_mixing", v, 1, &n))) return rc;
    out->dN_mixing = (int64_t)v[0];
    if ((rc = tamcmc_cfg_numbers(c, "MALA", "Nt_learn", v, max_learn < 16 ? max_learn : 16, &n))) return rc;
    for (int i = 0; i < n; i++) Nt_learn[i] = (int64_t)v[i];
    const int nl = n;
    if ((rc = tamcmc_cfg_numbers(c, "MALA", "periods_learn", v, max_learn < 16 ? max_learn : 16, &n))) return rc;
    if (n != nl - 1) return cfail(TAMCMC_IO_ERR_SYNTAX, "periods_learn must have one entry less than Nt_learn (config_default.cfg:18)");
    for (int i = 0; i < n; i++) periods_learn[i] = (int64_t)v[i];
    out->Nt_learn = Nt_learn;
    out->periods_learn = periods_learn;
    out->n_Nt_learn = nl;
    if ((rc = tamcmc_cfg_numbers(c, "Modeling", "likelihood_params", v, 1, &n))) return rc;
    out->likelihood_params = v[0];
    char name[128];
    if ((rc = tamcmc_cfg_string(c, "Modeling", "likelihood_fct_name", name, sizeof name))) return rc;
    if (std::strcmp(name, "chi(2,2p)") == 0) out->likelihood_id = 0;       // Config/default/likelihoods_ctrl.list
    else if (std::strcmp(name, "chi_square") == 0) out->likelihood_id = 1;
    else return cfail(TAMCMC_IO_ERR_UNSUPPORTED, std::string("unknown likelihood ") + name);
    if ((rc = tamcmc_cfg_string(c, "Modeling", "prior_fct_name", name, sizeof name))) return rc;
    static const char *classes[] = {"priors_Kallinger2014_Gaussian", "priors_Harvey_Gaussian", "io_MS_Global", "io_local", "io_asymptotic",
                                    "io_ajfit"};  // Config/default/priors_ctrl.list
    out->prior_class = -1;
    for (int i = 0; i < 6; i++)
        if (std::strcmp(name, classes[i]) == 0) out->prior_class = i;
    if (out->prior_class < 0) return cfail(TAMCMC_IO_ERR_UNSUPPORTED, std::string("unknown prior class ") + name);
    if (Nsamples) { if ((rc = tamcmc_cfg_numbers(c, "Outputs", "Nsamples", v, 1, &n))) return rc; *Nsamples = (int64_t)v[0]; }
    if (Nbuffer) { if ((rc = tamcmc_cfg_numbers(c, "Outputs", "Nbuffer", v, 1, &n))) return rc; *Nbuffer = (int64_t)v[0]; }
    return TAMCMC_IO_OK;
}

// err = A * value + B with (A, B) of the row whose name equals the parameter's name, 1 when there is none
// (MALA::init_proposal, MALA.cpp:252-262); names/values: the FREE parameters in order.
int tamcmc_io_init_errors(const char *errors_path, const char *const *names, const double *values, int64_t nvars, double *errors) {
    if (!errors_path || !names || !values || !errors || nvars < 0) return cfail(TAMCMC_IO_ERR_ARG, "bad argument");
    std::ifstream f(errors_path);
    if (!f.is_open()) return cfail(TAMCMC_IO_ERR_OPEN, std::string("cannot open ") + errors_path);
    std::vector<std::string> nm;
    std::vector<double> A, B;
    for (std::string raw; std::getline(f, raw);) {
        const std::string ln = trim(raw);
        if (ln.empty() || ln[0] == '#') continue;
        std::istringstream is(ln);
        std::string n;
        double a = 0, b = 0;
        if (!(is >> n >> a >> b)) return cfail(TAMCMC_IO_ERR_SYNTAX, "errors file row needs: name A B: " + ln);
        nm.push_back(n); A.push_back(a); B.push_back(b);
    }
    for (int64_t i = 0; i < nvars; i++) {
        errors[i] = 1.0;
        for (size_t j = 0; j < nm.size(); j++)
            if (nm[j] == names[i]) errors[i] = values[i] * A[j] + B[j];  // the LAST matching row wins, as in the reference's loop
    }
    return TAMCMC_IO_OK;
}

}  // extern "C"
