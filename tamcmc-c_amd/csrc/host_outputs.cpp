// host_outputs.cpp -- the reference's on-disk sample formats (row N3 of SURVEY 8f), so that its post-processing tools
// (bin2txt, getstats, getevidence, IDL/Python readers) can consume GPU runs:
//   <root>params.hdr + <root>params_chain-<m>.bin   Outputs::write_bin_params          outputs.cpp:1231-1333
//   <root>stat_criteria.hdr + .bin                    Outputs::write_bin_stat_criteria  outputs.cpp:1472-1550
// and the summary statistics the reference's tools print per variable (tools/quick_samples_stats.cpp:4-35,
// used by tools/bin2txt_params.cpp:165-168): mean, median, population standard deviation.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/tamcmc_sampler.h"

extern "C" {

int tamcmc_outputs_write_params(const char *root, const double *samples, int64_t n, int32_t Nchains, int32_t Nvars,
                                int64_t Nsamples_total, const int32_t *relax, const int32_t *plength, int64_t Nparams,
                                const double *inputs, const char *const *names, int32_t append) {
    if (!root || !samples || n < 0 || Nchains < 1 || Nvars < 1 || !relax || !plength || !inputs) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    if (!append) {
        std::ofstream h((base + "params.hdr").c_str());
        if (!h.is_open()) return TAMCMC_ERR_BAD_ARG;
        h << "# This is the header file of the BINARY output file for the model parameters \n";
        h << "# This file contains values for vars[0:Nchains-1][ 0:Nvars-1]. Each matrix is in a different file, indexed by the chain number\n";
        h << "! Nsamples= " << Nsamples_total << "\n";
        h << "! Nchains= " << Nchains << "\n";
        h << "! Nsamples_done=" << n << "\n";
        h << "! Nvars= " << Nvars << "\n";
        h << "! Ncons= " << (Nparams - Nvars) << "\n";
        h << "! relax= ";
        for (int64_t i = 0; i < Nparams; i++) h << relax[i] << (i + 1 < Nparams ? " " : "");
        h << "\n! plength= ";
        for (int i = 0; i < 11; i++) h << plength[i] << (i < 10 ? " " : "");
        h << "\n! constant_names= ";
        bool any_cons = false;
        for (int64_t i = 0; i < Nparams; i++)
            if (relax[i] != 1) { h << (names ? names[i] : ("p" + std::to_string(i)).c_str()) << "   "; any_cons = true; }
        if (!any_cons) h << "None   ";
        h << "\n! constant_values= ";
        if (!any_cons) h << "-1";
        else {
            h.precision(12);
            bool first = true;
            for (int64_t i = 0; i < Nparams; i++)
                if (relax[i] != 1) { h << (first ? "" : " ") << inputs[i]; first = false; }
        }
        h << "\n! variable_names=";
        for (int64_t i = 0; i < Nparams; i++)
            if (relax[i] == 1) h << (names ? names[i] : ("p" + std::to_string(i)).c_str()) << "   ";
        h << "\n";
    }
    for (int32_t m = 0; m < Nchains; m++) {
        const std::string fn = base + "params_chain-" + std::to_string(m) + ".bin";
        std::ofstream f(fn.c_str(), append ? (std::ofstream::app | std::ofstream::binary) : std::ofstream::binary);
        if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
        for (int64_t i = 0; i < n; i++)  // raw little-endian doubles, [sample][var]
            f.write(reinterpret_cast<const char *>(samples + ((size_t)i * Nchains + (size_t)m) * Nvars), (std::streamsize)(sizeof(double) * (size_t)Nvars));
        f.flush();
    }
    return TAMCMC_OK;
}

int tamcmc_outputs_write_stat_criteria(const char *root, const double *stats, int64_t n, int32_t Nchains, int32_t append) {
    if (!root || !stats || n < 0 || Nchains < 1) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    if (!append) {
        std::ofstream h((base + "stat_criteria.hdr").c_str());
        if (!h.is_open()) return TAMCMC_ERR_BAD_ARG;
        h << "# This is the header of the BINARY output file for the statistical information.\n";
        h << "# This file contains values for the logLikelihood (columns 0:Nchains-1), logPrior (columns Nchains:2*Nchains-1) and logPosterior (columns 2*Nchains:3*Nchains-1),  \n";
        h << "! Nsamples_done=" << n << "\n";
        h << "! Nchains= " << Nchains << "\n";
        h << "! labels= ";
        const char *labels[3] = {"logLikelihood", "logPrior", "logPosterior"};
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < Nchains; i++) h << labels[k] << "[" << i << "]   ";
        h << "\n";
    }
    std::ofstream f((base + "stat_criteria.bin").c_str(), append ? (std::ofstream::app | std::ofstream::binary) : std::ofstream::binary);
    if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)       // per sample: all chains' logL, then logPrior, then logPosterior
            for (int32_t m = 0; m < Nchains; m++)
                f.write(reinterpret_cast<const char *>(stats + ((size_t)i * Nchains + (size_t)m) * 3 + k), sizeof(double));
    return TAMCMC_OK;
}

// Reads <root>params.hdr (Nchains, Nvars) and one chain's .bin (tools/bin2txt_params.cpp:95-140).
int tamcmc_outputs_read_params(const char *root, int32_t chain, double *samples, int64_t max_samples, int64_t *n_read,
                               int32_t *Nchains, int32_t *Nvars) {
    if (!root || !n_read || !Nvars) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    std::ifstream h((base + "params.hdr").c_str());
    if (!h.is_open()) return TAMCMC_ERR_BAD_ARG;
    int nv = -1, nc = -1;
    std::string line;
    while (std::getline(h, line)) {
        if (line.rfind("! Nvars=", 0) == 0) nv = std::atoi(line.substr(8).c_str());
        if (line.rfind("! Nchains=", 0) == 0) nc = std::atoi(line.substr(10).c_str());
    }
    if (nv < 1 || nc < 1 || chain < 0 || chain >= nc) return TAMCMC_ERR_BAD_ARG;
    *Nvars = nv;
    if (Nchains) *Nchains = nc;
    std::ifstream f((base + "params_chain-" + std::to_string(chain) + ".bin").c_str(), std::ifstream::binary);
    if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
    f.seekg(0, std::ios::end);
    const int64_t total = (int64_t)f.tellg() / (int64_t)(sizeof(double) * (size_t)nv);
    f.seekg(0, std::ios::beg);
    const int64_t n = samples ? std::min(total, max_samples) : 0;
    if (n > 0) f.read(reinterpret_cast<char *>(samples), (std::streamsize)(sizeof(double) * (size_t)nv * (size_t)n));
    *n_read = samples ? n : total;
    return TAMCMC_OK;
}

// mean / median / population standard deviation per variable (tools/quick_samples_stats.cpp:4-35)
int tamcmc_params_summary(const double *samples, int64_t n, int32_t Nvars, int64_t row_stride, double *mean, double *median,
                          double *stddev) {
    if (!samples || n < 1 || Nvars < 1 || row_stride < Nvars || !mean || !median || !stddev) return TAMCMC_ERR_BAD_ARG;
    std::vector<double> col((size_t)n);
    for (int32_t v = 0; v < Nvars; v++) {
        double s = 0;
        for (int64_t i = 0; i < n; i++) { col[(size_t)i] = samples[(size_t)i * (size_t)row_stride + (size_t)v]; s += col[(size_t)i]; }
        const double mu = s / (double)n;
        double q = 0;
        for (int64_t i = 0; i < n; i++) q = q + (col[(size_t)i] - mu) * (col[(size_t)i] - mu);
        std::sort(col.begin(), col.end());
        mean[v] = mu;
        median[v] = (n % 2 != 0) ? col[(size_t)(n / 2)] : (col[(size_t)((n - 1) / 2)] + col[(size_t)(n / 2)]) / 2.0;
        stddev[v] = std::sqrt(q / (double)n);
    }
    return TAMCMC_OK;
}

}  // extern "C"
