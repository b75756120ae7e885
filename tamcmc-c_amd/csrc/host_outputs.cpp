// host_outputs.cpp -- the reference's on-disk sample formats (row N3 of SURVEY 8f), so that its post-processing tools
// (bin2txt, getstats, getevidence, IDL/Python readers) can consume GPU runs:
//   <root>params.hdr + <root>params_chain-<m>.bin   Outputs::write_bin_params          outputs.cpp:1231-1333
//   <root>stat_criteria.hdr + .bin                    Outputs::write_bin_stat_criteria  outputs.cpp:1472-1550
// and the summary statistics the reference's tools print per variable (tools/quick_samples_stats.cpp:4-35,
// used by tools/bin2txt_params.cpp:165-168): mean, median, population standard deviation;
//   <file>.txt evidence diagnostic                     Diagnostics::evidence_calc / write_evidence  diagnostics.cpp:980-1066
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/tamcmc_sampler.h"

namespace {
// One row of numbers the way the reference streams them: `stream << eigen_vector.transpose()` (Eigen's default IOFormat) prints
// every coefficient with the stream's precision (6 unless set), right-aligned to the width of the widest one, separated by " ".
std::string eigen_row(const double *v, int64_t n, int precision = 6) {
    std::vector<std::string> t((size_t)n);
    size_t w = 0;
    char buf[64];
    for (int64_t i = 0; i < n; i++) {
        std::snprintf(buf, sizeof buf, "%.*g", precision, v[i]);
        t[(size_t)i] = buf;
        w = std::max(w, t[(size_t)i].size());
    }
    std::string out;
    for (int64_t i = 0; i < n; i++) {
        if (i) out += " ";
        out.append(w - t[(size_t)i].size(), ' ');
        out += t[(size_t)i];
    }
    return out;
}
std::string eigen_row_int(const int32_t *v, int64_t n) {
    std::vector<double> d((size_t)n);
    for (int64_t i = 0; i < n; i++) d[(size_t)i] = (double)v[i];
    return eigen_row(d.data(), n, 17);
}
// rows already in a raw [sample][...] file of `row_bytes` per sample (0 when absent): the samples written by earlier buffers
int64_t rows_on_disk(const std::string &file, size_t row_bytes) {
    std::ifstream f(file.c_str(), std::ifstream::binary);
    if (!f.is_open() || row_bytes == 0) return 0;
    f.seekg(0, std::ios::end);
    return (int64_t)f.tellg() / (int64_t)row_bytes;
}
}  // namespace

extern "C" {

// Outputs::write_txt_acceptance (outputs.cpp:747-790): `strg << xaxis; strg << " "; strg << acceptance_rate.transpose() << "\n"`.
int tamcmc_outputs_write_acceptance(const char *file, double xaxis, const double *rates, int32_t Nchains, int32_t first) {
    if (!file || !rates || Nchains < 1) return TAMCMC_ERR_BAD_ARG;
    std::ofstream f;
    if (first) f.open(file);
    else f.open(file, std::ofstream::app);
    if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
    if (first) {
        f << "# This is an output file for the acceptance rate. \n";
        f << "# This file contains values for the acceptance_rate[0:Nchains-1] in function of the average sample position\n";
        f << "# Averaging is done over Nbuffer \n";
        f << "! Nchains= " << Nchains << "\n";
    }
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.6g", xaxis);  // a double through an ostream with the default precision
    f << buf << " " << eigen_row(rates, Nchains) << "\n";
    f.flush();
    return f.good() ? TAMCMC_OK : TAMCMC_ERR_BAD_ARG;
}

int tamcmc_outputs_read_acceptance(const char *file, int32_t *Nchains, int64_t max_rows, int64_t *n_rows, double *xaxis, double *rates) {
    if (!file || !Nchains || !n_rows) return TAMCMC_ERR_BAD_ARG;
    std::ifstream f(file);
    if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
    std::string ln;
    int nc = -1;
    int64_t n = 0;
    while (std::getline(f, ln)) {
        size_t p0 = ln.find_first_not_of(" \t\r");
        if (p0 == std::string::npos || ln[p0] == '#') continue;
        if (ln[p0] == '!') {
            const size_t eq = ln.find('=');
            if (eq == std::string::npos) return TAMCMC_ERR_BAD_ARG;
            nc = std::atoi(ln.c_str() + eq + 1);
            if (nc < 1) return TAMCMC_ERR_BAD_ARG;
            continue;
        }
        if (nc < 1) return TAMCMC_ERR_BAD_ARG;  // a data line before the header
        const char *q = ln.c_str() + p0;
        char *e = nullptr;
        const double x = std::strtod(q, &e);
        if (e == q) return TAMCMC_ERR_BAD_ARG;
        std::vector<double> r((size_t)nc);
        for (int m = 0; m < nc; m++) {
            q = e;
            r[(size_t)m] = std::strtod(q, &e);
            if (e == q) return TAMCMC_ERR_BAD_ARG;
        }
        if (n < max_rows) {
            if (xaxis) xaxis[n] = x;
            if (rates) std::memcpy(rates + (size_t)n * (size_t)nc, r.data(), (size_t)nc * sizeof(double));
        }
        n++;
    }
    if (nc < 1) return TAMCMC_ERR_BAD_ARG;
    *Nchains = nc;
    *n_rows = n;
    return TAMCMC_OK;
}

// Outputs::write_bin_params (outputs.cpp:1231-1333).  Like the reference, the ASCII header is rewritten with EVERY buffer and carries
// the cumulative sample count (outputs.cpp:1268: Nbuffer*Ncopy + counts + Nsamples_sofar) -- its tools read exactly Nsamples_done
// rows (getstats.cpp:156, getevidence.cpp:197); append != 0 = a later buffer of the same run (the .bin files grow).
int tamcmc_outputs_write_params(const char *root, const double *samples, int64_t n, int32_t Nchains, int32_t Nvars,
                                int64_t Nsamples_total, const int32_t *relax, const int32_t *plength, int32_t n_plength, int64_t Nparams,
                                const double *inputs, const char *const *names, int32_t append) {
    if (!root || !samples || n < 0 || Nchains < 1 || Nvars < 1 || !relax || !plength || n_plength < 1 || !inputs) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    const int64_t before = append ? rows_on_disk(base + "params_chain-0.bin", sizeof(double) * (size_t)Nvars) : 0;
    {
        std::ofstream h((base + "params.hdr").c_str());
        if (!h.is_open()) return TAMCMC_ERR_BAD_ARG;
        h << "# This is the header file of the BINARY output file for the model parameters \n";
        h << "# This file contains values for vars[0:Nchains-1][ 0:Nvars-1]. Each matrix is in a different file, indexed by the chain number\n";
        h << "! Nsamples= " << Nsamples_total << "\n";
        h << "! Nchains= " << Nchains << "\n";
        h << "! Nsamples_done=" << (before + n) << "\n";
        h << "! Nvars= " << Nvars << "\n";
        h << "! Ncons= " << (Nparams - Nvars) << "\n";
        h << "! relax= " << eigen_row_int(relax, Nparams) << "\n";
        h << "! plength= " << eigen_row_int(plength, n_plength) << "\n";
        h << "! constant_names= ";
        std::vector<double> cons;
        for (int64_t i = 0; i < Nparams; i++)
            if (relax[i] != 1) { h << (names ? names[i] : ("p" + std::to_string(i)).c_str()) << "   "; cons.push_back(inputs[i]); }
        if (cons.empty()) h << "None   ";
        h << "\n! constant_values= ";
        if (cons.empty()) h << "-1";
        else h << eigen_row(cons.data(), (int64_t)cons.size());  // the reference streams them at the default 6 significant digits
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

// Outputs::write_bin_stat_criteria (outputs.cpp:1472-1550); header rewritten per buffer with the cumulative count (:1502)
int tamcmc_outputs_write_stat_criteria(const char *root, const double *stats, int64_t n, int32_t Nchains, int32_t append) {
    if (!root || !stats || n < 0 || Nchains < 1) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    const int64_t before = append ? rows_on_disk(base + "stat_criteria.bin", sizeof(double) * 3 * (size_t)Nchains) : 0;
    {
        std::ofstream h((base + "stat_criteria.hdr").c_str());
        if (!h.is_open()) return TAMCMC_ERR_BAD_ARG;
        h << "# This is the header of the BINARY output file for the statistical information.\n";
        h << "# This file contains values for the logLikelihood (columns 0:Nchains-1), logPrior (columns Nchains:2*Nchains-1) and logPosterior (columns 2*Nchains:3*Nchains-1),  \n";
        h << "! Nsamples_done=" << (before + n) << "\n";
        h << "! Nchains= " << Nchains << "\n";
        h << "! labels= ";
        const char *labels[3] = {"logLikelihood", "logPrior", "logPosteriors"};  // (sic, outputs.cpp:1485)
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
    int64_t done = -1;
    std::string line;
    while (std::getline(h, line)) {
        if (line.rfind("! Nvars=", 0) == 0) nv = std::atoi(line.substr(8).c_str());
        if (line.rfind("! Nchains=", 0) == 0) nc = std::atoi(line.substr(10).c_str());
        if (line.rfind("! Nsamples_done=", 0) == 0) done = std::atoll(line.substr(16).c_str());
    }
    if (nv < 1 || nc < 1 || chain < 0 || chain >= nc) return TAMCMC_ERR_BAD_ARG;
    *Nvars = nv;
    if (Nchains) *Nchains = nc;
    std::ifstream f((base + "params_chain-" + std::to_string(chain) + ".bin").c_str(), std::ifstream::binary);
    if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
    f.seekg(0, std::ios::end);
    int64_t total = (int64_t)f.tellg() / (int64_t)(sizeof(double) * (size_t)nv);
    if (done >= 0 && done < total) total = done;  // the reference's tools read exactly Nsamples_done rows (getstats.cpp:156)
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

// ---- evidence diagnostic from the tempered ladder (Diagnostics::evidence_calc, diagnostics.cpp:980-1019) ----
// beta_m = 1/T_m, L_beta_m = mean over the samples of chain m's recorded log-likelihood; both ladders are resampled to
// interp_factor*Nchains points on the INDEX axis (quad_interpol, interpol.cpp:46-59: half-sample parabolas, linear in the first and
// last half interval) and the figure the reference calls the evidence is the plain average of the resampled L_beta.
namespace {
long double resample_at(const double *a, int n, long double x) {  // interp2 / interp1 / parabola, interpol.cpp:64-101
    if (x <= .5L || x >= (long double)n - 1.5L) {
        if (x <= 0) return a[0];
        if (x >= n - 1) return a[n - 1];
        const int j = (int)x;
        return a[j] + (x - j) * (a[j + 1] - a[j]);
    }
    const int j = (int)(x + .5L);
    const long double t = 2 * (x - j);
    const long double fm = ((long double)a[j - 1] + a[j]) / 2, f0 = a[j], fp = ((long double)a[j] + a[j + 1]) / 2;
    if (t <= -1) return fm;
    if (t >= 1) return fp;
    const long double l = f0 - t * (fm - f0), r = f0 + t * (fp - f0);
    return (l + r + t * (r - l)) / 2;
}
void resample(const double *a, int n, int m, double *out) {
    const long double step = (long double)((double)(n - 1) / (m - 1));
    for (int j = 0; j < m; j++) out[j] = (double)resample_at(a, n, j * step);
}
}  // namespace

// logL[i*row_stride + m*col_stride] = recorded log-likelihood of chain m at sample i (tamcmc_sampler_run's stats block:
// row_stride 3*Nchains, col_stride 3; the reference's stat_criteria rows: 3*Nchains, 1).  beta, L_beta: [Nchains];
// beta_interp, L_beta_interp: [interp_factor*Nchains] (either may be NULL).
int tamcmc_evidence_calc(const double *Tcoefs, int32_t Nchains, const double *logL, int64_t n, int64_t row_stride, int64_t col_stride,
                         int32_t interp_factor, double *beta, double *L_beta, double *beta_interp, double *L_beta_interp, double *evidence) {
    if (!Tcoefs || !logL || !beta || !L_beta || !evidence || Nchains < 1 || n < 1 || interp_factor < 1 || row_stride < 1 || col_stride < 1)
        return TAMCMC_ERR_BAD_ARG;
    const int npts = interp_factor * Nchains;
    if (npts < 2) return TAMCMC_ERR_BAD_ARG;  // the resampling step divides by npts - 1
    for (int32_t m = 0; m < Nchains; m++) {
        beta[m] = 1. / Tcoefs[m];
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += logL[(size_t)i * (size_t)row_stride + (size_t)m * (size_t)col_stride];
        L_beta[m] = s / (double)n;
    }
    std::vector<double> bi((size_t)npts), li((size_t)npts);
    resample(beta, Nchains, npts, bi.data());
    resample(L_beta, Nchains, npts, li.data());
    double tot = 0;
    for (int j = 0; j < npts; j++) tot += li[(size_t)j];
    *evidence = tot / npts;
    if (beta_interp) std::copy(bi.begin(), bi.end(), beta_interp);
    if (L_beta_interp) std::copy(li.begin(), li.end(), L_beta_interp);
    return TAMCMC_OK;
}

// One line of <file> per call: sample count, L_beta[0:Nchains], evidence; header on the first call (write_evidence, diagnostics.cpp:1021-1066)
int tamcmc_outputs_write_evidence(const char *file, int64_t n_samples, int32_t Nchains, const double *beta, const double *L_beta,
                                  int32_t interp_factor, double evidence, int32_t first) {
    if (!file || !beta || !L_beta || Nchains < 1) return TAMCMC_ERR_BAD_ARG;
    FILE *f = std::fopen(file, first ? "w" : "a");
    if (!f) return TAMCMC_ERR_BAD_ARG;
    if (first) {
        std::fprintf(f, "# This is an output file for the evidence. Evidence is calculated after a quadratic interpolation of L_beta. \n");
        std::fprintf(f, "# This file contains values for the L_beta[0:Nchains-1], the evidence calculated at each time the buffer was written \n");
        std::fprintf(f, "# col(1): Number of samples used to compute the evidence \n");
        std::fprintf(f, "# col(2:2+Nchains): averaged probability <P(D|M,I)> over the samples of each chain \n");
        std::fprintf(f, "# col(2+Nchains+1): Evidence P(M|D, I) computed by (1) interpolation and (2) averaging \n");
        std::fprintf(f, "! beta=%s\n", eigen_row(beta, Nchains).c_str());
        std::fprintf(f, "! interpolation_factor=%d\n", (int)interp_factor);
    }
    // the reference streams an Eigen row behind std::setw(20) << std::setprecision(10) (diagnostics.cpp:1048-1052): the pending
    // width pads Eigen's empty matrix prefix to 20 blanks and every coefficient to 20 columns; Eigen then RESTORES the width, so the
    // " " that follows is padded to 20 columns as well, and the evidence takes its own setw(20)
    std::fprintf(f, "%lld %20s", (long long)n_samples, "");
    for (int32_t m = 0; m < Nchains; m++) std::fprintf(f, "%s%20.10g", m ? " " : "", L_beta[m]);
    std::fprintf(f, "%20s%20.10g\n", "", evidence);
    return std::fclose(f) == 0 ? TAMCMC_OK : TAMCMC_ERR_BAD_ARG;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Restore files (checkpoint / resume): <root>1.dat = chain positions, <root>2.dat = sigmas and mus, <root>3.dat = covariance
// matrices; layout of Outputs::write_buffer_restore (tamcmc/sources/outputs.cpp:863-1025), read back as
// Config::read_restore_files does (config.cpp:1734-1990): '#' comment lines, "! key= value" lines, matrix rows, "*<chain>"
// block markers.  Two deliberate differences: numbers are written with 17 significant digits (the reference streams Eigen
// rows at the default 6, so its restart is lossy), and the *_mean blocks (averages over the reference's output buffer, which
// this build does not keep) repeat the last values.
namespace {
// comment block + size keys of one restore file, line for line as Outputs::write_buffer_restore writes them (outputs.cpp:874-894,
// :926-946, :981-1001); `dr` = the wording of the three "Use this" lines, which differs between file 1 and files 2/3
void restore_header(std::ofstream &f, int file_no, const char *what, int Nchains, int Nvars, int64_t iteration, const char *const *names) {
    const char *dr = file_no == 1 ? "do_restore_[X]=1" : "do_restore=1";
    const char *dp = file_no == 1 ? "do_restore_proposal=1" : "do_restore=1";
    f << "# This is an output file containing what is required to restore a run to its last saved position \n";
    f << "# File number: " << file_no << " \n";
    f << "# Contains " << what << "\n";
    f << "# Use this if you wish to: \n";
    f << "#       (1) complete a finished job that requires more samples ==> set erase_old_file=0 and " << dr << " \n";
    f << "#       (2) restart a finished job by ignoring old samples (e.g. ignoring a Burn-in) ==> set erase_old_file=1 and " << dp << " \n";
    f << "#       (3) terminate an unfinished job which failed to finished (e.g. due to computer unexpected shutdown) ==> set erase_old_file=0 and " << dr << " \n";
    f << "! Nchains= " << Nchains << "\n";
    f << "! Nvars= " << Nvars << "\n";
    f << "! iteration=" << iteration << "\n";
    f << "! variable_names=";
    for (int i = 0; i < Nvars; i++) f << (names && names[i] ? names[i] : "var") << "   ";
    f << "\n";
}
// one matrix row, aligned like an Eigen row but with 17 significant digits (lossless restart; the reference: 6)
void put_row(std::ofstream &f, const double *v, int n) { f << eigen_row(v, n, 17) << "\n"; }
// numbers following `key` in a restore file: on the key's own line after '=', then on the following lines until the next
// line starting with '!' or '#'; lines starting with '*' (chain markers) are skipped
bool read_block(const std::string &path, const std::string &key, size_t want, std::vector<double> &out, int *Nchains, int *Nvars,
                int64_t *iteration) {
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    out.clear();
    bool in_block = false;
    for (std::string ln; std::getline(f, ln);) {
        size_t a = ln.find_first_not_of(" \t\r");
        if (a == std::string::npos) continue;
        if (ln[a] == '#') { in_block = false; continue; }
        if (ln[a] == '!') {
            const size_t eq = ln.find('=');
            std::string k = ln.substr(a, eq == std::string::npos ? std::string::npos : eq - a);
            while (!k.empty() && (k.back() == ' ' || k.back() == '\t')) k.pop_back();
            const std::string rest = eq == std::string::npos ? "" : ln.substr(eq + 1);
            if (k == "! Nchains" && Nchains) *Nchains = std::atoi(rest.c_str());
            if (k == "! Nvars" && Nvars) *Nvars = std::atoi(rest.c_str());
            if (k == "! iteration" && iteration) *iteration = std::atoll(rest.c_str());
            in_block = (k == "! " + key);
            if (in_block) {
                std::istringstream is(rest);
                for (double v; is >> v;) out.push_back(v);
            }
            continue;
        }
        if (!in_block || ln[a] == '*') continue;
        std::istringstream is(ln);
        for (double v; is >> v;) out.push_back(v);
    }
    return out.size() == want;
}
}  // namespace

extern "C" int tamcmc_outputs_write_restore(const char *root, int32_t Nchains, int32_t Nvars, int64_t iteration, const char *const *names,
                                            const double *vars, const double *sigmas, const double *mus, const double *covarmats) {
    if (!root || !vars || !sigmas || !mus || !covarmats || Nchains < 1 || Nvars < 1) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    {
        std::ofstream f((base + "1.dat").c_str());
        if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
        restore_header(f, 1, "the last values for the variables vars[0:Nchain-1]. vars_mean denotes averaged values of Nbuffer ", Nchains, Nvars,
                       iteration, names);
        for (const char *key : {"! vars= ", "! vars_mean= "}) {
            f << key << "\n";
            for (int m = 0; m < Nchains; m++) put_row(f, vars + (size_t)m * Nvars, Nvars);
        }
    }
    {
        std::ofstream f((base + "2.dat").c_str());
        if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
        restore_header(f, 2, "the last values of (a) sigmas[0:Nchains-1] and (b) mus[0:Nchains-1, 0:Nvars-1].  sigmas_mean and mus_mean denotes averaged values of Nbuffer",
                       Nchains, Nvars, iteration, names);
        for (int pass = 0; pass < 2; pass++) {
            f << (pass ? "! sigmas_mean= " : "! sigmas= ");
            put_row(f, sigmas, Nchains);
            f << (pass ? "! mus_mean= " : "! mus= ") << "\n";
            for (int m = 0; m < Nchains; m++) put_row(f, mus + (size_t)m * Nvars, Nvars);
        }
    }
    {
        std::ofstream f((base + "3.dat").c_str());
        if (!f.is_open()) return TAMCMC_ERR_BAD_ARG;
        restore_header(f, 3, "the last value of the covariance matrix covarmats[0:Nchains-1, 0:Nvars-1, 0:Nvars-1]. covarmats_mean denotes the averaged values over Nbuffer",
                       Nchains, Nvars, iteration, names);
        for (const char *key : {"! covarmats= ", "! covarmats_mean= "}) {
            f << key << "\n";
            for (int m = 0; m < Nchains; m++) {
                f << "*" << m << "\n";
                for (int i = 0; i < Nvars; i++) put_row(f, covarmats + ((size_t)m * Nvars + i) * Nvars, Nvars);
            }
        }
    }
    return TAMCMC_OK;
}

// Sizes first (arrays may be NULL), then the arrays: vars [Nchains x Nvars], sigmas [Nchains], mus [Nchains x Nvars],
// covarmats [Nchains x Nvars x Nvars].
extern "C" int tamcmc_outputs_read_restore(const char *root, int32_t *Nchains, int32_t *Nvars, int64_t *iteration, double *vars, double *sigmas,
                                           double *mus, double *covarmats) {
    if (!root || !Nchains || !Nvars) return TAMCMC_ERR_BAD_ARG;
    const std::string base(root);
    int nc = 0, nv = 0;
    int64_t it = 0;
    std::vector<double> tmp;
    read_block(base + "1.dat", "\x01none", 0, tmp, &nc, &nv, &it);  // header scan
    if (nc < 1 || nv < 1) return TAMCMC_ERR_BAD_ARG;
    *Nchains = nc; *Nvars = nv;
    if (iteration) *iteration = it;
    const size_t C = (size_t)nc, V = (size_t)nv;
    if (vars) {
        if (!read_block(base + "1.dat", "vars", C * V, tmp, nullptr, nullptr, nullptr)) return TAMCMC_ERR_BAD_ARG;
        std::copy(tmp.begin(), tmp.end(), vars);
    }
    if (sigmas) {
        if (!read_block(base + "2.dat", "sigmas", C, tmp, nullptr, nullptr, nullptr)) return TAMCMC_ERR_BAD_ARG;
        std::copy(tmp.begin(), tmp.end(), sigmas);
    }
    if (mus) {
        if (!read_block(base + "2.dat", "mus", C * V, tmp, nullptr, nullptr, nullptr)) return TAMCMC_ERR_BAD_ARG;
        std::copy(tmp.begin(), tmp.end(), mus);
    }
    if (covarmats) {
        if (!read_block(base + "3.dat", "covarmats", C * V * V, tmp, nullptr, nullptr, nullptr)) return TAMCMC_ERR_BAD_ARG;
        std::copy(tmp.begin(), tmp.end(), covarmats);
    }
    return TAMCMC_OK;
}
