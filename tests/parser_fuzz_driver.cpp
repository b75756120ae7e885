// Sanitizer driver of the input front end (CPU build, -fsanitize=address,undefined): parses every file given on the command
// line with every loader, then every PREFIX of it (cut after each line) and a set of single-line deletions.  Malformed input
// must come back as an error code -- never a crash, an out-of-bounds access or a leak of the half-built inputs.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../include/tamcmc_io.h"

static int try_all(const std::string &path) {
    int ok = 0;
    tamcmc_inputs *in = nullptr;
    for (int slice = 0; slice < 3; slice++)
        if (tamcmc_io_load_model_local(path.c_str(), slice, 0.01, &in) == TAMCMC_IO_OK) { ok++; tamcmc_inputs_free(in); }
    if (tamcmc_io_load_model_global(path.c_str(), 0.01, &in) == TAMCMC_IO_OK) { ok++; tamcmc_inputs_free(in); }
    if (tamcmc_io_load_model_asymptotic(path.c_str(), 0.01, &in) == TAMCMC_IO_OK) { ok++; tamcmc_inputs_free(in); }
    double *tab = nullptr;
    int64_t nr = 0, nc = 0;
    if (tamcmc_io_read_data(path.c_str(), &tab, &nr, &nc) == TAMCMC_IO_OK) {
        int64_t a, b;
        if (nc > 0) tamcmc_io_select_range(tab, nr, nc, 0, 0.0, 1e9, &a, &b);
        ok++;
        tamcmc_io_free(tab);
    }
    tamcmc_cfg *cfg = nullptr;
    if (tamcmc_cfg_open(path.c_str(), &cfg) == TAMCMC_IO_OK) {
        char buf[64];
        double v[8];
        int n = 0;
        tamcmc_cfg_string(cfg, "MALA", "proposal_type", buf, sizeof buf);
        tamcmc_cfg_numbers(cfg, "MALA", "Nt_learn", v, 8, &n);
        ok++;
        tamcmc_cfg_free(cfg);
    }
    return ok;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string scratch = argv[1];
    long parsed = 0, variants = 0;
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a]);
        std::vector<std::string> lines;
        for (std::string ln; std::getline(f, ln);) lines.push_back(ln);
        parsed += try_all(argv[a]);
        auto write = [&](const std::vector<std::string> &ls) {
            std::ofstream o(scratch.c_str());
            for (const auto &l : ls) o << l << "\n";
        };
        for (size_t cut = 0; cut <= lines.size(); cut += (lines.size() > 400 ? 37 : 1)) {  // every prefix (long data files: sampled)
            write(std::vector<std::string>(lines.begin(), lines.begin() + (long)cut));
            parsed += try_all(scratch);
            variants++;
        }
        for (size_t del = 0; del < lines.size() && del < 200; del++) {  // one line removed
            std::vector<std::string> ls(lines);
            ls.erase(ls.begin() + (long)del);
            write(ls);
            parsed += try_all(scratch);
            variants++;
        }
        for (size_t k = 0; k < lines.size() && k < 200; k++) {  // one line replaced by junk of several shapes
            for (const char *junk : {"", "#", "!", "* 1", "p", "p 9 x y z", "1 2", "name", "name prior", "a=b", "!G:", "1e999 nan -inf"}) {
                std::vector<std::string> ls(lines);
                ls[k] = junk;
                write(ls);
                parsed += try_all(scratch);
                variants++;
            }
        }
    }
    std::printf("variants %ld, successful parses %ld\n", variants, parsed);
    return 0;
}
