// rgb_unpack.h -- red-giant models (ids 25 / 27): the solver's and the row builder's per-vector inputs (Prep, RowIn) and the scalar
// unpack that fills them from a parameter vector, written once for the host (batched C-ABI path: one thread per vector) and the device
// (the device-resident sampler runs it inside its proposal kernel, one wave per chain).  Kernels: rgb_prestep.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/tamcmc_hip.h"
#include "mode_tables_impl.h"

namespace tamcmc {
namespace rgb {

constexpr int MAXP = 32;      // p modes per vector (fmax-fmin+2 Dnu)/Dnu + margins
constexpr int MAXSOL = 1024;  // mixed modes per vector before de-duplication

struct Prep {  // one parameter vector's solver inputs
    int Lp, Lg, ng_min, status;
    int probe_dense, pad_;         // TAMCMC_OPT_ARMM_DENSE_SCAN: walk the whole grid (the reference's way) instead of the pole-structured scan
    int ig0[MAXP];                 // first g mode inside the zone of p mode ip, -1: none (the reference skips the pair)
    double nu_p[MAXP], dnu_loc[MAXP], dnup[MAXP];
    double Dnu_p, DPl, alpha, q, zone, resol, fact, keep_lo, keep_hi;
};

constexpr int MAXL = 32;       // modes per degree listed in the parameter vector
constexpr int MAXNODE = 16;    // nodes of the bias spline
constexpr int CAP1 = 400;      // mixed modes per vector that get a table row

struct RowIn {  // everything the row builder needs besides the solver's output (host-filled, one per vector)
    int Nfl0, Nfl2, Nfl3, lmax, do_amp, bias_n, status, cte_width;  // cte_width: id 27, every width is g[0]
    double fl0[MAXL], Wl0[MAXL], Hl0[MAXL], fl2[MAXL], fl3[MAXL];
    double g[6], Vl[4], V[4][7];
    double eta0, asym, trunc_c, Hfactor, Wfactor, rot_env, rot_core, a2, a3, a4, a5, a6, fmin, fmax;
    double sx[MAXNODE], sy[MAXNODE], sb[MAXNODE], sc[MAXNODE], sd[MAXNODE], sc0;  // spline coefficients (host-computed), bias_n nodes
};

__host__ __device__ inline double nu_g_of(const Prep &p, int ig) { return 1e6 / (((double)(p.ng_min + ig) + p.alpha) * p.DPl); }

// ---------------------------------------------------------------- scalar unpack of one parameter vector (host AND device)
// natural cubic (type 1) or cubic Hermite (type 2) spline through the bias nodes, spline.h:242-498: coefficients into the RowIn
__host__ __device__ inline bool spline_set(RowIn &ri, const double *xn, const double *yn, int n, int type) {
    if (n < 3 || n > MAXNODE) return false;
    for (int i = 0; i < n - 1; i++)
        if (!(xn[i] < xn[i + 1])) return false;
    double *x = ri.sx, *y = ri.sy, *b = ri.sb, *c = ri.sc, *d = ri.sd;
    for (int i = 0; i < n; i++) { x[i] = xn[i]; y[i] = yn[i]; b[i] = 0; c[i] = 0; d[i] = 0; }
    if (type == 1) {
        // Thomas algorithm; the diagonal lives in d[] and the right-hand side in b[] until c[] is known (no private arrays: the device
        // would put them in scratch memory); sub- and super-diagonal are recomputed where they are used
        double *dia = d, *rhs = b;
        auto sub = [&](int i) { return (i >= 1 && i < n - 1) ? (x[i] - x[i - 1]) / 3.0 : 0.0; };
        auto sup = [&](int i) { return (i >= 1 && i < n - 1) ? (x[i + 1] - x[i]) / 3.0 : 0.0; };
        for (int i = 0; i < n; i++) { dia[i] = 2.0; rhs[i] = 0; }
        for (int i = 1; i < n - 1; i++) {
            dia[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1]);
            rhs[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
        }
        for (int i = 1; i < n; i++) {
            const double w = sub(i) / dia[i - 1];
            dia[i] -= w * sup(i - 1);
            rhs[i] -= w * rhs[i - 1];
        }
        c[n - 1] = rhs[n - 1] / dia[n - 1];
        for (int i = n - 2; i >= 0; i--) c[i] = (rhs[i] - sup(i) * c[i + 1]) / dia[i];
        for (int i = 0; i < n - 1; i++) {
            const double h = x[i + 1] - x[i];
            d[i] = (c[i + 1] - c[i]) / (3.0 * h);
            b[i] = (y[i + 1] - y[i]) / h - (2.0 * c[i] + c[i + 1]) * h / 3.0;
        }
        const double h = x[n - 1] - x[n - 2];
        d[n - 1] = 0;
        b[n - 1] = 3.0 * d[n - 2] * h * h + 2.0 * c[n - 2] * h + b[n - 2];
    } else {
        for (int i = 1; i < n - 1; i++) {
            const double h = x[i + 1] - x[i], hl = x[i] - x[i - 1];
            b[i] = -h / (hl * (hl + h)) * y[i - 1] + (h - hl) / (hl * h) * y[i] + hl / (h * (hl + h)) * y[i + 1];
        }
        b[0] = 0.5 * (-b[1] + 3.0 * (y[1] - y[0]) / (x[1] - x[0]));
        b[n - 1] = 0.5 * (-b[n - 2] + 3.0 * (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
        for (int i = 0; i < n - 1; i++) {
            const double h = x[i + 1] - x[i];
            c[i] = (3.0 * (y[i + 1] - y[i]) / h - (2.0 * b[i] + b[i + 1])) / h;
            d[i] = ((b[i + 1] - b[i]) / (3.0 * h) - 2.0 / 3.0 * c[i]) / h;
        }
    }
    ri.sc0 = c[0];
    ri.bias_n = n;
    return true;
}

// First g mode of the ladder inside [lo, hi] (the reference's pair loop skips every other pair of this p mode, solver_mm.cpp:340-352), or -1.
// nu_g_of() does not increase with ig, so it is the first ig with nu_g <= hi, if that one is >= lo: found from the closed form
// ig >= 1e6/(hi DPl) - alpha - ng_min and settled with the very comparisons a walk over the ladder would make (a proposal far out in its
// prior can ask for 1e5 g modes: no walk).
__host__ __device__ inline int first_g_in_zone(const Prep &P, double lo, double hi) {
    if (P.Lg < 1) return -1;
    const double est = 1e6 / (hi * P.DPl) - P.alpha - (double)P.ng_min;
    long k;
    if (est == est && est > -1e15 && est < 1e15) {
        k = (long)est - 1;
        if (k < 0) k = 0;
        if (k > P.Lg - 1) k = P.Lg - 1;
        while (k > 0 && nu_g_of(P, (int)(k - 1)) <= hi) k--;
        while (k < P.Lg && !(nu_g_of(P, (int)k) <= hi)) k++;
    } else {
        k = 0;
        while (k < P.Lg && !(nu_g_of(P, (int)k) <= hi)) k++;
    }
    if (k >= P.Lg) return -1;
    return nu_g_of(P, (int)k) >= lo ? (int)k : -1;
}

__host__ __device__ inline double app_width(const double g[6], double f) {  // models.cpp:4788-4794
    const double lnGamma0 = g[2] * log(f / g[0]) + log(g[3]);
    const double e = 2. * log(f / g[1]) / log(g[4] / g[0]);
    return exp(lnGamma0 + -log(g[5]) / (1. + e * e));
}

// How unpack_vector spreads its loops: the host runs it as one thread; in the sampler's proposal kernel all 64 lanes of a wave run it together -- every
// lane computes the scalars, the loops over radial orders / p modes / Wigner terms / noise parameters are dealt one index per lane.
struct OneThread {
    static constexpr bool coop = false;
    __host__ __device__ int lane() const { return 0; }
    __host__ __device__ int lanes() const { return 1; }
    __host__ __device__ void sync() const {}
    double *w = nullptr;
};
struct WaveLanes {  // ONE wave of a (possibly larger) workgroup, in step through a wavefront barrier (its LDS operations complete in order)
    static constexpr bool coop = true;
    __device__ int lane() const { return (int)(threadIdx.x & 63); }
    __device__ int lanes() const { return 64; }
    __device__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    double *w;  // LDS [40]: Wigner terms (28) and elements (12)
};

// m-visibilities of degrees 1..lmax by >= 28 lanes: one lane per TERM of each Wigner sum, one per element, one per degree -- the sums in
// the order of mt::wigner_d / mt::amplitude_ratio (function_rot.cpp:15-101), the same values.
template <class X>
__device__ inline void amplitude_ratios_lanes(const X &x, int lmax, double beta_deg, double (*V)[7]) {
    const int lane = x.lane();
    const double PI = 3.141592653589793238462643;
    const double ang = PI * beta_deg / 180.;
    {
        int sl = 0, my_l = 0, my_i = 0, my_s = 0;
        double my_b = 0;
        for (int l = 1; l <= 3; l++)
            for (int e = 0; e <= l + 1; e++) {  // e = l+1: the centre element d^l_{0,0}(-beta)
                const int i = (e <= l) ? e : 0;
                for (int t = 0; t <= l - i; t++, sl++)
                    if (sl == lane) { my_l = l; my_i = i; my_s = t; my_b = (e <= l) ? ang : -ang; }
            }
        if (my_l > 0 && my_l <= lmax) x.w[lane] = mt::wigner_term(my_l, my_i, 0, my_b, my_s);
    }
    x.sync();
    if (lane < 12) {
        int l = 1, e = lane;
        if (lane >= 3) { l = 2; e = lane - 3; }
        if (lane >= 7) { l = 3; e = lane - 7; }
        if (l <= lmax) {
            int sl = 0;
            bool found = false;
            for (int ll = 1; ll <= l && !found; ll++)
                for (int ee = 0; ee <= ll + 1; ee++) {
                    if (ll == l && ee == e) { found = true; break; }
                    sl += ll - ((ee <= ll) ? ee : 0) + 1;
                }
            const int i = (e <= l) ? e : 0;
            double sum = 0;
            for (int t = 0; t <= l - i; t++) sum = sum + x.w[sl + t];
            x.w[28 + lane] = mt::wigner_finish(l, i, 0, sum);
        }
    }
    x.sync();
    if (lane >= 1 && lane <= 3 && lane <= lmax) {  // mirror, centre overwrite, square (function_rot.cpp:25-41)
        const int l = lane, base = 28 + (l == 1 ? 0 : (l == 2 ? 3 : 7));
        double *Vl = V[l];
        for (int i = 0; i <= l; i++) Vl[l + i] = x.w[base + i];
        for (int i = -l; i <= 0; i++) Vl[l + i] = Vl[l - i] * mt::pow_m1(i);
        Vl[l] = x.w[base + l + 1] * mt::pow_m1(0);
        for (int i = 0; i <= 2 * l; i++) Vl[i] = Vl[i] * Vl[i];
    }
    x.sync();
}

// models.cpp:4727-4866 (id 25) / :4377-4470 (id 27, cte_width: one width parameter, Wl0 constant, :4407) + solver_mm.cpp:470-555 /
// :624-705 (everything before the pair loop), then what the row builder needs besides the solver's output (:4867-5000), the bias spline
// and the vector's noise row.  Written once: the host calls it per vector (batched C-ABI path, host-driven sampler), the device engine
// runs it in its proposal kernel (dev_sampler.hip, propose_common) on the proposal it has just drawn.  Returns the vector's status (also left in P.status / ri.status).
template <class X>
__host__ __device__ inline int unpack_vector(const X &x, const double *p, const int32_t *pl, double step, bool cte_width, int dense, Prep &P, RowIn &ri,
                                             double *noise_row, int32_t *nh_out, int32_t *nn_out, double *fmin_out) {
    typedef mt::xreal xreal;  // long double on the host (as the reference computes these), double on the device
    const xreal pi = (xreal)3.14159265358979323846;  // M_PI widened, models.cpp:4779
    const int lane = x.lane(), NL = x.lanes();
    P.Lp = 0; P.Lg = 0; P.ng_min = 0; P.status = 0; P.probe_dense = dense; P.pad_ = 0;
    ri.Nfl0 = ri.Nfl2 = ri.Nfl3 = ri.lmax = ri.do_amp = ri.bias_n = ri.status = ri.cte_width = 0;
    const int Nmax = pl[0], lmax = pl[1], Nfl0 = pl[2], Nfl1 = pl[3], Nfl2 = pl[4], Nfl3 = pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Nnoise = pl[8], Ninc = pl[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const int os = Nmax + lmax + Nf, onoise = os + Nsplit + Nwidth, ocfg = onoise + Nnoise + Ninc, o1 = Nmax + lmax + Nfl0;
    auto fail = [&](int st) {  // (every lane takes the same decisions: the scalars are computed by all of them)
        P.status = st; ri.status = st;
        noise_row[0] = 1.0;  // placeholder row; the caller rejects / NaNs the evaluation
        *nh_out = 0; *nn_out = 1;
        return st;
    };
    const double trunc_c = p[ocfg];
    const bool do_amp = p[ocfg + 1] != 0;
    const double model_type = p[ocfg + 3], bias_type = p[ocfg + 4];
    const int Nferr = (int)p[ocfg + 5];
    if (Nmax < 2 || Nmax != Nfl0 || Nferr < 0 || Nfl1 != 8 + 2 * Nferr || Nwidth < (cte_width ? 1 : 6) || Nsplit < 10 || lmax > 3 || Nfl0 > MAXL ||
        Nfl2 > MAXL || Nfl3 > MAXL)
        return fail(TAMCMC_ERR_BAD_ARG);
    double g[6];
    for (int k = 0; k < 6; k++) g[k] = k < (cte_width ? 1 : 6) ? fabs(p[os + Nsplit + k]) : 0.0;
    const double *fl0 = p + Nmax + lmax;
    double fmin = fl0[0], fmax = fl0[0];
    for (int n = 0; n < Nmax; n++) {
        fmin = fl0[n] < fmin ? fl0[n] : fmin;
        fmax = fl0[n] > fmax ? fl0[n] : fmax;
    }
    for (int n = lane; n < Nmax; n += NL) {
        const double W = cte_width ? g[0] : app_width(g, fl0[n]);
        ri.fl0[n] = fl0[n];
        ri.Wl0[n] = W;
        const xreal ha = p[n] * (1. / W / pi);
        ri.Hl0[n] = do_amp ? (double)(ha < 0 ? -ha : ha) : fabs(p[n]);
    }
    *fmin_out = fmin;
    const double delta0l = p[o1], DPl = fabs(p[o1 + 1]), alpha_g = fabs(p[o1 + 2]), q = fabs(p[o1 + 3]);
    double fit[2];
    mt::linfit_index(fl0, Nfl0, fit);
    const double Dnu_p = fit[0];
    // the reference exits (models.cpp:4851-4857; id 27 only tests it for model_type 0, :4459, and would otherwise hand its solver a zero
    // lower bound, i.e. an unbounded g-mode count: refused here too)
    if (!(Dnu_p > 0) || fmin - Dnu_p < 0) return fail(TAMCMC_ERR_BAD_ARG);
    P.Dnu_p = Dnu_p; P.DPl = DPl; P.alpha = alpha_g; P.q = q; P.resol = step; P.fact = 0.04;
    P.zone = 0; P.keep_lo = 0; P.keep_hi = 0;
    double fmin_s = fmin - Dnu_p, fmax_s = fmax + Dnu_p;
    bool no_modes = false;
    if (model_type == 0) {  // solve_mm_asymptotic_O2p(Dnu_p, eps, 1, delta0l, 0, 0, ...), fmin - Dnu .. fmax + Dnu
        const int n0 = (int)floor(fit[1] / Dnu_p);
        const double eps = fit[1] / Dnu_p - n0;
        const int el = 1;
        int np_min = (int)floor(fmin_s / Dnu_p - eps - el / 2 - delta0l);  // el/2: integer division, as in the reference
        int np_max = (int)ceil(fmax_s / Dnu_p - eps - el / 2 - delta0l);
        int ng_min = (int)floor(1e6 / (fmax_s * DPl) - alpha_g), ng_max = (int)ceil(1e6 / (fmin_s * DPl) - alpha_g);
        if (ng_min <= 0 && ng_max < 1) no_modes = true;  // "impossible star": no mixed modes, the model carries on (solver_mm.cpp:497-501)
        else {
            if (ng_min <= 0 && ng_max >= 1) ng_min = 1;
            P.zone = (ng_max - ng_min < 6) ? (double)np_max : 1.75;
            if (np_min <= 0) np_min = 1;
            P.Lp = np_max - np_min; P.Lg = ng_max - ng_min; P.ng_min = ng_min;
            if (P.Lg < 1) no_modes = true;  // no g mode in range
            else {
                if (P.Lp < 1 || P.Lp > MAXP) return fail(TAMCMC_ERR_BAD_ARG);
                for (int np = np_min + lane; np < np_max; np += NL) {
                    P.nu_p[np - np_min] = (double)((np + (xreal)eps + el / (xreal)2. + delta0l) * Dnu_p);
                    P.dnu_loc[np - np_min] = Dnu_p;  // alpha_p = 0
                }
                P.keep_lo = fmin_s; P.keep_hi = fmax_s;
            }
        }
    } else {  // solve_mm_asymptotic_O2from_l0(fl0, 1, delta0l, ...): the l=0 ladder shifted, three extra orders on each side
        if (fmin_s < 0) fmin_s = 0;
        int ng_min = (int)floor(1e6 / (fmax_s * DPl) - alpha_g), ng_max = (int)ceil(1e6 / (fmin_s * DPl) - alpha_g);
        if (ng_min <= 0 && ng_max < 1) no_modes = true;
        else {
            if (ng_min <= 0 && ng_max >= 1) ng_min = 1;
            P.zone = (ng_max - ng_min < 6) ? 20. : 1.75;
            int Lp = 0;
            const double shift = (double)(1 / (xreal)2. * Dnu_p + delta0l);
            for (int k = 0; k < Nfl0 + 6; k++) {  // (a running count: every lane walks the list)
                double e;
                if (k < 3) e = fmin - (3 - k) * Dnu_p;
                else if (k < 3 + Nfl0) e = fl0[k - 3];
                else e = fmax + (k - 2 - Nfl0) * Dnu_p;
                const double v = e + shift;
                if (v >= fmin_s && v <= fmax_s) {
                    if (Lp >= MAXP) return fail(TAMCMC_ERR_BAD_ARG);
                    P.nu_p[Lp++] = v;
                }
            }
            P.Lp = Lp; P.Lg = ng_max - ng_min; P.ng_min = ng_min;
            if (P.Lg < 1) no_modes = true;
            else if (P.Lp < 2) return fail(TAMCMC_ERR_BAD_ARG);
            P.keep_lo = fmin; P.keep_hi = fmax;
        }
    }
    if (no_modes) P.Lp = 0;
    if (fmin_s <= 150) P.fact = 0.01;
    if (fmin_s <= 50) P.fact = 0.005;
    x.sync();  // the p ladder is complete
    const int Lp = P.Lp;
    const double zone = P.zone;
    // first derivative of the p ladder on the index grid (derivatives_handler.cpp:425-457)
    for (int i = lane; i < Lp; i += NL) {
        double d;
        if (Lp == 1) d = 0;
        else if (i == 0) d = P.nu_p[1] - P.nu_p[0];
        else if (i == Lp - 1) d = P.nu_p[i] - P.nu_p[i - 1];
        else d = (P.nu_p[i + 1] - P.nu_p[i - 1]) / 2.;
        P.dnup[i] = d;
        if (model_type != 0) P.dnu_loc[i] = d;  // the from-l0 driver hands the local derivative to the solver (:717)
        P.ig0[i] = first_g_in_zone(P, P.nu_p[i] - zone * Dnu_p, P.nu_p[i] + zone * Dnu_p);
    }
    // ---- the row builder's inputs
    if (bias_type != 0) {
        if (!spline_set(ri, p + o1 + 8, p + o1 + 8 + Nferr, Nferr, bias_type == 1 ? 1 : 2)) return fail(TAMCMC_ERR_BAD_ARG);
    }
    ri.Nfl0 = Nfl0; ri.Nfl2 = Nfl2; ri.Nfl3 = Nfl3; ri.lmax = lmax; ri.do_amp = do_amp ? 1 : 0; ri.cte_width = cte_width ? 1 : 0;
    for (int k = lane; k < Nfl2; k += NL) ri.fl2[k] = fabs(p[o1 + Nfl1 + k]);
    for (int k = lane; k < Nfl3; k += NL) ri.fl3[k] = fabs(p[o1 + Nfl1 + Nfl2 + k]);
    for (int k = 0; k < 6; k++) ri.g[k] = g[k];
    const double inclination = fabs(p[onoise + Nnoise]);
    ri.Vl[0] = 1;
    for (int l = 1; l <= 3; l++) ri.Vl[l] = l <= lmax ? fabs(p[Nmax + l - 1]) : 0.0;
    for (int k = lane; k < 28; k += NL) ri.V[k / 7][k % 7] = (k == 0) ? 1.0 : 0.0;
    x.sync();
    if constexpr (X::coop) amplitude_ratios_lanes(x, lmax, inclination, ri.V);
    else
        for (int l = 1; l <= lmax; l++) mt::amplitude_ratio(l, inclination, ri.V[l]);
    ri.eta0 = (p[os + 8] == 1) ? mt::eta0_fct(fl0, Nfl0) : 0.0;
    ri.asym = p[os + 9]; ri.trunc_c = trunc_c;
    ri.Wfactor = fabs(p[o1 + 6]); ri.Hfactor = fabs(p[o1 + 7]);
    ri.rot_env = fabs(p[os]); ri.rot_core = fabs(p[os + 1]);
    ri.a2 = p[os + 2]; ri.a3 = p[os + 4]; ri.a4 = p[os + 5]; ri.a5 = p[os + 6]; ri.a6 = p[os + 7];
    ri.fmin = fmin; ri.fmax = fmax;
    for (int k = lane; k < Nnoise; k += NL) noise_row[k] = fabs(p[onoise + k]);
    *nh_out = (Nnoise - 1) / 3; *nn_out = Nnoise;
    return TAMCMC_OK;
}

// One chain group's slice of the pre-step workspace (device memory), as the sampler's proposal kernel sees it: it writes vector b's Prep /
// RowIn and zeroes the solver's counters; k_armm_solve and k_rgb_finish (rgb_prestep.hip) take it from there.
struct Slice {
    Prep *preps = nullptr;
    RowIn *rows = nullptr;
    unsigned long long *norm_bits = nullptr;
    int *nsol = nullptr;
    double step = 0;      // x[2] - x[1] (models.cpp:4719)
    int dense = 0;        // TAMCMC_OPT_ARMM_DENSE_SCAN
};

}  // namespace rgb
}  // namespace tamcmc
