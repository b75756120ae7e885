// rgb_prestep.hip -- red-giant model model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (id 25, tamcmc/sources/models.cpp:4684-5079):
// the per-proposal pre-step that turns a parameter vector into a VARIABLE-LENGTH multiplet table for k_loglike.
//
//   host   : scalar unpack (width law, l=0 heights, linear fit of the l=0 ladder, p / g asymptotic ladders)      models.cpp:4755-4866
//   device : ARMM mixed-mode solver -- scan p(nu)-g(nu) on the resol grid for sign changes, refine each on the
//            fine local grid by inverse linear interpolation, keep true intersections        external/ARMM/solver_mm.cpp:340-443
//            one workgroup per (parameter vector, p mode); sort + tolerance-unique per vector                     :586-593
//   device : + spline bias of the frequencies (coefficients from the host: cubic / Hermite, natural ends)   external/spline/src/spline.h:242-498
//   device : zeta function at the mixed modes and its normalisation, max over a 4-year-resolution grid of the
//            sum over all (p, g) pairs (collapsed to one term per p mode, see ksi_sum)   external/ARMM/bump_DP.cpp:46-78, :125-188
//   device : mixed-mode heights / widths / rotational splittings, windows, table rows    bump_DP.cpp:203-254, :531-547; models.cpp:4867-5000
//
// The reference runs the solver for every (p mode, g mode) pair whose g mode lies within the search zone of the p mode.  g(nu)
// depends on nu_g only through tan(pi 1e6 (1/nu - 1/nu_g)/DPl), which is the same function for every g mode of the ladder
// (1e6/(nu_g DPl) = n_g + alpha), so all those pairs return the same roots up to rounding and the duplicates are removed
// afterwards; here each p mode is solved ONCE, against the first g mode inside its zone (~40x less work, same roots to ~1e-12).
// Behaviour kept from the reference: a root closer than 2*resol to a pole of tan() is lost (its refinement window holds the
// pole, the interpolation extrapolates and the 0.1 % ratio test rejects it) -- tests/test_oracle_rgb.py documents it.
#include <hip/hip_runtime.h>

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"
#include "mode_tables.h"
#include "mode_tables_impl.h"
#include "rgb_prestep.h"
#include "kernels.h"

namespace tamcmc {
const mt::PolyTab &poly_table();  // mode_tables.cpp
namespace rgb {

constexpr int MAXP = 32;      // p modes per vector (fmax-fmin+2 Dnu)/Dnu + margins
constexpr int MAXSOL = 1024;  // mixed modes per vector before de-duplication
constexpr int WG = 256;

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

namespace {

__device__ __forceinline__ double f_pg(double nu, double nu_p, double nu_g, double Dnu, double DPl, double q) {
    const double PI = 3.141592653589793238;
    const double X = PI * (1. / nu - 1. / nu_g) * 1e6 / DPl;
    return (nu - nu_p) - Dnu * atan(q * tan(X)) / PI;
}
__device__ __forceinline__ bool changes_sign(double a, double b) {  // sign_change(), solver_mm.cpp:82-121
    return ((b >= 0 && a < 0) || (b > 0 && a <= 0)) || (b <= 0 && a >= 0);
}

// One wave per (vector b, p mode ip): solver_mm.cpp:330-406 for that pair.
//  scan  : candidate = grid index i with a sign change of p-g between grid points i and i+1 (the reference walks the whole grid).
//          p-g has a simple structure: tan X has its poles where kappa(nu) = 1e6/DPl (1/nu - 1/nu_g) - 1/2 is an integer; at a pole p-g
//          drops by Dl, between two poles it is continuous and strictly increasing (nu - nu_p increases, X decreases).  So the cells
//          with a sign change are at most ONE per pole-free stretch -- found by bisection on the grid index (the same grid points, ~11
//          evaluations instead of hundreds) -- and the three cells around each pole, which are evaluated directly (the pole's own
//          cell changes sign when the drop crosses zero; one cell of margin on each side absorbs the rounding of the pole position).
//          One LANE per unit: unit 0 = the stretch before the first pole, unit j+1 = pole j and the stretch after it.  Windows that
//          start at nu = 0 or hold an unreasonable number of poles take the dense walk (TAMCMC_OPT_ARMM_DENSE_SCAN forces it).
//  refine: one LANE per candidate: the local grid of step resol*fact over [x0 - 2 resol, x0 + 2 resol], lin_interpol with x = p-g,
//          y = nu at 0 (interpol.cpp:13-43), the 0.1 % ratio test.  Without a pole inside the window p-g is increasing there and the
//          bracketing pair is found by bisection; the rare windows that hold both a bracket and a pole are walked point by point by
//          the whole wave afterwards, like the reference does.
__global__ void __launch_bounds__(64) k_armm_scan(const Prep *preps, double *sols, int *nsol) {
    const int b = blockIdx.y, ip = blockIdx.x, lane = threadIdx.x;
    const Prep &P = preps[b];
    if (ip >= P.Lp || P.status != 0 || P.ig0[ip] < 0) return;
    const double nu_p = P.nu_p[ip], Dl = P.dnu_loc[ip], nu_g = nu_g_of(P, P.ig0[ip]);
    const double numin = nu_p - P.zone * P.Dnu_p, numax = nu_p + P.zone * P.Dnu_p;
    long n;
    double lo;
    if (numin >= 0) { n = (long)((numax - numin) / P.resol); lo = numin; }
    else { n = (long)(numax / P.resol); lo = 0; }
    if (n < 2) return;
    const double step = (numax - lo) / (double)(n - 1);
    auto grid = [&](long i) { return (i == n - 1) ? numax : lo + (double)i * step; };
    auto fg = [&](long i) { return f_pg(grid(i), nu_p, nu_g, Dl, P.DPl, P.q); };
    constexpr int MAXC = 1024;
    __shared__ long s_cand[MAXC];
    __shared__ int s_hard[MAXC];
    __shared__ int s_nc, s_nh;
    if (lane == 0) { s_nc = 0; s_nh = 0; }
    __syncthreads();
    auto push = [&](long i) {
        const int k = atomicAdd(&s_nc, 1);
        if (k < MAXC) s_cand[k] = i;
    };
    const double kap_scale = 1e6 / P.DPl, inv_g = 1. / nu_g;
    const double k_hi = kap_scale * (1. / lo - inv_g) - 0.5, k_lo = kap_scale * (1. / numax - inv_g) - 0.5;
    const bool structured = (lo > 0) && isfinite(k_hi) && isfinite(k_lo) && (k_hi - k_lo) < 1.0e5 && P.q > 0 && Dl > 0 && !P.probe_dense;
    if (structured) {
        const long kmax = (long)floor(k_hi) + 1, kmin = (long)ceil(k_lo) - 1;  // one pole beyond each end: harmless, clipped below
        const long np = kmax - kmin + 1;
        auto pole_cell = [&](long j) -> long {  // cell holding pole j (poles in ascending frequency), clipped to [-3, n+1]
            const double nu_k = 1. / (inv_g + ((double)(kmax - j) + 0.5) / kap_scale);
            double c = floor((nu_k - lo) / step);
            if (!(c > -3.0)) c = -3.0;
            if (!(c < (double)(n + 1))) c = (double)(n + 1);
            return (long)c;
        };
        for (long u = lane; u <= np; u += 64) {
            long a, e;  // the pole-free stretch of this unit: cells with left index a .. e
            if (u == 0) { a = 0; e = pole_cell(0) - 2; }
            else {
                const long j = u - 1, cj = pole_cell(j);
                const long prev_end = (j > 0) ? pole_cell(j - 1) + 1 : -1;  // last cell of the previous pole's zone
                for (long i = (cj - 1 > prev_end ? cj - 1 : prev_end + 1); i <= cj + 1; i++)
                    if (i >= 0 && i <= n - 2 && changes_sign(fg(i), fg(i + 1))) push(i);
                a = cj + 2;
                e = (j + 1 < np) ? pole_cell(j + 1) - 2 : n - 2;
            }
            if (a < 0) a = 0;
            if (e > n - 2) e = n - 2;
            if (a > e) continue;
            const double fa = fg(a), fe = fg(e + 1);
            if (fa > 0.0 || fe < 0.0) continue;      // increasing and of one sign: no change in this stretch
            if (!(fa < 0.0)) { push(a); continue; }  // an exact zero on the first point
            long lo_i = a, hi_i = e + 1;             // f[lo_i] < 0 <= f[hi_i]
            while (hi_i - lo_i > 1) {
                const long mid = lo_i + (hi_i - lo_i) / 2;
                if (fg(mid) >= 0.0) hi_i = mid; else lo_i = mid;
            }
            push(lo_i);
        }
    } else {
        // dense walk: each lane evaluates ONE point per pass; its right neighbour's value comes from the next lane (lane 63
        // evaluates that one point more)
        for (long i0 = 0; i0 < n - 1; i0 += 64) {
            const long i = i0 + lane;
            const double fa = (i < n) ? fg(i) : 0.0;
            double fb = __shfl_down(fa, 1, 64);
            if (lane == 63 && i + 1 < n) fb = fg(i + 1);
            if (i < n - 1 && changes_sign(fa, fb)) push(i);
        }
    }
    __syncthreads();
    if (s_nc > MAXC) {  // more sign changes than the candidate list holds: flag the vector (the host reports it), do not guess
        if (lane == 0) atomicAdd(&nsol[b], 2 * MAXSOL);
        return;
    }
    const int nc = s_nc;
    struct Local {  // the local grid of one candidate
        double rmin, rmax, ls;
        long nl;
    };
    auto local_of = [&](long cand) {
        Local L;
        const double x0 = grid(cand);
        L.rmin = x0 - 2 * P.resol; L.rmax = x0 + 2 * P.resol;
        L.nl = (long)((L.rmax - L.rmin) / (P.resol * P.fact));
        L.ls = L.nl >= 2 ? (L.rmax - L.rmin) / (double)(L.nl - 1) : 0.0;
        return L;
    };
    auto lg = [&](const Local &L, long j) { return (j == L.nl - 1) ? L.rmax : L.rmin + (double)j * L.ls; };
    auto fl = [&](const Local &L, long j) { return f_pg(lg(L, j), nu_p, nu_g, Dl, P.DPl, P.q); };
    // straight line through two local points evaluated at p-g = 0, then the ratio test (solver_mm.cpp:392-404)
    auto finish = [&](const Local &L, long best, double f_first, double f_last) {
        double a = 0, bb = 0;
        if (0.0 >= f_first && 0.0 <= f_last) {
            const long j = best < L.nl - 1 ? best : L.nl - 2;
            const double fa = fl(L, j), fb = fl(L, j + 1);
            a = (lg(L, j + 1) - lg(L, j)) / (fb - fa);
            bb = lg(L, j) - a * fa;
        }
        if (0.0 < f_first) {
            a = (lg(L, 1) - lg(L, 0)) / (fl(L, 1) - f_first);
            bb = lg(L, 0) - a * f_first;
        }
        if (0.0 > f_last) {
            const double fa = fl(L, L.nl - 2);
            a = (lg(L, L.nl - 1) - lg(L, L.nl - 2)) / (f_last - fa);
            bb = lg(L, L.nl - 2) - a * fa;
        }
        const double prop = a * 0.0 + bb;
        const double PI = 3.141592653589793238;
        const double X = PI * (1. / prop - 1. / nu_g) * 1e6 / P.DPl;
        const double ratio = (Dl * atan(P.q * tan(X)) / PI) / (prop - nu_p);
        if (ratio >= 0.999 && ratio <= 1.001 && prop >= P.keep_lo && prop <= P.keep_hi) {
            const int k = atomicAdd(&nsol[b], 1);
            if (k < MAXSOL) sols[(size_t)b * MAXSOL + k] = prop;
        }
    };
    for (int c = lane; c < nc; c += 64) {
        const Local L = local_of(s_cand[c]);
        if (L.nl < 2) continue;
        const double f_first = fl(L, 0), f_last = fl(L, L.nl - 1);
        // first j with f[j] <= 0 <= f[j+1] -- only needed when lin_interpol interpolates (f_first <= 0 <= f_last); otherwise it
        // extrapolates from the first or last two points (a pole of tan(): half of all candidates) and no search is made
        long best = L.nl;
        if (!(0.0 < f_first) && !(0.0 > f_last)) {
            const double ka = kap_scale * (1. / L.rmin - inv_g) - 0.5, kb = kap_scale * (1. / L.rmax - inv_g) - 0.5;
            const bool pole_inside = (floor(ka) != floor(kb)) || fabs(ka - rint(ka)) < 1e-9 || fabs(kb - rint(kb)) < 1e-9;
            if (pole_inside) {  // left to the whole wave below
                const int k = atomicAdd(&s_nh, 1);
                s_hard[k] = c;
                continue;
            }
            long lo_j = 0, hi_j = L.nl - 1;  // f[lo_j] <= 0 <= f[hi_j], p-g increasing: the last point with f <= 0
            while (hi_j - lo_j > 1) {
                const long mid = lo_j + (hi_j - lo_j) / 2;
                if (fl(L, mid) <= 0.0) lo_j = mid; else hi_j = mid;
            }
            // step back over exact zeros so that the FIRST pair with f[j] <= 0 <= f[j+1] is the one used
            best = lo_j;
            while (best > 0 && fl(L, best - 1) <= 0.0 && fl(L, best) >= 0.0 && !(fl(L, best) > 0.0)) best--;
        }
        finish(L, best, f_first, f_last);
    }
    __syncthreads();
    const int nh = s_nh;
    for (int hc = 0; hc < nh; hc++) {  // wave-uniform
        const Local L = local_of(s_cand[s_hard[hc]]);
        const double f_first = fl(L, 0), f_last = fl(L, L.nl - 1);
        long best = L.nl;
        for (long j0 = 0; j0 < L.nl - 1; j0 += 64) {
            const long j = j0 + lane;
            bool hit = false;
            if (j < L.nl - 1) {
                const double fa = fl(L, j), fb = fl(L, j + 1);
                hit = !(0.0 < fa || 0.0 > fb);  // the loop condition of lin_interpol, negated
            }
            const unsigned long long m = __ballot(hit);
            if (m) { best = j0 + (long)(__ffsll((long long)m) - 1); break; }
        }
        if (lane == 0) finish(L, best, f_first, f_last);
    }
}

// One workgroup per vector: bitonic sort of its solutions, then std::unique with |a-b| <= 2 resol (solver_mm.cpp:586-593).
__global__ void __launch_bounds__(WG) k_armm_sort_unique(const Prep *preps, const RowIn *rows_in, double *sols, int *nsol, double *fl1) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ double s[MAXSOL];
    if (nsol[b] > MAXSOL) return;  // overflow flag: left for the host
    const int n = nsol[b];
    int N2 = 64;  // bitonic network over the next power of two (padding sorts to the end)
    while (N2 < n) N2 <<= 1;
    for (int i = tid; i < N2; i += WG) s[i] = i < n ? sols[(size_t)b * MAXSOL + i] : INFINITY;
    __syncthreads();
    for (int k = 2; k <= N2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < N2; i += WG) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const double a = s[i], c = s[l];
                    if ((a > c) == up) { s[i] = c; s[l] = a; }
                }
            }
            __syncthreads();
        }
    __shared__ int s_m;
    if (tid == 0) {  // std::unique keeps an element unless it is within tol of the last KEPT one: a serial chain, run in LDS (in place)
        const double tol = 2 * preps[b].resol;
        int m = 0;
        double last = 0;
        for (int i = 0; i < n; i++) {
            const double v = s[i];
            if (m == 0 || !(fabs(last - v) <= tol)) { s[m++] = v; last = v; }
        }
        nsol[b] = m;
        s_m = m;
    }
    __syncthreads();
    for (int i = tid; i < s_m; i += WG) sols[(size_t)b * MAXSOL + i] = s[i];
    // frequency bias of the mixed modes: spline through the (fref, ferr) nodes (models.cpp:4833-4842, :4873-4880; spline.h:476-498)
    const RowIn &R = rows_in[b];
    const int m = s_m;
    for (int i = tid; i < m; i += WG) {
        const double v = s[i];
        double bias = 0;
        if (R.bias_n >= 3) {
            const int nn = R.bias_n;
            int idx = 0;
            while (idx + 1 < nn && R.sx[idx + 1] <= v) idx++;
            const double h = v - R.sx[idx];
            if (v < R.sx[0]) bias = (R.sc0 * h + R.sb[0]) * h + R.sy[0];
            else if (v > R.sx[nn - 1]) bias = (R.sc[nn - 1] * h + R.sb[nn - 1]) * h + R.sy[nn - 1];
            else bias = ((R.sd[idx] * h + R.sc[idx]) * h + R.sb[idx]) * h + R.sy[idx];
        }
        fl1[(size_t)b * MAXSOL + i] = v + bias;
    }
}

// One (p, g) term of the zeta function (bump_DP.cpp:46-78):
//   1 / (1 + front cos^2(up)/cos^2(down)),  up = pi 1e6 (1/nu - 1/nu_g)/DPl,  down = pi (nu - nu_p)/Dnu_p,  front = 1e-6 nu^2 DPl/(q Dnu_p).
// Sum over all (p, g) pairs.  cos^2 has period pi and the g ladder is regular (1e6/(nu_g DPl) = n_g + alpha), so the term does not
// depend on WHICH g mode is used: the reference's inner loop over the g modes adds L_g copies of the same number (up to the
// rounding of its argument, ~1e-13 relative); here it is evaluated once, with the ladder's middle mode, and multiplied.
// The g-mode cosine and nu^2 DPl/q are common to all p modes; 1/(1 + front cu^2/cd^2) = cd^2/(cd^2 + front cu^2) (one division; the
// limits cd -> 0 and front -> inf give the same 0, 0/0 the same NaN).
__device__ __forceinline__ double ksi_sum(const Prep &P, double nu) {
    const double PI = 3.14159265358979323846;
    const double nu_g = nu_g_of(P, P.Lg / 2);
    const double cu = cos(PI * 1e6 * (1. / nu - 1. / nu_g) / P.DPl);
    const double cu2 = cu * cu, fr = 1e-6 * nu * nu * P.DPl / P.q;
    double s = 0;
    for (int ip = 0; ip < P.Lp; ip++) {
        const double cd = cos(PI * (nu - P.nu_p[ip]) / P.dnup[ip]);
        const double cd2 = cd * cd;
        s += (double)P.Lg * (cd2 / (cd2 + (fr / P.dnup[ip]) * cu2));
    }
    return s;
}

// grid (chunks, B): un-normalised zeta at the vector's modes (chunk 0) and the maximum of the same sum over the high-resolution grid
__global__ void __launch_bounds__(WG) k_zeta(const Prep *preps, const double *fl1, const int *n1, double *ksi, unsigned long long *norm_bits,
                                             int chunks) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const Prep &P = preps[b];
    if (P.status != 0 || P.Lp < 1 || P.Lg < 1) return;  // failed vector, or no g mode in range (no mixed modes)
    if (blockIdx.x == 0)
        for (int i = tid; i < n1[b]; i += WG) ksi[(size_t)b * MAXSOL + i] = ksi_sum(P, fl1[(size_t)b * MAXSOL + i]);
    double pmin = P.nu_p[0], pmax = P.nu_p[0];
    for (int i = 1; i < P.Lp; i++) { pmin = fmin(pmin, P.nu_p[i]); pmax = fmax(pmax, P.nu_p[i]); }
    const double gmax = nu_g_of(P, 0), gmin = nu_g_of(P, P.Lg - 1);  // the g ladder decreases with n_g
    const double lo = pmin >= gmin ? gmin : pmin, hi = pmax >= gmax ? pmax : gmax;
    const double resol = 1e6 / (4 * 365. * 86400.);
    const long nh = (long)((hi - lo) / resol);
    double best = 0;
    if (nh >= 2) {
        const double step = (hi - lo) / (double)(nh - 1);
        for (long i = (long)blockIdx.x * WG + tid; i < nh; i += (long)chunks * WG) {
            const double v = ksi_sum(P, (i == nh - 1) ? hi : lo + (double)i * step);
            if (v > best) best = v;
        }
    }
    // one atomic per workgroup (atomics on one address serialise in L2)
    __shared__ double s_best[WG / 64];
    for (int off = 32; off >= 1; off >>= 1) best = fmax(best, __shfl_down(best, off, 64));
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < WG / 64; w++) best = fmax(best, s_best[w]);
        if (best > 0) atomicMax(&norm_bits[b], (unsigned long long)__double_as_longlong(best));  // positive doubles order as integers
    }
}

// One workgroup per vector: the table rows (l=0 list, mixed modes, l=2, l=3 lists) written into the likelihood kernel's input block,
// in the reference's accumulation order (models.cpp:4915-5000); mixed-mode scalars: bump_DP.cpp:203-254, :531-547.
__global__ void __launch_bounds__(WG) k_rgb_rows(const Prep *preps, const RowIn *rows_in, const mt::PolyTab *poly, const double *fl1, const int *n1,
                                                 const double *ksi, const unsigned long long *norm_bits, double x_first, double x_last, long Nx,
                                                 double step, int per, tamcmc_multiplet *mults, int *pairs, int *status) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const RowIn &R = rows_in[b];
    __shared__ double s_fi[MAXL + 4], s_hi[MAXL + 4];
    __shared__ int s_st;
    if (tid == 0) s_st = (R.status != 0 || preps[b].status != 0) ? (R.status ? R.status : preps[b].status) : 0;
    const int ni = R.Nfl0 + 4;
    if (tid < ni) {  // l=0 heights on a grid that falls to zero beyond the observed orders (models.cpp:4884-4897)
        double f, h;
        if (tid == 0) { f = R.fmin * 0.6; h = 0; }
        else if (tid == 1) { f = R.fmin * 0.8; h = R.Hl0[0] / 4; }
        else if (tid == ni - 2) { f = R.fmax * 1.2; h = R.Hl0[R.Nfl0 - 1] / 4; }
        else if (tid == ni - 1) { f = R.fmax * 1.4; h = 0; }
        else { f = R.fl0[tid - 2]; h = R.Hl0[tid - 2]; }
        s_fi[tid] = f; s_hi[tid] = h;
    }
    __syncthreads();
    int N1 = n1[b];
    if (s_st == 0 && (N1 > CAP1 || N1 < 0)) { if (tid == 0) s_st = TAMCMC_ERR_BAD_ARG; N1 = 0; }
    __syncthreads();
    const bool ok = (s_st == 0);
    const int total = ok ? R.Nfl0 + N1 + R.Nfl2 + R.Nfl3 : 0;
    double norm = __longlong_as_double((long long)norm_bits[b]);
    const double PIL = 3.14159265358979323846;
    for (int k = tid; k < total; k += WG) {
        int l;
        double f, H, W, a[7] = {0, 0, 0, 0, 0, 0, 0}, eta0 = R.eta0;
        if (k < R.Nfl0) { l = 0; f = R.fl0[k]; H = R.Hl0[k]; W = R.Wl0[k]; eta0 = 0.0; }
        else if (k < R.Nfl0 + N1) {
            l = 1;
            const int i = k - R.Nfl0;
            f = fl1[(size_t)b * MAXSOL + i];
            double z = ksi[(size_t)b * MAXSOL + i] / norm;
            if (z > 1) z = 1;
            double hr = sqrt(1. - R.Hfactor * z);
            if (hr > -1e-5 && hr < 1e-5) hr = 1e-10;
            const double t = mt::lin_interpol(s_fi, s_hi, ni, f);
            const double Hp = t < 0 ? 0.0 : fabs(t);
            H = fabs(hr * (Hp * R.Vl[1]));
            W = mt::lin_interpol(R.fl0, R.Wl0, R.Nfl0, f) * (1. - R.Wfactor * z) / sqrt(hr);
            a[1] = fabs(z * (R.rot_core / 2 - R.rot_env) + R.rot_env);
        } else {
            const bool is2 = k < R.Nfl0 + N1 + R.Nfl2;
            l = is2 ? 2 : 3;
            f = is2 ? R.fl2[k - R.Nfl0 - N1] : R.fl3[k - R.Nfl0 - N1 - R.Nfl2];
            if (R.cte_width) W = R.g[0];  // models.cpp:4561, :4582
            else {
                const double lnGamma0 = R.g[2] * log(f / R.g[0]) + log(R.g[3]);
                const double e = 2. * log(f / R.g[1]) / log(R.g[4] / R.g[0]);
                W = exp(lnGamma0 + -log(R.g[5]) / (1. + e * e));
            }
            H = mt::lin_interpol(R.fl0, R.Hl0, R.Nfl0, f);
            H = R.do_amp ? fabs(H / (PIL * W) * R.Vl[l]) : fabs(H * R.Vl[l]);
            a[1] = R.rot_env; a[2] = R.a2; a[3] = R.a3; a[4] = R.a4;
            if (!is2) { a[5] = R.a5; a[6] = R.a6; }
        }
        tamcmc_multiplet *r = &mults[(size_t)b * per + k];
        int i0 = 0, i1 = 0;
        const int rs = mt::set_imin_imax(x_first, x_last, Nx, l, f, W, a[1], R.trunc_c, step, &i0, &i1);
        if (rs) { s_st = rs; continue; }
        r->l = l; r->i0 = i0; r->i1 = i1; r->flags = 0; r->fc = f; r->gamma = W; r->asym = R.asym;
        for (int q = 0; q < 7; q++) { r->nu[q] = 0; r->hv[q] = 0; }
        for (int m = -l; m <= l; m++) {
            r->nu[m + l] = l == 0 ? f : mt::nu_nlm_aj(*poly, f, a, eta0, l, m);
            r->hv[m + l] = H * R.V[l][m + l];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const bool good = (s_st == 0);
        pairs[2 * b] = b * per;
        pairs[2 * b + 1] = good ? b * per + total : b * per;
        status[b] = s_st;
    }
}

__global__ void k_fill_poly_rgb(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
}

// ---------------------------------------------------------------- host side
struct Spline {  // natural cubic (type 1) or cubic Hermite (type 2) through the bias nodes, spline.h:242-498
    std::vector<double> x, y, b, c, d;
    double c0 = 0;
    bool set(const double *xn, const double *yn, int n, int type) {
        if (n < 3) return false;
        for (int i = 0; i < n - 1; i++)
            if (!(xn[i] < xn[i + 1])) return false;
        x.assign(xn, xn + n); y.assign(yn, yn + n); b.assign((size_t)n, 0); c.assign((size_t)n, 0); d.assign((size_t)n, 0);
        if (type == 1) {
            std::vector<double> sub((size_t)n, 0), dia((size_t)n, 2.0), sup((size_t)n, 0), rhs((size_t)n, 0);
            for (int i = 1; i < n - 1; i++) {
                sub[(size_t)i] = (x[(size_t)i] - x[(size_t)i - 1]) / 3.0;
                dia[(size_t)i] = 2.0 / 3.0 * (x[(size_t)i + 1] - x[(size_t)i - 1]);
                sup[(size_t)i] = (x[(size_t)i + 1] - x[(size_t)i]) / 3.0;
                rhs[(size_t)i] = (y[(size_t)i + 1] - y[(size_t)i]) / (x[(size_t)i + 1] - x[(size_t)i]) -
                                 (y[(size_t)i] - y[(size_t)i - 1]) / (x[(size_t)i] - x[(size_t)i - 1]);
            }
            for (int i = 1; i < n; i++) {
                const double w = sub[(size_t)i] / dia[(size_t)i - 1];
                dia[(size_t)i] -= w * sup[(size_t)i - 1];
                rhs[(size_t)i] -= w * rhs[(size_t)i - 1];
            }
            c[(size_t)n - 1] = rhs[(size_t)n - 1] / dia[(size_t)n - 1];
            for (int i = n - 2; i >= 0; i--) c[(size_t)i] = (rhs[(size_t)i] - sup[(size_t)i] * c[(size_t)i + 1]) / dia[(size_t)i];
            for (int i = 0; i < n - 1; i++) {
                const double h = x[(size_t)i + 1] - x[(size_t)i];
                d[(size_t)i] = (c[(size_t)i + 1] - c[(size_t)i]) / (3.0 * h);
                b[(size_t)i] = (y[(size_t)i + 1] - y[(size_t)i]) / h - (2.0 * c[(size_t)i] + c[(size_t)i + 1]) * h / 3.0;
            }
            const double h = x[(size_t)n - 1] - x[(size_t)n - 2];
            b[(size_t)n - 1] = 3.0 * d[(size_t)n - 2] * h * h + 2.0 * c[(size_t)n - 2] * h + b[(size_t)n - 2];
        } else {
            for (int i = 1; i < n - 1; i++) {
                const double h = x[(size_t)i + 1] - x[(size_t)i], hl = x[(size_t)i] - x[(size_t)i - 1];
                b[(size_t)i] = -h / (hl * (hl + h)) * y[(size_t)i - 1] + (h - hl) / (hl * h) * y[(size_t)i] + hl / (h * (hl + h)) * y[(size_t)i + 1];
            }
            b[0] = 0.5 * (-b[1] + 3.0 * (y[1] - y[0]) / (x[1] - x[0]));
            b[(size_t)n - 1] = 0.5 * (-b[(size_t)n - 2] + 3.0 * (y[(size_t)n - 1] - y[(size_t)n - 2]) / (x[(size_t)n - 1] - x[(size_t)n - 2]));
            for (int i = 0; i < n - 1; i++) {
                const double h = x[(size_t)i + 1] - x[(size_t)i];
                c[(size_t)i] = (3.0 * (y[(size_t)i + 1] - y[(size_t)i]) / h - (2.0 * b[(size_t)i] + b[(size_t)i + 1])) / h;
                d[(size_t)i] = ((b[(size_t)i + 1] - b[(size_t)i]) / (3.0 * h) - 2.0 / 3.0 * c[(size_t)i]) / h;
            }
        }
        c0 = c[0];
        return true;
    }
    double operator()(double v) const {
        const size_t n = x.size();
        size_t idx = 0;
        while (idx + 1 < n && x[idx + 1] <= v) idx++;
        const double h = v - x[idx];
        if (v < x[0]) return (c0 * h + b[0]) * h + y[0];
        if (v > x[n - 1]) return (c[n - 1] * h + b[n - 1]) * h + y[n - 1];
        return ((d[idx] * h + c[idx]) * h + b[idx]) * h + y[idx];
    }
};

struct Unpacked {  // host scalars of one vector
    int Nmax, lmax, Nfl0, Nfl1, Nfl2, Nfl3, Nnoise, onoise, ocfg, os, o1;
    bool do_amp;
    double g[6], trunc_c, model_type, bias_type, Hfactor, Wfactor, rot_env, rot_core, inclination, Vl[4], eta0, asym;
    int Nferr;
    std::vector<double> Wl0, Hl0;
    double fmin, fmax;
};

double app_width(const double g[6], double f) {  // models.cpp:4788-4794
    const double lnGamma0 = g[2] * std::log(f / g[0]) + std::log(g[3]);
    const double e = 2. * std::log(f / g[1]) / std::log(g[4] / g[0]);
    return std::exp(lnGamma0 + -std::log(g[5]) / (1. + std::pow(e, 2)));
}

// models.cpp:4727-4866 (id 25) / :4377-4470 (id 27, cte_width: one width parameter, Wl0 constant, :4407) + solver_mm.cpp:470-555 /
// :624-705 (everything before the pair loop)
int unpack(const double *p, const int32_t *pl, double step, bool cte_width, Unpacked &u, Prep &P) {
    const long double pi = M_PI;
    std::memset(&P, 0, sizeof P);
    u.Nmax = pl[0]; u.lmax = pl[1]; u.Nfl0 = pl[2]; u.Nfl1 = pl[3]; u.Nfl2 = pl[4]; u.Nfl3 = pl[5];
    const int Nsplit = pl[6], Nwidth = pl[7], Ninc = pl[9];
    u.Nnoise = pl[8];
    const int Nf = u.Nfl0 + u.Nfl1 + u.Nfl2 + u.Nfl3;
    u.os = u.Nmax + u.lmax + Nf;
    u.onoise = u.os + Nsplit + Nwidth;
    u.ocfg = u.onoise + u.Nnoise + Ninc;
    u.o1 = u.Nmax + u.lmax + u.Nfl0;
    u.trunc_c = p[u.ocfg];
    u.do_amp = p[u.ocfg + 1] != 0;
    u.model_type = p[u.ocfg + 3];
    u.bias_type = p[u.ocfg + 4];
    u.Nferr = (int)p[u.ocfg + 5];
    if (u.Nmax < 2 || u.Nmax != u.Nfl0 || u.Nferr < 0 || u.Nfl1 != 8 + 2 * u.Nferr || Nwidth < (cte_width ? 1 : 6) || Nsplit < 10 || u.lmax > 3)
        return TAMCMC_ERR_BAD_ARG;
    for (int k = 0; k < 6; k++) u.g[k] = k < (cte_width ? 1 : 6) ? std::fabs(p[u.os + Nsplit + k]) : 0.0;
    const double *fl0 = p + u.Nmax + u.lmax;
    u.Wl0.resize((size_t)u.Nmax); u.Hl0.resize((size_t)u.Nmax);
    for (int n = 0; n < u.Nmax; n++) u.Wl0[(size_t)n] = cte_width ? u.g[0] : app_width(u.g, fl0[n]);
    for (int n = 0; n < u.Nmax; n++)
        u.Hl0[(size_t)n] = u.do_amp ? (double)fabsl(p[n] * (1. / u.Wl0[(size_t)n] / pi)) : std::fabs(p[n]);
    const double delta0l = p[u.o1], DPl = std::fabs(p[u.o1 + 1]), alpha_g = std::fabs(p[u.o1 + 2]), q = std::fabs(p[u.o1 + 3]);
    u.Wfactor = std::fabs(p[u.o1 + 6]); u.Hfactor = std::fabs(p[u.o1 + 7]);
    u.rot_env = std::fabs(p[u.os]); u.rot_core = std::fabs(p[u.os + 1]);
    u.asym = p[u.os + 9];
    u.inclination = std::fabs(p[u.onoise + u.Nnoise]);
    u.Vl[0] = 1;
    for (int l = 1; l <= 3; l++) u.Vl[l] = l <= u.lmax ? std::fabs(p[u.Nmax + l - 1]) : 0.0;
    u.eta0 = (p[u.os + 8] == 1) ? mt::eta0_fct(fl0, u.Nfl0) : 0.0;
    u.fmin = *std::min_element(fl0, fl0 + u.Nfl0);
    u.fmax = *std::max_element(fl0, fl0 + u.Nfl0);
    double fit[2];
    mt::linfit_index(fl0, u.Nfl0, fit);
    const double Dnu_p = fit[0];
    // the reference exits (models.cpp:4851-4857; id 27 only tests it for model_type 0, :4459, and would otherwise hand its solver a zero
    // lower bound, i.e. an unbounded g-mode count: refused here too)
    if (!(Dnu_p > 0) || u.fmin - Dnu_p < 0) return TAMCMC_ERR_BAD_ARG;
    P.Dnu_p = Dnu_p; P.DPl = DPl; P.alpha = alpha_g; P.q = q; P.resol = step; P.fact = 0.04;
    double fmin_s, fmax_s;
    if (u.model_type == 0) {  // solve_mm_asymptotic_O2p(Dnu_p, eps, 1, delta0l, 0, 0, ...), fmin - Dnu .. fmax + Dnu
        const int n0 = (int)std::floor(fit[1] / Dnu_p);
        const double eps = fit[1] / Dnu_p - n0;
        fmin_s = u.fmin - Dnu_p; fmax_s = u.fmax + Dnu_p;
        const int el = 1;
        int np_min = (int)std::floor(fmin_s / Dnu_p - eps - el / 2 - delta0l);  // el/2: integer division, as in the reference
        int np_max = (int)std::ceil(fmax_s / Dnu_p - eps - el / 2 - delta0l);
        int ng_min = (int)std::floor(1e6 / (fmax_s * DPl) - alpha_g), ng_max = (int)std::ceil(1e6 / (fmin_s * DPl) - alpha_g);
        if (ng_min <= 0 && ng_max < 1) { P.Lp = 0; return TAMCMC_OK; }  // "impossible star": no mixed modes, the model carries on (solver_mm.cpp:497-501)
        if (ng_min <= 0 && ng_max >= 1) ng_min = 1;
        P.zone = (ng_max - ng_min < 6) ? (double)np_max : 1.75;
        if (np_min <= 0) np_min = 1;
        P.Lp = np_max - np_min; P.Lg = ng_max - ng_min; P.ng_min = ng_min;
        if (P.Lg < 1) { P.Lp = 0; return TAMCMC_OK; }  // no g mode in range
        if (P.Lp < 1 || P.Lp > MAXP) return TAMCMC_ERR_BAD_ARG;
        for (int np = np_min; np < np_max; np++) {
            P.nu_p[np - np_min] = (double)((np + (long double)eps + el / 2.L + delta0l) * Dnu_p);
            P.dnu_loc[np - np_min] = Dnu_p;  // alpha_p = 0
        }
        P.keep_lo = fmin_s; P.keep_hi = fmax_s;
    } else {  // solve_mm_asymptotic_O2from_l0(fl0, 1, delta0l, ...): the l=0 ladder shifted, three extra orders on each side
        fmin_s = u.fmin - Dnu_p; fmax_s = u.fmax + Dnu_p;
        if (fmin_s < 0) fmin_s = 0;
        int ng_min = (int)std::floor(1e6 / (fmax_s * DPl) - alpha_g), ng_max = (int)std::ceil(1e6 / (fmin_s * DPl) - alpha_g);
        if (ng_min <= 0 && ng_max < 1) { P.Lp = 0; return TAMCMC_OK; }
        if (ng_min <= 0 && ng_max >= 1) ng_min = 1;
        P.zone = (ng_max - ng_min < 6) ? 20. : 1.75;
        std::vector<double> ext;
        ext.push_back(u.fmin - 3 * Dnu_p); ext.push_back(u.fmin - 2 * Dnu_p); ext.push_back(u.fmin - Dnu_p);
        for (int k = 0; k < u.Nfl0; k++) ext.push_back(fl0[k]);
        ext.push_back(u.fmax + Dnu_p); ext.push_back(u.fmax + 2 * Dnu_p); ext.push_back(u.fmax + 3 * Dnu_p);
        int Lp = 0;
        for (double e : ext) {
            const double v = e + (double)(1 / 2.L * Dnu_p + delta0l);
            if (v >= fmin_s && v <= fmax_s) {
                if (Lp >= MAXP) return TAMCMC_ERR_BAD_ARG;
                P.nu_p[Lp++] = v;
            }
        }
        P.Lp = Lp; P.Lg = ng_max - ng_min; P.ng_min = ng_min;
        if (P.Lg < 1) { P.Lp = 0; return TAMCMC_OK; }
        if (P.Lp < 2) return TAMCMC_ERR_BAD_ARG;
        P.keep_lo = u.fmin; P.keep_hi = u.fmax;
    }
    if (fmin_s <= 150) P.fact = 0.01;
    if (fmin_s <= 50) P.fact = 0.005;
    // first derivative of the p ladder on the index grid (derivatives_handler.cpp:425-457)
    for (int i = 0; i < P.Lp; i++) {
        if (P.Lp == 1) P.dnup[i] = 0;
        else if (i == 0) P.dnup[i] = P.nu_p[1] - P.nu_p[0];
        else if (i == P.Lp - 1) P.dnup[i] = P.nu_p[i] - P.nu_p[i - 1];
        else P.dnup[i] = (P.nu_p[i + 1] - P.nu_p[i - 1]) / 2.;
    }
    if (u.model_type != 0)
        for (int i = 0; i < P.Lp; i++) P.dnu_loc[i] = P.dnup[i];  // the from-l0 driver hands the local derivative to the solver (:717)
    for (int ip = 0; ip < P.Lp; ip++) {
        P.ig0[ip] = -1;
        const double lo = P.nu_p[ip] - P.zone * Dnu_p, hi = P.nu_p[ip] + P.zone * Dnu_p;
        for (int ig = 0; ig < P.Lg; ig++) {
            const double g = nu_g_of(P, ig);
            if (g >= lo && g <= hi) { P.ig0[ip] = ig; break; }
        }
    }
    return TAMCMC_OK;
}

}  // namespace
}  // namespace rgb

// Builds the B tables of model 25 / 27 in the DEVICE staging block c->d_stage (layout StageLayout(B, stride, B*per), the one run_staged
// launches on); only the small header (counts, noise rows) goes through the host block.  One stream synchronisation (for the
// per-vector status); no table data crosses PCIe.
int rgb_stage_params(tamcmc_hip_ctx *c, int model_id, int B, const double *params, int64_t Nparams, const int32_t *plength, int32_t *status,
                     int *per_out, int *stride_out, int *first_err, int *tile_rot_out) {
    using namespace rgb;
    const bool cte_width = (model_id == TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4_ID);
    const bool dense_scan = c->armm_dense != 0;  // TAMCMC_OPT_ARMM_DENSE_SCAN
    const double *hx = c->hx.data();
    const int64_t Nx = c->Nx;
    const double step = hx[2] - hx[1];  // models.cpp:4719
    const int stride = plength[8] > 0 ? plength[8] : 1;
    if ((stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    if (plength[2] > MAXL || plength[4] > MAXL || plength[5] > MAXL) return TAMCMC_ERR_BAD_ARG;
    std::vector<Unpacked> U((size_t)B);
    // Prep and RowIn arrays are filled in ONE pinned block (a single asynchronous upload); the per-vector status words come back
    // into the same block and are read by rgb_collect_status after the caller's final synchronisation
    const size_t bytes_prep = ((size_t)B * sizeof(Prep) + 15) & ~(size_t)15, bytes_rows = ((size_t)B * sizeof(RowIn) + 15) & ~(size_t)15;
    HIPCHK(c, c->h_rgb.reserve(bytes_prep + bytes_rows + (size_t)B * sizeof(int)));
    Prep *P = (Prep *)c->h_rgb.p;
    RowIn *R = (RowIn *)(c->h_rgb.p + bytes_prep);
    const int per = plength[2] + plength[4] + plength[5] + CAP1;
    const StageLayout L(B, stride, (size_t)B * per);
    HIPCHK(c, c->h_stage.reserve(L.off_mults));
    HIPCHK(c, c->d_stage.reserve(L.bytes));
    unsigned char *h = c->h_stage.p;
    int32_t *h_nh = (int32_t *)(h + L.off_nh), *h_nn = (int32_t *)(h + L.off_nn);
    double *h_noise = (double *)(h + L.off_noise);
    int nthr = omp_get_max_threads();  // the scalar unpack of a vector (width law, fits, spline coefficients) is independent of the others
    if (nthr > 8) nthr = 8;
    if (nthr > B / 4) nthr = B / 4 > 0 ? B / 4 : 1;
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int b = 0; b < B; b++) {
        const double *p = params + (size_t)b * Nparams;
        Unpacked &u = U[(size_t)b];
        RowIn &ri = R[b];
        std::memset(&ri, 0, sizeof ri);
        int st = unpack(p, plength, step, cte_width, u, P[b]);
        if (st == TAMCMC_OK && u.bias_type != 0) {
            Spline s;
            if (u.Nferr > MAXNODE || !s.set(p + u.o1 + 8, p + u.o1 + 8 + u.Nferr, u.Nferr, u.bias_type == 1 ? 1 : 2)) st = TAMCMC_ERR_BAD_ARG;
            else {
                ri.bias_n = u.Nferr;
                for (int i = 0; i < u.Nferr; i++) {
                    ri.sx[i] = s.x[(size_t)i]; ri.sy[i] = s.y[(size_t)i]; ri.sb[i] = s.b[(size_t)i]; ri.sc[i] = s.c[(size_t)i]; ri.sd[i] = s.d[(size_t)i];
                }
                ri.sc0 = s.c0;
            }
        }
        status[b] = st;
        P[b].status = st;
        P[b].probe_dense = dense_scan ? 1 : 0;
        ri.status = st;
        if (st == TAMCMC_OK) {
            ri.Nfl0 = u.Nfl0; ri.Nfl2 = u.Nfl2; ri.Nfl3 = u.Nfl3; ri.lmax = u.lmax; ri.do_amp = u.do_amp ? 1 : 0; ri.cte_width = cte_width ? 1 : 0;
            const double *fl0 = p + u.Nmax + u.lmax;
            for (int k = 0; k < u.Nfl0; k++) { ri.fl0[k] = fl0[k]; ri.Wl0[k] = u.Wl0[(size_t)k]; ri.Hl0[k] = u.Hl0[(size_t)k]; }
            for (int k = 0; k < u.Nfl2; k++) ri.fl2[k] = std::fabs(p[u.o1 + u.Nfl1 + k]);
            for (int k = 0; k < u.Nfl3; k++) ri.fl3[k] = std::fabs(p[u.o1 + u.Nfl1 + u.Nfl2 + k]);
            for (int k = 0; k < 6; k++) ri.g[k] = u.g[k];
            for (int l = 0; l < 4; l++) ri.Vl[l] = u.Vl[l];
            ri.V[0][0] = 1.0;
            for (int l = 1; l <= u.lmax; l++) mt::amplitude_ratio(l, u.inclination, ri.V[l]);
            ri.eta0 = u.eta0; ri.asym = u.asym; ri.trunc_c = u.trunc_c; ri.Hfactor = u.Hfactor; ri.Wfactor = u.Wfactor;
            ri.rot_env = u.rot_env; ri.rot_core = u.rot_core;
            ri.a2 = p[u.os + 2]; ri.a3 = p[u.os + 4]; ri.a4 = p[u.os + 5]; ri.a5 = p[u.os + 6]; ri.a6 = p[u.os + 7];
            ri.fmin = u.fmin; ri.fmax = u.fmax;
            for (int k = 0; k < u.Nnoise; k++) h_noise[(size_t)b * stride + k] = std::fabs(p[u.onoise + k]);
            h_nh[b] = (u.Nnoise - 1) / 3; h_nn[b] = u.Nnoise;
        } else {
            h_noise[(size_t)b * stride] = 1.0;  // placeholder row; logL[b] is overwritten with NaN
            h_nh[b] = 0; h_nn[b] = 1;
        }
    }
    double fmin_all = 1e300;
    *first_err = TAMCMC_OK;
    for (int b = 0; b < B; b++) {
        if (status[b] == TAMCMC_OK && U[(size_t)b].fmin < fmin_all) fmin_all = U[(size_t)b].fmin;
        if (status[b] != TAMCMC_OK && *first_err == TAMCMC_OK) *first_err = status[b];
    }
    hipStream_t st = c->stream;
    // ---- device workspace
    const size_t nsolbuf = (size_t)B * MAXSOL;
    HIPCHK(c, c->d_rgb.reserve(bytes_prep + bytes_rows + nsolbuf * 3 * sizeof(double) + (size_t)B * (3 * sizeof(int) + sizeof(unsigned long long)) + 64));
    unsigned char *base = c->d_rgb.p;
    Prep *d_prep = (Prep *)base;
    RowIn *d_rows = (RowIn *)(base + bytes_prep);
    double *d_sols = (double *)(base + bytes_prep + bytes_rows);
    double *d_fl1 = d_sols + nsolbuf, *d_ksi = d_fl1 + nsolbuf;
    unsigned long long *d_norm = (unsigned long long *)(d_ksi + nsolbuf);
    int *d_nsol = (int *)(d_norm + B), *d_status = d_nsol + B;
    if (!c->poly_ready) {  // Pslm/Qlm tables in device memory (shared with the finite-difference builder)
        HIPCHK(c, c->d_poly.reserve(sizeof(mt::PolyTab)));
        hipLaunchKernelGGL(k_fill_poly_rgb, dim3(1), dim3(64), 0, st, (mt::PolyTab *)c->d_poly.p);
        c->poly_ready = true;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_stage.p, c->h_stage.p, L.off_mults, hipMemcpyHostToDevice, st));  // header only
    HIPCHK(c, hipMemcpyAsync(d_prep, P, bytes_prep + bytes_rows, hipMemcpyHostToDevice, st));  // Prep and RowIn arrays, contiguous on both sides
    HIPCHK(c, hipMemsetAsync(d_norm, 0, (size_t)B * (sizeof(unsigned long long) + 2 * sizeof(int)), st));
    hipLaunchKernelGGL(k_armm_scan, dim3(MAXP, B), dim3(64), 0, st, d_prep, d_sols, d_nsol);
    hipLaunchKernelGGL(k_armm_sort_unique, dim3(B), dim3(WG), 0, st, d_prep, d_rows, d_sols, d_nsol, d_fl1);
    const int chunks = 24;
    hipLaunchKernelGGL(k_zeta, dim3(chunks, B), dim3(WG), 0, st, d_prep, d_fl1, d_nsol, d_ksi, d_norm, chunks);
    hipLaunchKernelGGL(k_rgb_rows, dim3(B), dim3(WG), 0, st, d_prep, d_rows, (const mt::PolyTab *)c->d_poly.p, d_fl1, d_nsol, d_ksi, d_norm, hx[0],
                       hx[Nx - 1], (long)Nx, step, per, (tamcmc_multiplet *)(c->d_stage.p + L.off_mults), (int *)(c->d_stage.p + L.off_pairs), d_status);
    HIPCHK(c, hipGetLastError());
    // the device-side status words (row builder, solver overflow) travel back behind the kernels; no synchronisation here: the
    // likelihood launch that follows runs on whatever rows were written (a failed vector has an empty range) and the caller reads
    // the words after ITS synchronisation (rgb_collect_status)
    HIPCHK(c, hipMemcpyAsync(c->h_rgb.p + bytes_prep + bytes_rows, d_status, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    *per_out = per;
    *stride_out = stride;
    const int tb = tile_bins(c->wgs, c->K), ntiles = (int)((Nx + tb - 1) / tb);
    double t = (fmin_all < 1e299) ? (fmin_all - hx[0]) / (hx[1] - hx[0]) / (double)tb - 3.0 : 0.0;  // first near-field tile: a little below the lowest radial mode
    *tile_rot_out = (t > 0 && t < ntiles) ? (int)t : 0;
    return TAMCMC_OK;
}

// After rgb_stage_params(B) and a stream synchronisation: vector b's mixed modes as the pre-step left them in the workspace --
// frequencies (spline bias included) and the normalised zeta function, clipped at 1 like ksi_fct2_precise (bump_DP.cpp:180-186).
int rgb_fetch_modes(tamcmc_hip_ctx *c, int B, int b, int max_modes, double *nu_m, double *zeta, int *n_out) {
    using namespace rgb;
    const size_t bytes_prep = ((size_t)B * sizeof(Prep) + 15) & ~(size_t)15, bytes_rows = ((size_t)B * sizeof(RowIn) + 15) & ~(size_t)15;
    const size_t nsolbuf = (size_t)B * MAXSOL;
    unsigned char *base = c->d_rgb.p;
    const double *d_sols = (const double *)(base + bytes_prep + bytes_rows);
    const double *d_fl1 = d_sols + nsolbuf, *d_ksi = d_fl1 + nsolbuf;
    const unsigned long long *d_norm = (const unsigned long long *)(d_ksi + nsolbuf);
    const int *d_nsol = (const int *)(d_norm + B);
    int n = 0;
    unsigned long long nb = 0;
    HIPCHK(c, hipMemcpy(&n, d_nsol + b, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&nb, d_norm + b, sizeof nb, hipMemcpyDeviceToHost));
    if (n < 0 || n > MAXSOL) return TAMCMC_ERR_BAD_ARG;
    *n_out = n;
    if (n > max_modes) n = max_modes;
    if (n > 0 && nu_m) HIPCHK(c, hipMemcpy(nu_m, d_fl1 + (size_t)b * MAXSOL, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (n > 0 && zeta) {
        HIPCHK(c, hipMemcpy(zeta, d_ksi + (size_t)b * MAXSOL, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        double norm;
        std::memcpy(&norm, &nb, sizeof norm);
        for (int i = 0; i < n; i++) { zeta[i] = zeta[i] / norm; if (zeta[i] > 1) zeta[i] = 1; }
    }
    return TAMCMC_OK;
}

// After the stream has been synchronised: merges the device-side status words into status[] / first_err.
void rgb_collect_status(tamcmc_hip_ctx *c, int B, int32_t *status, int *first_err) {
    using namespace rgb;
    const size_t bytes_prep = ((size_t)B * sizeof(Prep) + 15) & ~(size_t)15, bytes_rows = ((size_t)B * sizeof(RowIn) + 15) & ~(size_t)15;
    const int *dst = (const int *)(c->h_rgb.p + bytes_prep + bytes_rows);
    for (int b = 0; b < B; b++) {
        if (status[b] == TAMCMC_OK) status[b] = dst[b];
        if (status[b] != TAMCMC_OK && *first_err == TAMCMC_OK) *first_err = status[b];
    }
}

}  // namespace tamcmc
