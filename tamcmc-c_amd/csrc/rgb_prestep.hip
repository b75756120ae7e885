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
#include "rgb_unpack.h"
#include "bg_series.h"
#include "kernels.h"

namespace tamcmc {
const mt::PolyTab &poly_table();  // mode_tables.cpp
namespace rgb {

constexpr int WG = 256;
constexpr int SOLVE_NT = 256;  // lanes per (vector, p mode) workgroup of the solver


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
__device__ void armm_scan_pair(const Prep *preps, double *sols, int *nsol) {
    const int b = blockIdx.y, ip = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = SOLVE_NT, NW = NT / 64;
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
    if (tid == 0) { s_nc = 0; s_nh = 0; }
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
        // two work items per unit, one LANE each: (odd) the cells around pole u-1, (even) the pole-free stretch of unit u
        for (long w = tid; w <= 2 * np + 1; w += NT) {
            const long u = w >> 1;
            if (w & 1) {
                if (u == 0) continue;
                const long j = u - 1, cj = pole_cell(j);
                const long prev_end = (j > 0) ? pole_cell(j - 1) + 1 : -1;  // last cell of the previous pole's zone
                long i = (cj - 1 > prev_end ? cj - 1 : prev_end + 1);
                const long i_end = cj + 1 < n - 2 ? cj + 1 : n - 2;
                if (i < 0) i = 0;
                if (i > i_end) continue;
                double fa = fg(i);  // each point of the run is evaluated once (the right end of a cell is the left end of the next)
                for (; i <= i_end; i++) {
                    const double fb = fg(i + 1);
                    if (changes_sign(fa, fb)) push(i);
                    fa = fb;
                }
                continue;
            }
            long a, e;  // the pole-free stretch of this unit: cells with left index a .. e
            if (u == 0) { a = 0; e = pole_cell(0) - 2; }
            else {
                const long j = u - 1;
                a = pole_cell(j) + 2;
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
        // of each wave evaluates that one point more)
        for (long i0 = (long)wave * 64; i0 < n - 1; i0 += NT) {
            const long i = i0 + lane;
            const double fa = (i < n) ? fg(i) : 0.0;
            double fb = __shfl_down(fa, 1, 64);
            if (lane == 63 && i + 1 < n) fb = fg(i + 1);
            if (i < n - 1 && changes_sign(fa, fb)) push(i);
        }
    }
    __syncthreads();
    if (s_nc > MAXC) {  // more sign changes than the candidate list holds: flag the vector (the host reports it), do not guess
        if (tid == 0) atomicAdd(&nsol[b], 2 * MAXSOL);
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
    // straight line through two local points evaluated at p-g = 0, then the ratio test (solver_mm.cpp:392-404).  (j, fa, fb): the
    // bracketing pair lin_interpol uses when it interpolates and the function there (values the search already holds)
    auto finish = [&](const Local &L, long j, double fa_j, double fb_j, double f_first, double f_last) {
        double a = 0, bb = 0;
        if (0.0 >= f_first && 0.0 <= f_last) {
            a = (lg(L, j + 1) - lg(L, j)) / (fb_j - fa_j);
            bb = lg(L, j) - a * fa_j;
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
    for (int c = tid; c < nc; c += NT) {
        const Local L = local_of(s_cand[c]);
        if (L.nl < 2) continue;
        const double f_first = fl(L, 0), f_last = fl(L, L.nl - 1);
        // first j with f[j] <= 0 <= f[j+1] -- only needed when lin_interpol interpolates (f_first <= 0 <= f_last); otherwise it
        // extrapolates from the first or last two points (a pole of tan(): half of all candidates) and no search is made
        long best = 0;
        double f_lo = f_first, f_hi = f_last;
        if (!(0.0 < f_first) && !(0.0 > f_last)) {
            const double ka = kap_scale * (1. / L.rmin - inv_g) - 0.5, kb = kap_scale * (1. / L.rmax - inv_g) - 0.5;
            const bool pole_inside = (floor(ka) != floor(kb)) || fabs(ka - rint(ka)) < 1e-9 || fabs(kb - rint(kb)) < 1e-9;
            if (pole_inside) {  // left to a whole wave below
                const int k = atomicAdd(&s_nh, 1);
                s_hard[k] = c;
                continue;
            }
            long lo_j = 0, hi_j = L.nl - 1;  // f[lo_j] <= 0 <= f[hi_j], p-g increasing: the last point with f <= 0
            while (hi_j - lo_j > 1) {
                const long mid = lo_j + (hi_j - lo_j) / 2;
                const double fm = fl(L, mid);
                if (fm <= 0.0) { lo_j = mid; f_lo = fm; } else { hi_j = mid; f_hi = fm; }
            }
            // step back over exact zeros so that the FIRST pair with f[j] <= 0 <= f[j+1] is the one used
            best = lo_j;
            while (best > 0 && f_lo == 0.0) {
                const double fp = fl(L, best - 1);
                if (!(fp <= 0.0)) break;
                best--; f_hi = f_lo; f_lo = fp;
            }
        }
        finish(L, best, f_lo, f_hi, f_first, f_last);
    }
    __syncthreads();
    const int nh = s_nh;
    for (int hc = wave; hc < nh; hc += NW) {  // wave-uniform: one wave per window, the reference's point-by-point walk
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
        if (lane == 0) {
            const long j = best < L.nl - 1 ? best : L.nl - 2;
            finish(L, j, fl(L, j), fl(L, j + 1), f_first, f_last);
        }
    }
}

// One workgroup per vector: its solutions sorted, then std::unique with |a-b| <= 2 resol (solver_mm.cpp:586-593).
// (part of k_rgb_finish; every lane of the workgroup runs through it).  The result is left in s[0 .. *s_m_p).
__device__ void sort_unique_bias(const int b, const Prep *preps, const RowIn *rows_in, double *sols, int *nsol, double *fl1, double *s /* LDS [MAXSOL] */,
                                 double *t /* LDS [MAXSOL] */, unsigned char *keep /* LDS [MAXSOL] */, int *s_m_p /* LDS */) {
    const int tid = threadIdx.x;
    int &s_m = *s_m_p;
    const int n_raw = nsol[b];
    const int n = (n_raw > MAXSOL || n_raw < 0) ? 0 : n_raw;  // overflow: the count stays as it is, the row builder refuses the vector
    for (int i = tid; i < n; i += WG) s[i] = sols[(size_t)b * MAXSOL + i];
    __syncthreads();                                          // (every lane has read nsol[b] before lane 0 rewrites it)
    // sort by rank: element i goes to the number of elements that sort before it (ties by index); LDS broadcast reads, no barrier per stage
    for (int i = tid; i < n; i += WG) {
        const double v = s[i];
        int r = 0;
        for (int j = 0; j < n; j++) {
            const double u = s[j];
            r += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        t[r] = v;
    }
    __syncthreads();
    // std::unique keeps an element unless it is within tol of the last KEPT one.  An element further than tol from its left neighbour
    // is always kept (the last kept one is not larger than that neighbour), so the chain restarts there: one lane per such run.
    const double tol = 2 * preps[b].resol;
    for (int i = tid; i < n; i += WG) {
        if (i > 0 && fabs(t[i - 1] - t[i]) <= tol) continue;  // not the head of a run
        double last = t[i];
        keep[i] = 1;
        for (int j = i + 1; j < n && fabs(t[j - 1] - t[j]) <= tol; j++) {
            const double v = t[j];
            if (!(fabs(last - v) <= tol)) { keep[j] = 1; last = v; } else keep[j] = 0;
        }
    }
    __syncthreads();
    {   // compaction: exclusive prefix sum of the keep flags, 4 consecutive elements per lane
        __shared__ int s_wsum[WG / 64];
        const int base = tid * 4;
        int k4[4], mine = 0;
        for (int q = 0; q < 4; q++) { k4[q] = (base + q < n) ? keep[base + q] : 0; mine += k4[q]; }
        int inc = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(inc, off, 64);
            if ((tid & 63) >= off) inc += up;
        }
        if ((tid & 63) == 63) s_wsum[tid >> 6] = inc;
        __syncthreads();
        int pos = inc - mine;
        for (int w = 0; w < (tid >> 6); w++) pos += s_wsum[w];
        for (int q = 0; q < 4; q++)
            if (k4[q]) s[pos++] = t[base + q];
        if (tid == WG - 1) {
            s_m = pos;
            if (n_raw == n) nsol[b] = pos;
        }
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
    __syncthreads();
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

// Maximum of the same sum over the high-resolution grid (the normalisation of zeta, bump_DP.cpp:125-188), chunk `chunk` of `chunks`
// by one workgroup of NT lanes; the maximum does not depend on how the grid is cut.
template <int NT>
__device__ void zeta_norm_chunk(const Prep &P, int chunk, int chunks, unsigned long long *norm_bits_b) {
    const int tid = threadIdx.x;
    if (P.status != 0 || P.Lp < 1 || P.Lg < 1) return;  // failed vector, or no g mode in range (no mixed modes); workgroup-uniform
    double pmin = P.nu_p[0], pmax = P.nu_p[0];
    for (int i = 1; i < P.Lp; i++) { pmin = fmin(pmin, P.nu_p[i]); pmax = fmax(pmax, P.nu_p[i]); }
    const double gmax = nu_g_of(P, 0), gmin = nu_g_of(P, P.Lg - 1);  // the g ladder decreases with n_g
    const double lo = pmin >= gmin ? gmin : pmin, hi = pmax >= gmax ? pmax : gmax;
    const double resol = 1e6 / (4 * 365. * 86400.);
    const long nh = (long)((hi - lo) / resol);
    double best = 0;
    if (nh >= 2) {
        const double step = (hi - lo) / (double)(nh - 1);
        for (long i = (long)chunk * NT + tid; i < nh; i += (long)chunks * NT) {
            const double v = ksi_sum(P, (i == nh - 1) ? hi : lo + (double)i * step);
            if (v > best) best = v;
        }
    }
    // one atomic per workgroup (atomics on one address serialise in L2)
    __shared__ double s_best[NT / 64];
    for (int off = 32; off >= 1; off >>= 1) best = fmax(best, __shfl_down(best, off, 64));
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; w++) best = fmax(best, s_best[w]);
        if (best > 0) atomicMax(norm_bits_b, (unsigned long long)__double_as_longlong(best));  // positive doubles order as integers
    }
}

// grid (MAXP + ZCHUNKS, B), SOLVE_NT lanes: blocks [0, MAXP) solve one p mode each (armm_scan_pair), the others take a chunk of the zeta
// normalisation grid -- it needs the vector's ladders only, not the roots, so it runs beside the solver instead of after it.
constexpr int ZCHUNKS = 16;
__global__ void __launch_bounds__(SOLVE_NT) k_armm_solve(const Prep *preps, double *sols, int *nsol, unsigned long long *norm_bits) {
    const int b = blockIdx.y;
    if ((int)blockIdx.x < MAXP) armm_scan_pair(preps, sols, nsol);
    else zeta_norm_chunk<SOLVE_NT>(preps[b], (int)blockIdx.x - MAXP, ZCHUNKS, norm_bits + b);
}

// One workgroup per vector: the table rows (l=0 list, mixed modes, l=2, l=3 lists) written into the likelihood kernel's input block,
// in the reference's accumulation order (models.cpp:4915-5000); mixed-mode scalars: bump_DP.cpp:203-254, :531-547.
struct BgOut {  // optional by-product of k_rgb_finish: the FAST far-field background series of the vector's tiles ([chain][tile][NH])
    double *out = nullptr;
    const double *noise = nullptr;
    const int *nh = nullptr, *nn = nullptr;
    int stride = 0, ntiles = 0, tile_bins = 0;
    double x0 = 0, step = 0;
};

// One workgroup per vector, after k_armm_solve: sort + unique of the roots and their spline bias, zeta at the modes, then the rows.
// b0: the engine-side arrays (mults, pairs, status) are indexed by b0 + b (a chain group of the device engine works on chains b0..),
// the workspace arrays by b.
__global__ void __launch_bounds__(WG) k_rgb_finish(const Prep *preps, const RowIn *rows_in, const mt::PolyTab *poly, double *sols, int *n1, double *fl1,
                                                   double *ksi, const unsigned long long *norm_bits, double x_first, double x_last, long Nx,
                                                   double step, int per, int b0, tamcmc_multiplet *mults, int *pairs, int *status, const BgOut bg) {
    const int b = blockIdx.x, tid = threadIdx.x, e = b0 + b;
    const RowIn &R = rows_in[b];
    if (bg.out) {  // the vector's background series per likelihood tile (k_bg_poly's arithmetic; the noise row is in place since the proposal kernel)
        const int nn = bg.nn[e];
        const double *nz = bg.noise + (size_t)e * bg.stride;
        for (int tile = tid; nn > 0 && tile < bg.ntiles; tile += WG) {
            double xc, h;
            bg::tile_geometry(tile, bg.tile_bins, bg.x0, bg.step, xc, h);
            if (!bg::series_valid(xc, h)) continue;
            double o[bg::NH];
            bg::tile_series([nz](int i) { return nz[i]; }, bg.nh[e], nn, xc, h, o);
            for (int k = 0; k < bg::NH; k++) bg.out[((size_t)e * bg.ntiles + tile) * bg::NH + k] = o[k];
        }
    }
    static_assert(MAXSOL <= 4 * WG, "the compaction handles 4 elements per lane");
    __shared__ double s_sol[MAXSOL], s_tmp[MAXSOL];
    __shared__ unsigned char s_keep[MAXSOL];
    __shared__ int s_m;
    sort_unique_bias(b, preps, rows_in, sols, n1, fl1, s_sol, s_tmp, s_keep, &s_m);
    {   // un-normalised zeta at the vector's modes
        const Prep &P = preps[b];
        if (!(P.status != 0 || P.Lp < 1 || P.Lg < 1))
            for (int i = tid; i < s_m; i += WG) ksi[(size_t)b * MAXSOL + i] = ksi_sum(P, fl1[(size_t)b * MAXSOL + i]);
    }
    __syncthreads();
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
        tamcmc_multiplet *r = &mults[(size_t)e * per + k];
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
        pairs[2 * e] = e * per;
        pairs[2 * e + 1] = good ? e * per + total : e * per;
        status[e] = s_st;
    }
}

__global__ void k_fill_poly_rgb(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
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
    std::vector<double> fmins((size_t)B, 1e300);
    int nthr = omp_get_max_threads();  // the scalar unpack of a vector (width law, fits, spline coefficients) is independent of the others
    if (nthr > 8) nthr = 8;
    if (nthr > B / 4) nthr = B / 4 > 0 ? B / 4 : 1;
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int b = 0; b < B; b++)
        status[b] = unpack_vector(OneThread(), params + (size_t)b * Nparams, plength, step, cte_width, dense_scan ? 1 : 0, P[b], R[b], h_noise + (size_t)b * stride,
                                  h_nh + b, h_nn + b, &fmins[(size_t)b]);
    double fmin_all = 1e300;
    *first_err = TAMCMC_OK;
    for (int b = 0; b < B; b++) {
        if (status[b] == TAMCMC_OK && fmins[(size_t)b] < fmin_all) fmin_all = fmins[(size_t)b];
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
    hipLaunchKernelGGL(k_armm_solve, dim3(MAXP + ZCHUNKS, B), dim3(SOLVE_NT), 0, st, d_prep, d_sols, d_nsol, d_norm);
    hipLaunchKernelGGL(k_rgb_finish, dim3(B), dim3(WG), 0, st, d_prep, d_rows, (const mt::PolyTab *)c->d_poly.p, d_sols, d_nsol, d_fl1, d_ksi, d_norm, hx[0],
                       hx[Nx - 1], (long)Nx, step, per, 0, (tamcmc_multiplet *)(c->d_stage.p + L.off_mults), (int *)(c->d_stage.p + L.off_pairs), d_status, BgOut());
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

// Device engine (dev_sampler.hip): the same pre-step on parameter vectors that are ALREADY in device memory -- nothing crosses PCIe and
// nothing is synchronised.  rgb_device_prepare() sizes the workspace once (outside the iteration loop); the sampler's proposal kernel
// runs the scalar unpack itself (rgb_unpack.h) into the slice rgb_device_slice() describes; rgb_device_stage() enqueues the solver and
// the sort / zeta / row kernel on `st`, writing the tables into the engine's own likelihood input block T.
static size_t rgb_slice_bytes(int B) {
    using namespace rgb;
    const size_t bytes_prep = ((size_t)B * sizeof(Prep) + 15) & ~(size_t)15, bytes_rows = ((size_t)B * sizeof(RowIn) + 15) & ~(size_t)15;
    return (bytes_prep + bytes_rows + (size_t)B * MAXSOL * 3 * sizeof(double) + (size_t)B * (3 * sizeof(int) + sizeof(unsigned long long)) + 255) & ~(size_t)255;
}

int rgb_device_prepare(tamcmc_hip_ctx *c, int Bmax, int slices, const int32_t *plength, int *per_out, int *stride_out) {
    using namespace rgb;
    const int stride = plength[8] > 0 ? plength[8] : 1;
    if ((stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    if (plength[2] > MAXL || plength[4] > MAXL || plength[5] > MAXL || plength[2] < 2) return TAMCMC_ERR_BAD_ARG;
    HIPCHK(c, c->d_rgb.reserve(rgb_slice_bytes(Bmax) * (size_t)slices));
    if (!c->poly_ready) {
        HIPCHK(c, c->d_poly.reserve(sizeof(mt::PolyTab)));
        hipLaunchKernelGGL(k_fill_poly_rgb, dim3(1), dim3(64), 0, c->stream, (mt::PolyTab *)c->d_poly.p);
        c->poly_ready = true;
    }
    *per_out = plength[2] + plength[4] + plength[5] + CAP1;
    *stride_out = stride;
    return TAMCMC_OK;
}

rgb::Slice rgb_device_slice(tamcmc_hip_ctx *c, int Bmax, int slice) {
    using namespace rgb;
    const size_t bytes_prep = ((size_t)Bmax * sizeof(Prep) + 15) & ~(size_t)15, bytes_rows = ((size_t)Bmax * sizeof(RowIn) + 15) & ~(size_t)15;
    const size_t nsolbuf = (size_t)Bmax * MAXSOL;
    unsigned char *base = c->d_rgb.p + rgb_slice_bytes(Bmax) * (size_t)slice;
    Slice S;
    S.preps = (Prep *)base;
    S.rows = (RowIn *)(base + bytes_prep);
    S.norm_bits = (unsigned long long *)((double *)(base + bytes_prep + bytes_rows) + 3 * nsolbuf);
    S.nsol = (int *)(S.norm_bits + Bmax);
    S.step = c->hx[2] - c->hx[1];
    S.dense = c->armm_dense ? 1 : 0;
    return S;
}

int rgb_device_stage(tamcmc_hip_ctx *c, int b0, int B, int Bmax, int slice, int per, const RgbDeviceTables &T, hipStream_t st) {
    using namespace rgb;
    const double *hx = c->hx.data();
    const int64_t Nx = c->Nx;
    const double step = hx[2] - hx[1];  // models.cpp:4719
    const Slice S = rgb_device_slice(c, Bmax, slice);
    double *d_sols = (double *)((unsigned char *)S.rows + (((size_t)Bmax * sizeof(RowIn) + 15) & ~(size_t)15));
    const size_t nsolbuf = (size_t)Bmax * MAXSOL;
    double *d_fl1 = d_sols + nsolbuf, *d_ksi = d_fl1 + nsolbuf;
    hipLaunchKernelGGL(k_armm_solve, dim3(MAXP + ZCHUNKS, B), dim3(SOLVE_NT), 0, st, S.preps, d_sols, S.nsol, S.norm_bits);
    BgOut bg;
    bg.out = T.bg; bg.noise = T.noise; bg.stride = T.stride; bg.nh = T.nh; bg.nn = T.nn; bg.ntiles = T.ntiles; bg.tile_bins = T.tile_bins;
    bg.x0 = hx[0]; bg.step = hx[1] - hx[0];
    hipLaunchKernelGGL(k_rgb_finish, dim3(B), dim3(WG), 0, st, S.preps, S.rows, (const mt::PolyTab *)c->d_poly.p, d_sols, S.nsol, d_fl1, d_ksi, S.norm_bits, hx[0],
                       hx[Nx - 1], (long)Nx, step, per, b0, T.mults, T.pairs, T.status, bg);
    HIPCHK(c, hipGetLastError());
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
