// dev_unpack.h -- device-side (workgroup-cooperative) log-prior and params -> multiplet-table unpack,
// shared by the device-resident sampler (dev_sampler.hip) and the finite-difference batch builder (capi.hip).
// The arithmetic itself is mode_tables_impl.h / priors_impl.h (written once for host and device); this file only
// spreads it over the lanes of one workgroup: prior terms one per lane (tree-summed), hard constraints sliced over the
// lanes, one lane per Wigner term / element, one lane per multiplet.
#pragma once
#include <hip/hip_runtime.h>

#include "bg_series.h"
#include "mode_tables_impl.h"
#include "priors_impl.h"

namespace tamcmc {

struct ModelDesc {  // constant description of one star's model (device pointers)
    int model_id, prior_class, Np, per, stride, Nx;
    double x_first, x_last, step;
    const int *plength, *priors_switch;
    const double *priors, *extra;
    const void *poly;  // mt::PolyTab in device memory
};

struct TablePtrs {  // the likelihood kernel's input block
    tamcmc_multiplet *mults;
    int *pairs, *nh, *nn;
    double *noise;
    double *bg = nullptr;      // [slots x ntiles x 8] background series per (slot, tile) for the FAST far field, or nullptr
    int ntiles = 0, tile_bins = 0;
};

// LDS scratch of the cooperative routines: 8 + 40 doubles, a PolyTab, a Shared, 4 ints/doubles
struct UnpackLds {
    double *red;        // [8]
    double *w;          // [40] Wigner terms (0..27) and elements (28..39)
    mt::PolyTab *poly;
    mt::Shared *S;
    int *status, *reject;
    double *dnu;
};
__host__ __device__ inline size_t unpack_lds_bytes() {
    return (8 + 40) * sizeof(double) + sizeof(mt::PolyTab) + sizeof(mt::Shared) + 64;
}
__device__ inline UnpackLds carve_unpack_lds(unsigned char *p) {
    UnpackLds u;
    u.red = (double *)p;
    u.w = u.red + 8;
    u.poly = (mt::PolyTab *)(u.w + 40);
    u.S = (mt::Shared *)(u.poly + 1);
    u.dnu = (double *)(((uintptr_t)(u.S + 1) + 7) & ~(uintptr_t)7);
    u.status = (int *)(u.dnu + 1);
    u.reject = u.status + 1;
    return u;
}

__device__ __forceinline__ double wg_sum(double v, double *s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double s = s_red[0];
    for (int w = 1; w < nw; w++) s = s + s_red[w];
    __syncthreads();
    return s;
}

// Call once per workgroup before the routines below (ends with a barrier).
__device__ inline void unpack_begin(const ModelDesc &d, const UnpackLds &u) {
    const double *src = (const double *)d.poly;
    double *dst = (double *)u.poly;
    for (int i = threadIdx.x; i < (int)(sizeof(mt::PolyTab) / sizeof(double)); i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) { *u.status = TAMCMC_OK; *u.reject = 0; }
    __syncthreads();
}

// m-visibilities (function_rot.cpp): one lane per TERM of each Wigner sum d^l_{i,0}(beta), i=0..l, and of the centre
// elements d^l_{0,0}(-beta); slot layout: for l=1..3, for i=0..l, then the centre (i=0, -beta): terms s=0..l-i (28 slots, 12 elements).
// Which (l, i, s) a lane owns and where an element's terms begin are compile-time tables packed into 64-bit constants (two or five
// bits per entry): no search loop, no memory access -- this stage sits on the longest dependent chain of the sampler's fused step.
// `lane` = index within the cooperating lanes (>= 28 of them), `sync` = their barrier: __syncthreads() for a whole
// workgroup, a wavefront fence when ONE wave does the stage beside the others (LDS operations of a wave complete in order).
namespace vis {
struct Pack {
    unsigned long long l, i, s, neg;  // per term slot: degree, row, term index (2 bits each); the centre's -beta (1 bit)
    unsigned long long first, el, ei; // per element: first term slot (5 bits), degree, row (2 bits each)
};
constexpr Pack make_pack() {
    Pack p{0, 0, 0, 0, 0, 0, 0};
    int sl = 0, el = 0;
    for (int l = 1; l <= 3; l++)
        for (int e = 0; e <= l + 1; e++) {
            const int i = (e <= l) ? e : 0;
            p.first |= (unsigned long long)sl << (5 * el);
            p.el |= (unsigned long long)l << (2 * el);
            p.ei |= (unsigned long long)i << (2 * el);
            el++;
            for (int s = 0; s <= l - i; s++, sl++) {
                p.l |= (unsigned long long)l << (2 * sl);
                p.i |= (unsigned long long)i << (2 * sl);
                p.s |= (unsigned long long)s << (2 * sl);
                if (e > l) p.neg |= 1ull << sl;
            }
        }
    return p;
}
}  // namespace vis
template <class Sync>
__device__ inline void visibilities_stage(const UnpackLds &u, int lane, Sync sync) {
    mt::Shared *S = u.S;
    constexpr vis::Pack P = vis::make_pack();
    if (lane < 28) {
        const double PI = 3.141592653589793238462643;
        const double ang = PI * S->inc / 180.;
        const int my_l = (int)((P.l >> (2 * lane)) & 3), my_i = (int)((P.i >> (2 * lane)) & 3), my_s = (int)((P.s >> (2 * lane)) & 3);
        const double my_b = ((P.neg >> lane) & 1) ? -ang : ang;
        if (S->need_ratio[my_l]) u.w[lane] = mt::wigner_term(my_l, my_i, 0, my_b, my_s);
    }
    sync();
    if (lane < 12) {  // one lane per ELEMENT: sum its terms in order, normalise (dmm's tail)
        const int l = (int)((P.el >> (2 * lane)) & 3), i = (int)((P.ei >> (2 * lane)) & 3), sl = (int)((P.first >> (5 * lane)) & 31);
        if (S->need_ratio[l]) {
            double sum = 0;
            for (int s = 0; s <= l - i; s++) sum = sum + u.w[sl + s];
            u.w[28 + lane] = mt::wigner_finish(l, i, 0, sum);
        }
    }
    sync();
    if (lane >= 1 && lane <= 3 && S->need_ratio[lane]) {  // mirror, centre overwrite, square (function_rot.cpp:25-41)
        const int l = lane, base = 28 + (l == 1 ? 0 : (l == 2 ? 3 : 7));
        double *V = S->ratios[l];
        for (int i = 0; i <= l; i++) V[l + i] = u.w[base + i];
        for (int i = -l; i <= 0; i++) V[l + i] = V[l - i] * mt::pow_m1(i);
        V[l] = u.w[base + l + 1] * mt::pow_m1(0);
        for (int i = 0; i <= 2 * l; i++) V[i] = V[i] * V[i];
    }
    sync();
}
struct WgSync { __device__ void operator()() const { __syncthreads(); } };
struct WaveSync {
    __device__ void operator()() const {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// Background series of the tiles [t_lo, t_hi) of evaluation slot `slot` (bg_series.h), strided over lanes first .. first+nw-1.
__device__ inline void wg_bg_tiles(const ModelDesc &d, const double *s_params, const mt::Shared *S, int slot, const TablePtrs &T, int first,
                                   int nw, int t_lo = 0, int t_hi = 1 << 30) {
    const int tid = threadIdx.x;
    const double *np_ = s_params + S->L.o_noise;
    const int nh = S->nharvey, nn = S->L.Nnoise;
    if (t_hi > T.ntiles) t_hi = T.ntiles;
    if (tid >= first && tid < first + nw)
        for (int t = t_lo + tid - first; t < t_hi; t += nw) {
            double xc, h;
            bg::tile_geometry(t, T.tile_bins, d.x_first, d.step, xc, h);
            if (!bg::series_valid(xc, h)) continue;
            double o[bg::NH];
            bg::tile_series([np_](int i) { return fabs(np_[i]); }, nh, nn, xc, h, o);
            double *dst = T.bg + ((size_t)slot * T.ntiles + t) * bg::NH;
#pragma unroll
            for (int k = 0; k < bg::NH; k++) dst[k] = o[k];
        }
}

// log-prior of the parameter vector in LDS (call_prior, model_def.cpp:421-464): returns the same value in every lane.
// While the additive terms are summed, the LAST lane prepares the unpack's shared scalars (different wave: overlaps).
// vis_in_prior: the LAST wave leaves the prior to the others and prepares the whole unpack instead (shared scalars AND the
// m-visibilities): pass vis_done = true to wg_unpack afterwards.
// early_rows (with vis_in_prior, 256 threads): a third group of lanes writes the table rows of slot `early_slot` while the
// visibilities are still being computed, with hv = H; wg_unpack(rows_done = true) multiplies the visibilities in.
__device__ inline double wg_log_prior(const ModelDesc &d, const double *s_params, const UnpackLds &u, bool prepare_unpack,
                                      bool vis_in_prior = false, const TablePtrs *early_rows = nullptr, int early_slot = 0) {
    const int Np = d.Np;
    const bool split = vis_in_prior && prepare_unpack && blockDim.x >= 128 && (blockDim.x & 63) == 0;
    const bool rows_early = split && early_rows && blockDim.x >= 256;
    const int tid = threadIdx.x, nt = split ? (int)blockDim.x - 64 : (int)blockDim.x;  // nt = lanes working on the prior
    if (tid >= nt) {  // wave-uniform: the helper wave, part 1 (beside the hard constraints)
        if (tid == nt) mt::shared_scalars_base(d.model_id, s_params, d.plength, *u.S);
    } else {
        int st = TAMCMC_OK;
        mt::xreal c = 0;
        if (d.prior_class == 2) {
            c = pr::ms_global_constraints(s_params, d.plength, d.priors_switch, d.extra, &st, tid, nt);
            if (tid == 1) {
                double fit[2];
                mt::linfit_index(s_params + d.plength[0] + d.plength[1], d.plength[2], fit);
                *u.dnu = fit[0];
            }
        } else if (d.prior_class == 3) {
            if (tid == 0) c = pr::local_constraints(s_params, d.plength, d.priors_switch, d.extra);
        } else {
            c = pr::neg_inf();
            st = TAMCMC_ERR_BAD_MODEL;
        }
        if (c != 0) *u.reject = 1;
        if (st != TAMCMC_OK) *u.status = st;
    }
    __syncthreads();
    const int n_extra = (d.prior_class == 2) ? pr::ms_global_extra_terms(d.plength, d.extra) : 0;
    double f = 0;
    int st = TAMCMC_OK;
    const int ntp = rows_early ? nt - 64 : nt;  // lanes on the additive terms
    if (tid >= nt) visibilities_stage(u, tid - nt, WaveSync());  // helper wave, part 2 (beside the additive terms)
    else if (tid >= ntp) {  // the wave before it: table rows that do not need the visibilities yet
        for (int idx = tid - ntp; idx < d.per; idx += 64) {
            const int rs = mt::build_multiplet(d.model_id, *u.poly, s_params, *u.S, idx, d.x_first, d.x_last, d.Nx, d.step,
                                               &early_rows->mults[(size_t)early_slot * d.per + idx], true);
            if (rs) st = rs;
        }
    } else
        for (int t = tid; t < Np + n_extra; t += ntp) {
            if (t < Np) f = f + pr::generic_prior_term(s_params, Np, d.priors, d.priors_switch, t, &st);
            else f = f + pr::ms_global_extra_term(s_params, d.plength, d.extra, *u.dnu, t - Np);
        }
    if (rows_early && early_rows->bg) wg_bg_tiles(d, s_params, u.S, early_slot, *early_rows, 0, ntp);  // the term lanes, once done
    if (st != TAMCMC_OK) *u.status = st;
    if (prepare_unpack && !split && tid == nt - 1) mt::shared_scalars_base(d.model_id, s_params, d.plength, *u.S);
    f = wg_sum(f, u.red);
    return *u.reject ? -INFINITY : f;
}

#ifdef TAMCMC_PROBE
#define UPSTAMP(k) do { if (pst) pst[k] = (long)wall_clock64(); } while (0)
#else
#define UPSTAMP(k)
#endif
// The same log-prior by single waves (blockDim.x == 64), bit for bit the value wg_log_prior returns in the proposal kernel's 256-thread
// layout: there `virt` (= 128) lanes take the additive terms (term t on lane t mod virt, in increasing t), each wave of 64 lanes sums
// its lanes with a shuffle tree and the waves are added in order.  Here ONE wave plays the lanes [64 h, 64 h + 64) of that layout and
// returns their sum; the caller adds the halves in order (h = 0, 1, ..).  Half 0 also checks the hard constraints (*rejected: the
// log-prior is -inf whatever the sums).  Two waves instead of one that plays both halves: each is on the fused step's longest chain.
__device__ inline double wave_log_prior_part(const ModelDesc &d, const double *s_params, const UnpackLds &u, int virt, int h, int *rejected,
                                             long *pst = nullptr) {
    const int Np = d.Np, tid = threadIdx.x;
    const int n_extra = (d.prior_class == 2) ? pr::ms_global_extra_terms(d.plength, d.extra) : 0;
    UPSTAMP(0);
    bool need_dnu = false;  // does this half hold a term that needs the large separation?  (virt is a power of two: 128)
    for (int t = Np; t < Np + n_extra; t++) need_dnu = need_dnu || (((t & (virt - 1)) >> 6) == h);
    {
        int st = TAMCMC_OK;
        mt::xreal c = 0;
        if (d.prior_class == 2) {
            if (h == 0) c = pr::ms_global_constraints(s_params, d.plength, d.priors_switch, d.extra, &st, tid, 64);
            if (need_dnu && tid == (h == 0 ? 1 : 0)) {
                double fit[2];
                mt::linfit_index(s_params + d.plength[0] + d.plength[1], d.plength[2], fit);
                *u.dnu = fit[0];
            }
        } else if (d.prior_class == 3) {
            if (h == 0 && tid == 0) c = pr::local_constraints(s_params, d.plength, d.priors_switch, d.extra);
        } else {
            c = pr::neg_inf();
            st = TAMCMC_ERR_BAD_MODEL;
        }
        if (c != 0) *u.reject = 1;
        if (st != TAMCMC_OK) *u.status = st;
    }
    __syncthreads();
    UPSTAMP(1);
    int st = TAMCMC_OK;
    double f = 0;
    for (int t = h * 64 + tid; t < Np + n_extra; t += virt) {
        if (t < Np) f = f + pr::generic_prior_term(s_params, Np, d.priors, d.priors_switch, t, &st);
        else f = f + pr::ms_global_extra_term(s_params, d.plength, d.extra, *u.dnu, t - Np);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) f = f + __shfl_down(f, off, 64);
    f = __shfl(f, 0, 64);
    if (st != TAMCMC_OK) *u.status = st;
    __syncthreads();
    UPSTAMP(2);
    *rejected = *u.reject;
    return f;
}

// params (LDS) -> table rows of evaluation slot `slot` (+ noise row, range, counts).  `live` = the prior is finite
// (model_def.cpp:472,476-480 skips the model otherwise).  u.S must hold shared_scalars_base (wg_log_prior did it).
// empty_on_fail: a failed table leaves an EMPTY slot (nn = 0: the likelihood kernel skips it) instead of a placeholder noise row.
__device__ inline void wg_unpack(const ModelDesc &d, const double *s_params, const UnpackLds &u, int slot, const TablePtrs &T,
                                 bool live, bool vis_done = false, bool rows_done = false, bool empty_on_fail = false, long *pst = nullptr) {
    const int tid = threadIdx.x, nt = blockDim.x, per = d.per;
    mt::Shared *S = u.S;
    UPSTAMP(0);
    if (live) {
        if (!vis_done) visibilities_stage(u, tid, WgSync());  // workgroup-uniform
        UPSTAMP(1);
        if (rows_done) {  // rows already written with hv = H (wg_log_prior, early_rows): the visibilities are known now
            for (int e = tid; e < per * 7; e += nt) {
                tamcmc_multiplet *r = &T.mults[(size_t)slot * per + e / 7];
                const int k = e % 7, l = r->l;
                if (k < 2 * l + 1) r->hv[k] = r->hv[k] * S->ratios[l][k];
            }
        } else
            for (int idx = tid; idx < per; idx += nt) {  // rows go straight to the likelihood kernel's table
                const int st = mt::build_multiplet(d.model_id, *u.poly, s_params, *S, idx, d.x_first, d.x_last, d.Nx, d.step,
                                                   &T.mults[(size_t)slot * per + idx], false, pst);
                if (st) *u.status = st;
            }
        for (int i = tid; i < S->L.Nnoise; i += nt) T.noise[(size_t)slot * d.stride + i] = fabs(s_params[S->L.o_noise + i]);
        if (T.bg && !rows_done)  // (with early rows wg_log_prior's term lanes already did it)
            wg_bg_tiles(d, s_params, S, slot, T, (nt > 64) ? 64 : 0, (nt > 64) ? nt - 64 : nt);  // beside the first wave's multiplet rows
    }
    __syncthreads();
    UPSTAMP(2);
    if (tid == 0) {
        const bool ok = live && (*u.status == TAMCMC_OK);
        T.pairs[2 * slot] = slot * per;
        T.pairs[2 * slot + 1] = ok ? (slot + 1) * per : slot * per;
        T.nh[slot] = ok ? S->nharvey : 0;
        T.nn[slot] = ok ? S->L.Nnoise : (empty_on_fail ? 0 : 1);
        if (!ok) T.noise[(size_t)slot * d.stride] = 1.0;  // placeholder row; the caller rejects / NaNs the evaluation
    }
    if (T.bg && !(live && (*u.status == TAMCMC_OK)))  // ... with the matching background series (constant 1)
        for (int t = tid; t < T.ntiles; t += nt) {
            double *dst = T.bg + ((size_t)slot * T.ntiles + t) * bg::NH;
            for (int k = 0; k < bg::NH; k++) dst[k] = (k == 0) ? 1.0 : 0.0;
        }
}

}  // namespace tamcmc
