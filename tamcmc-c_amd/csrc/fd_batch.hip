// fd_batch.hip -- finite-difference gradient batches built ON the device.
//
// The forward-difference gradient (the drift the reference leaves as a stub, MALA.cpp:321-337) needs, per chain,
// Nvars+1 evaluations.  Building those tables on the host costs far more than evaluating them (C3: 1880 tables,
// ~20 us each, against ~4 ms of likelihood kernel); here one workgroup per (chain, perturbed variable) perturbs the
// parameter vector, evaluates the log-prior (and the prior at the backward point, for the one-sided fallback at the
// edge of a prior's support) and writes its multiplet table straight into the likelihood kernel's input block:
//   k_fd_unpack (C*(Nvars+1) workgroups) -> k_loglike (one launch, B = C*(Nvars+1)) -> k_finalize,
// or, windowed (FAST arithmetic: the default):
//   k_fd_unpack (tables + delta tables) -> k_loglike on the C base points (planes 1/M0, y/M0, M0 kept) -> k_fd_moments (tile moments of the
//   base points) -> k_fd_far (far-only tiles of the light evaluations from the moments) -> k_loglike<DELTA> (everything else) -> k_finalize.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"
#include "dev_unpack.h"
#include "kernels.h"
#include "fd_batch.h"

namespace tamcmc {
namespace {

constexpr int FB = 128;

__global__ void k_fill_poly_fd(mt::PolyTab *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mt::fill_poly(*t);
}

struct FdArgs {
    ModelDesc desc;
    TablePtrs T;
    int C, E, Nv;
    const double *params;  // [C x Np]
    const int *idx;        // [Nv] index_to_relax
    const double *h;       // [Nv] steps
    double *logPr_plus;    // [C x E] log-prior at the evaluation point (e=0: the base point)
    double *logPr_minus;   // [C x E] log-prior at theta - h e_k (e>=1)
    int *status;           // [C x E]
    // windowed mode (delta tables): rows [B*per, 2B*per) of T.mults hold each block's copy of the BASE table,
    // D is the delta launch's input block (2*per rows per evaluation: +new / -old of the changed multiplets)
    int windowed;
    TablePtrs D;
    int *d_range, *d_flags, *d_row;
    double *d_noise_old;
    TablePtrs Bs;          // base launch (C evaluations): ranges / counts / noise rows indexed by chain
    int full_tables;       // 1: "full table" delta evaluations allowed (the base launch's background series are at hand: FAST arithmetic)
};

__global__ void __launch_bounds__(FB) k_fd_unpack(const FdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int Np = a.desc.Np;
    double *s_params = (double *)s_raw;
    const UnpackLds U = carve_unpack_lds((unsigned char *)(s_params + Np));
    const int slot = blockIdx.x, c = slot / a.E, e = slot - c * a.E, tid = threadIdx.x;
    const int B = a.C * a.E, per = a.desc.per;
    __shared__ int s_lo, s_hi, s_nchg, s_noise_chg, s_base_status;
    for (int i = tid; i < Np; i += FB) s_params[i] = a.params[(size_t)c * Np + i];
    unpack_begin(a.desc, U);
    if (a.windowed && e > 0) {
        // this block's own copy of the BASE table (slot B+slot): the changed rows are found by comparing with it
        if (tid == FB - 1) mt::shared_scalars_base(a.desc.model_id, s_params, a.desc.plength, *U.S);
        __syncthreads();
        wg_unpack(a.desc, s_params, U, B + slot, a.T, true);
        __syncthreads();
        if (tid == 0) { s_base_status = *U.status; *U.status = TAMCMC_OK; *U.reject = 0; }
        __syncthreads();
    }
    const bool with_prior = a.desc.prior_class != 0;
    const int ip = e > 0 ? a.idx[e - 1] : 0;
    const double x0 = e > 0 ? a.params[(size_t)c * Np + ip] : 0.0, hh = e > 0 ? a.h[e - 1] : 0.0;
    if (e > 0) {
        if (tid == 0) s_params[ip] = x0 + hh;
        __syncthreads();
    }
    double lp = 0.0;
    if (with_prior) lp = wg_log_prior(a.desc, s_params, U, true);
    else {
        if (tid == FB - 1) mt::shared_scalars_base(a.desc.model_id, s_params, a.desc.plength, *U.S);
        __syncthreads();
    }
    // the likelihood is evaluated whatever the prior says: the caller combines the parts
    if (tid == 0) *U.status = TAMCMC_OK;
    __syncthreads();
    wg_unpack(a.desc, s_params, U, slot, a.T, true);
    if (tid == 0) {
        a.logPr_plus[slot] = lp;
        a.status[slot] = *U.status;
    }
    // the log-prior at theta - h e_k is only ever looked at when the forward point lies outside a prior's support (one-sided fallback of
    // the gradient's prior share): evaluated in that case alone (workgroup-uniform: every lane holds the same lp)
    double lp_minus = 0.0;
    if (e > 0 && with_prior && !isfinite(lp)) {
        __syncthreads();
        if (tid == 0) { s_params[ip] = x0 - hh; *U.reject = 0; }
        __syncthreads();
        lp_minus = wg_log_prior(a.desc, s_params, U, false);
    }
    if (tid == 0) a.logPr_minus[slot] = lp_minus;
    if (!a.windowed) return;
    __syncthreads();
    const int stride = a.desc.stride;
    if (e == 0) {
        // base evaluation of chain c: the base launch (B = C, model rows written) reads the SAME table rows
        if (tid == 0) {
            a.Bs.pairs[2 * c] = a.T.pairs[2 * slot];
            a.Bs.pairs[2 * c + 1] = a.T.pairs[2 * slot + 1];
            a.Bs.nh[c] = a.T.nh[slot];
            a.Bs.nn[c] = a.T.nn[slot];
            // no delta work for this slot
            a.D.pairs[2 * slot] = 0; a.D.pairs[2 * slot + 1] = 0; a.D.nh[slot] = 0; a.D.nn[slot] = 1;
            a.d_range[2 * slot] = 0; a.d_range[2 * slot + 1] = 0; a.d_flags[slot] = 0; a.d_row[slot] = c;
        }
        for (int i = tid; i < stride; i += FB) a.Bs.noise[(size_t)c * stride + i] = a.T.noise[(size_t)slot * stride + i];
        return;
    }
    // ---- delta table: the multiplets whose row differs from the base row, +new / -old, and the affected bin range ----
    __shared__ int s_namp, s_nshape;
    if (tid == 0) { s_lo = a.desc.Nx; s_hi = 0; s_nchg = 0; s_noise_chg = 0; s_namp = 0; s_nshape = 0; }
    __syncthreads();
    const bool ok = (a.status[slot] == TAMCMC_OK) && (s_base_status == TAMCMC_OK);
    tamcmc_multiplet *drows = a.D.mults + (size_t)slot * 2 * per;
    for (int i = tid; i < stride; i += FB) {
        const double vn = a.T.noise[(size_t)slot * stride + i], vo = a.T.noise[(size_t)(B + slot) * stride + i];
        a.D.noise[(size_t)slot * stride + i] = vn;
        a.d_noise_old[(size_t)slot * stride + i] = vo;
        if (ok && i < a.T.nn[slot] && vn != vo) s_noise_chg = 1;
    }
    // how row jdx changed: 0 not at all, 1 heights only (a parameter that only rescales heights -- inclination, visibilities, heights --
    // leaves frequencies, width, asymmetry and window untouched: +new and -old are then ONE row with the height differences), 2 otherwise
    auto change_of = [&](int jdx) -> int {
        if (!ok || jdx >= per) return 0;
        const unsigned long long *pn = (const unsigned long long *)&a.T.mults[(size_t)slot * per + jdx];
        const unsigned long long *po = (const unsigned long long *)&a.T.mults[(size_t)(B + slot) * per + jdx];
        bool chg = false;
        for (int w = 0; w < (int)(sizeof(tamcmc_multiplet) / 8); w++) chg = chg || (pn[w] != po[w]);
        if (!chg) return 0;
        const tamcmc_multiplet &rn = a.T.mults[(size_t)slot * per + jdx], &ro = a.T.mults[(size_t)(B + slot) * per + jdx];
        bool amp_only = (rn.l == ro.l) && (rn.i0 == ro.i0) && (rn.i1 == ro.i1) && (rn.fc == ro.fc) && (rn.gamma == ro.gamma) && (rn.asym == ro.asym);
        for (int m = 0; m < 7; m++) amp_only = amp_only && (rn.nu[m] == ro.nu[m]);
        return amp_only ? 1 : 2;
    };
    for (int j0 = 0; j0 < per; j0 += FB) {
        const int k = change_of(j0 + tid);
        if (k == 1) atomicAdd(&s_namp, 1);
        else if (k == 2) atomicAdd(&s_nshape, 1);
    }
    __syncthreads();
    // "full table": the pairs would be longer than the perturbed point's whole table (a parameter that moves most multiplets: a splitting
    // coefficient, the asymmetry) -- the delta launch then evaluates that table and subtracts the base model row (loglike_tile.h)
    const bool full = ok && !s_noise_chg && a.full_tables && (s_namp + 2 * s_nshape > per);
    if (full) {
        for (int jdx = tid; jdx < per; jdx += FB) drows[jdx] = a.T.mults[(size_t)slot * per + jdx];
    } else
        for (int j0 = 0; j0 < per; j0 += FB) {
            const int jdx = j0 + tid, k = change_of(jdx);
            if (k) {
                const tamcmc_multiplet &rn = a.T.mults[(size_t)slot * per + jdx];
                tamcmc_multiplet ro = a.T.mults[(size_t)(B + slot) * per + jdx];
                const int pos = atomicAdd(&s_nchg, k);   // order of the changed rows is irrelevant (a sum)
                if (k == 1) {
                    for (int m = 0; m < 7; m++) ro.hv[m] = rn.hv[m] - ro.hv[m];
                    drows[pos] = ro;
                } else {
                    for (int m = 0; m < 7; m++) ro.hv[m] = -ro.hv[m];
                    drows[pos] = rn;
                    drows[pos + 1] = ro;
                }
                atomicMin(&s_lo, min(rn.i0, ro.i0));
                atomicMax(&s_hi, max(rn.i1, ro.i1));
            }
        }
    __syncthreads();
    if (tid == 0) {
        const int n = full ? per : s_nchg;
        const bool all_bins = s_noise_chg || full;
        a.D.pairs[2 * slot] = slot * 2 * per;
        a.D.pairs[2 * slot + 1] = slot * 2 * per + n;  // n = rows written (one or two per changed multiplet, or the whole table)
        a.D.nh[slot] = a.T.nh[slot];
        a.D.nn[slot] = a.T.nn[slot];
        a.d_flags[slot] = s_noise_chg | (full ? 2 : 0);
        a.d_row[slot] = c;
        a.d_range[2 * slot] = all_bins ? 0 : (n ? s_lo : 0);
        a.d_range[2 * slot + 1] = all_bins ? a.desc.Nx : (n ? s_hi : 0);
    }
}

}  // namespace
}  // namespace tamcmc

using namespace tamcmc;

// ---------------------------------------------------------------------------------------------------------------
// One finite-difference batch = C chains x (Nvars + 1) evaluations.  fd_layout() places its constants, tables and results in ONE
// device block; fd_enqueue() launches the batch on the context's stream from parameter vectors that are ALREADY on the device and
// leaves the results there (the device-resident Langevin step, dev_mala.hip, consumes them in its next kernel); fd_run() is the host
// entry: upload, enqueue, download, gradient assembly.
namespace tamcmc {

static size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }

int FdBatch::layout(tamcmc_hip_ctx *c, int model_id_, int prior_class_, int C_, int64_t Nparams, const int32_t *plength, int Nvars_) {
    model_id = model_id_; prior_class = prior_class_; C = C_; Np = Nparams; Nvars = Nvars_;
    per = mt::count_multiplets(model_id, plength);
    if (per < 0) return TAMCMC_ERR_BAD_MODEL;
    stride = plength[8] > 0 ? plength[8] : 1;
    if ((stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    E = Nvars + 1; B = C * E;
    const size_t Nv = (size_t)Nvars;
    size_t o = 0;
    o_params = o; o = al16(o + (size_t)C * Np * 8);
    o_h = o; o = al16(o + Nv * 8);
    o_pr = o; o = al16(o + 4 * (size_t)Np * 8);
    o_ex = o; o = al16(o + 10 * 8);
    o_pl = o; o = al16(o + 11 * 4);
    o_idx = o; o = al16(o + Nv * 4);
    o_sw = o; o = al16(o + (size_t)Np * 4);
    in_bytes = o;
    o_lpp = o; o = al16(o + (size_t)B * 8);
    o_lpm = o; o = al16(o + (size_t)B * 8);
    o_st = o; o = al16(o + (size_t)B * 4);
    out_bytes = o - in_bytes;
    // windowed finite differences (FAST modes): only the multiplets a perturbation changes are re-evaluated, on their
    // windows, against the stored base model row ("the main algorithmic lever")
    windowed = c->fd_windowed && c->precision != TAMCMC_PRECISION_STRICT && delta_geometry(c->wgs, c->K) && Nvars > 0;
    const int nslots = windowed ? 2 * B : B;                   // windowed: slots [B, 2B) = per-block copies of the base table
    const StageLayout L(nslots, stride, (size_t)nslots * per);
    o_tab = o; o = al16(o + L.bytes);
    o_dtab = o_btab = o_drange = o_dflags = o_drow = o_dnold = 0;
    if (windowed) {
        const StageLayout LD(B, stride, (size_t)B * 2 * per);      // delta launch input block
        const StageLayout LB(C, stride, 0);                        // base launch: ranges / counts / noise rows by chain
        o_dtab = o; o = al16(o + LD.bytes);
        o_btab = o; o = al16(o + LB.bytes);
        o_drange = o; o = al16(o + (size_t)2 * B * 4);
        o_dflags = o; o = al16(o + (size_t)B * 4);
        o_drow = o; o = al16(o + (size_t)B * 4);
        o_dnold = o; o = al16(o + (size_t)B * stride * 8);
    }
    total_bytes = o;
    const int tbins = tile_bins(c->wgs, c->K);
    ntiles = (int)((c->Nx + tbins - 1) / tbins);
    nS = windowed ? (size_t)C + B : (size_t)B;
    return TAMCMC_OK;
}

// db = the batch's device block (total_bytes), constants in place; d_params: C x Np parameter vectors on the device (nullptr: the
// block's own params area); part / S / model: scratch sized nS*ntiles*2, nS, 3*C*Nx + 2*C*ntiles*FD_MOM + ceil(B*ntiles/8) (model only when windowed); bgbuf: C or B x ntiles x 8.
int FdBatch::enqueue(tamcmc_hip_ctx *c, unsigned char *db, const double *d_params, double *part, double *S, double *model, double *bgbuf,
                     hipEvent_t ev0, hipEvent_t ev1) {
    hipStream_t st = c->stream;
    auto table_ptrs = [](unsigned char *base, const StageLayout &Lx) {
        TablePtrs T;
        T.mults = (tamcmc_multiplet *)(base + Lx.off_mults); T.pairs = (int *)(base + Lx.off_pairs);
        T.nh = (int *)(base + Lx.off_nh); T.nn = (int *)(base + Lx.off_nn); T.noise = (double *)(base + Lx.off_noise);
        return T;
    };
    const int nslots = windowed ? 2 * B : B;
    const StageLayout L(nslots, stride, (size_t)nslots * per), LD(B, stride, (size_t)B * 2 * per), LB(C, stride, 0);
    FdArgs fa;
    fa.desc.model_id = model_id; fa.desc.prior_class = prior_class; fa.desc.Np = (int)Np; fa.desc.per = per;
    fa.desc.stride = stride; fa.desc.Nx = (int)c->Nx;
    fa.desc.x_first = c->hx[0]; fa.desc.x_last = c->hx[(size_t)c->Nx - 1]; fa.desc.step = c->hx[1] - c->hx[0];
    fa.desc.plength = (const int *)(db + o_pl); fa.desc.priors_switch = (const int *)(db + o_sw);
    fa.desc.priors = (const double *)(db + o_pr); fa.desc.extra = (const double *)(db + o_ex); fa.desc.poly = c->d_poly.p;
    fa.T = table_ptrs(db + o_tab, L);
    fa.C = C; fa.E = E; fa.Nv = Nvars;
    fa.params = d_params ? d_params : (const double *)(db + o_params);
    fa.idx = (const int *)(db + o_idx); fa.h = (const double *)(db + o_h);
    fa.logPr_plus = (double *)(db + o_lpp); fa.logPr_minus = (double *)(db + o_lpm); fa.status = (int *)(db + o_st);
    fa.windowed = windowed ? 1 : 0;
    fa.full_tables = (windowed && c->precision == TAMCMC_PRECISION_FAST && bgbuf) ? 1 : 0;
    fa.D = fa.T; fa.Bs = fa.T;
    fa.d_range = nullptr; fa.d_flags = nullptr; fa.d_row = nullptr; fa.d_noise_old = nullptr;
    if (windowed) {
        fa.D = table_ptrs(db + o_dtab, LD);
        fa.Bs = table_ptrs(db + o_btab, LB);
        fa.d_range = (int *)(db + o_drange); fa.d_flags = (int *)(db + o_dflags); fa.d_row = (int *)(db + o_drow);
        fa.d_noise_old = (double *)(db + o_dnold);
    }
    const size_t lds = (size_t)Np * 8 + unpack_lds_bytes() + 32;
    hipLaunchKernelGGL(k_fd_unpack, dim3(B), dim3(FB), lds, st, fa);
    HIPCHK(c, hipGetLastError());

    const int Nx = (int)c->Nx;
    LoglikeArgs a;
    a.x = c->dx.p; a.y = c->dy.p; a.logx = c->dlogx.p; a.Nx = Nx; a.ntiles = ntiles;
    a.x0 = c->hx[0]; a.step = c->hx[1] - c->hx[0];
    a.noise_stride = stride; a.model = nullptr;
    if (ev0) HIPCHK(c, hipEventRecord(ev0, st));
    if (!windowed) {
        a.B = B;
        a.mults = fa.T.mults; a.offsets = fa.T.pairs; a.noise = fa.T.noise; a.nharvey = fa.T.nh; a.nnoise = fa.T.nn;
        a.partials = part;
        if (c->precision == TAMCMC_PRECISION_FAST) {
            HIPCHK(c, launch_bg_poly(a, c->wgs, c->K, bgbuf, st));
            a.bg_poly = bgbuf;
        }
        HIPCHK(c, launch_loglike(a, c->precision, c->wgs, c->K, false, st));
        HIPCHK(c, launch_finalize(part, B, ntiles, S, st));
    } else {
        // (1) the C base points: full evaluation, model rows kept
        a.B = C;
        a.mults = fa.T.mults; a.offsets = fa.Bs.pairs; a.noise = fa.Bs.noise; a.nharvey = fa.Bs.nh; a.nnoise = fa.Bs.nn;
        a.partials = part; a.model = model; a.fd_rows = model; a.fd_plane = (size_t)C * Nx;  // (1/M0, y/M0 and M0 planes instead of the rows)
        if (c->precision == TAMCMC_PRECISION_FAST) {
            HIPCHK(c, launch_bg_poly(a, c->wgs, c->K, bgbuf, st));
            a.bg_poly = bgbuf;
        }
        HIPCHK(c, launch_loglike(a, c->precision, c->wgs, c->K, true, st));
        HIPCHK(c, launch_finalize(part, C, ntiles, S, st));
        // (1b) moments of the base points per tile, behind the three planes: the delta launch's far-only tiles take their sums from them
        double *mom = (c->precision == TAMCMC_PRECISION_FAST && c->wgs == 64) ? model + 3 * (size_t)C * Nx : nullptr;
        double *momT = mom ? mom + (size_t)C * ntiles * FD_MOM : nullptr;
        if (mom) HIPCHK(c, launch_fd_moments(a, c->wgs, c->K, mom, momT, st));
        // (2) the C*Nvars perturbed points: log-likelihood DIFFERENCES from the delta tables
        LoglikeArgs d = a;
        d.fd_mom = mom;
        d.fd_momT = momT;
        d.B = B; d.model = nullptr; d.fd_rows = nullptr;
        d.bg_poly = fa.full_tables ? bgbuf : nullptr;  // (rows by base point: read by the "full table" evaluations only)
        d.mults = fa.D.mults; d.offsets = fa.D.pairs; d.noise = fa.D.noise; d.nharvey = fa.D.nh; d.nnoise = fa.D.nn;
        d.partials = part + (size_t)C * ntiles * 2;
        d.d_range = fa.d_range; d.d_flags = fa.d_flags; d.d_row = fa.d_row; d.d_noise_old = fa.d_noise_old; d.model0 = model;
        if (mom) {  // (2a) the far-only tiles of the light evaluations, one lane per tile; the delta launch skips what this marks done
            unsigned char *done = (unsigned char *)(momT + (size_t)C * ntiles * FD_MOM);
            HIPCHK(c, launch_fd_far(d, c->wgs, c->K, done, st));
            d.d_done = done;
        }
        d_done = d.d_done;
        tile_bins_ = tile_bins(c->wgs, c->K);
        HIPCHK(c, launch_loglike_delta(d, c->precision, c->wgs, c->K, st));
        HIPCHK(c, launch_finalize(d.partials, B, ntiles, S + C, st));
    }
    if (ev1) HIPCHK(c, hipEventRecord(ev1, st));
    return TAMCMC_OK;
}

long FdBatch::bins_not_walked() const {
    if (!d_done) return 0;
    std::vector<unsigned char> f((size_t)B * ntiles);
    if (hipMemcpy(f.data(), d_done, f.size(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    long n = 0;
    for (unsigned char v : f) n += v ? 1 : 0;
    return n * (long)tile_bins_;
}

int fd_ensure_poly(tamcmc_hip_ctx *c) {
    if (!c->poly_ready) {
        HIPCHK(c, c->d_poly.reserve(sizeof(mt::PolyTab)));
        hipLaunchKernelGGL(k_fill_poly_fd, dim3(1), dim3(64), 0, c->stream, (mt::PolyTab *)c->d_poly.p);
        HIPCHK(c, hipGetLastError());
        c->poly_ready = true;
    }
    return TAMCMC_OK;
}

}  // namespace tamcmc

static int fd_run(tamcmc_hip_ctx *c, int model_id, int prior_class, int C, const double *params, int64_t Nparams,
                  const int32_t *plength, const int32_t *index_to_relax, int Nvars, const double *hstep, const double *Tcoefs,
                  double p, const double *priors, const int32_t *priors_switch, const double *extra_priors, double *logL0,
                  double *logPr0, double *grad, double *grad_prior) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (c->Nx <= 0) return TAMCMC_ERR_NO_SPECTRUM;
    if (C < 0 || Nvars < 0 || !params || !plength || !index_to_relax || !hstep || !logL0 || !grad || Nparams < 1) return TAMCMC_ERR_BAD_ARG;
    if (prior_class != 0 && (!priors || !priors_switch || !extra_priors)) return TAMCMC_ERR_BAD_ARG;
    if (C == 0) return TAMCMC_OK;
    long psum = 0;
    for (int i = 0; i < 11; i++) psum += plength[i];
    if (psum != Nparams) return TAMCMC_ERR_BAD_ARG;
    for (int k = 0; k < Nvars; k++)
        if (index_to_relax[k] < 0 || index_to_relax[k] >= Nparams) return TAMCMC_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    FdBatch fb;
    int rc = fb.layout(c, model_id, prior_class, C, Nparams, plength, Nvars);
    if (rc) return rc;
    rc = fd_ensure_poly(c);
    if (rc) return rc;
    const int E = fb.E, B = fb.B;
    const size_t Np = (size_t)Nparams, Nv = (size_t)Nvars;
    const bool windowed = fb.windowed;
    const size_t in_bytes = fb.in_bytes, out_bytes = fb.out_bytes, o_lpp = fb.o_lpp, o_lpm = fb.o_lpm, o_st = fb.o_st, o_drange = fb.o_drange;
    HIPCHK(c, c->h_fd.reserve(in_bytes + out_bytes));
    HIPCHK(c, c->d_fd.reserve(fb.total_bytes));
    unsigned char *hb = c->h_fd.p, *db = c->d_fd.p;
    std::memcpy(hb + fb.o_params, params, (size_t)C * Np * 8);
    std::memcpy(hb + fb.o_h, hstep, Nv * 8);
    if (prior_class != 0) {
        std::memcpy(hb + fb.o_pr, priors, 4 * Np * 8);
        std::memcpy(hb + fb.o_ex, extra_priors, 10 * 8);
        std::memcpy(hb + fb.o_sw, priors_switch, Np * 4);
    }
    std::memcpy(hb + fb.o_pl, plength, 11 * 4);
    std::memcpy(hb + fb.o_idx, index_to_relax, Nv * 4);
    HIPCHK(c, hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, st));
    const size_t nS = fb.nS;
    HIPCHK(c, c->d_part.reserve(nS * fb.ntiles * 2));
    HIPCHK(c, c->d_S.reserve(nS));
    HIPCHK(c, c->h_S.reserve(nS));
    if (windowed) HIPCHK(c, c->d_model.reserve(3 * (size_t)C * c->Nx + 2 * (size_t)C * fb.ntiles * FD_MOM + ((size_t)fb.B * fb.ntiles + 7) / 8));  // three planes: 1/M0, y/M0, M0; tile moments (two layouts); done flags
    if (c->precision == TAMCMC_PRECISION_FAST) HIPCHK(c, c->d_bg.reserve((size_t)(windowed ? C : B) * fb.ntiles * 8));
    rc = fb.enqueue(c, db, nullptr, c->d_part.p, c->d_S.p, c->d_model.p, c->d_bg.p, c->timing ? c->ev0 : nullptr, c->timing ? c->ev1 : nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_S.p, c->d_S.p, nS * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hb + in_bytes, db + in_bytes, out_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->timing) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->kernel_ms += ms;
        c->launches += 1;
        c->evals += B;
        if (windowed) {  // what the delta launch really touched (roofline bookkeeping of bench.py)
            std::vector<int> rg((size_t)2 * B);
            HIPCHK(c, hipMemcpy(rg.data(), db + o_drange, rg.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int s = 0; s < B; s++) c->fd_bins += rg[2 * (size_t)s + 1] - rg[2 * (size_t)s];
            c->fd_bins -= fb.bins_not_walked();  // (far-only tiles taken from the base point's moments)
            c->fd_delta_evals += B;
            std::vector<int> fl((size_t)B);
            HIPCHK(c, hipMemcpy(fl.data(), db + fb.o_dflags, fl.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int s = 0; s < B; s++) c->fd_full_evals += (fl[(size_t)s] & 2) ? 1 : 0;
        }
    }
    const double *lpp = (const double *)(hb + o_lpp), *lpm = (const double *)(hb + o_lpm);
    const int *stt = (const int *)(hb + o_st);
    const long pl = (long)p;
    int first_err = TAMCMC_OK;
    for (int ch = 0; ch < C; ch++) {
        const double T = Tcoefs ? Tcoefs[ch] : 1.0;
        // S of evaluation e: full sums (non-windowed) or base sum + difference (windowed)
        auto scaled = [&](double S) {
            long double f = S;
            f = -pl * f;
            return (double)(f / T);
        };
        auto failed = [&](int e) {
            const size_t s = (size_t)ch * E + e;
            if (stt[s] != TAMCMC_OK) { if (first_err == TAMCMC_OK) first_err = stt[s]; return true; }
            return false;
        };
        const double L0 = failed(0) ? (double)NAN : scaled(windowed ? c->h_S.p[ch] : c->h_S.p[(size_t)ch * E]);
        auto dlogL_of = [&](int e) {  // logL(theta + h e_k) - logL(theta)
            if (failed(e)) return (double)NAN;
            if (windowed) return scaled(c->h_S.p[(size_t)C + (size_t)ch * E + e]);
            return scaled(c->h_S.p[(size_t)ch * E + e]) - L0;
        };
        logL0[ch] = L0;
        if (logPr0) logPr0[ch] = lpp[(size_t)ch * E];
        const double pr0 = lpp[(size_t)ch * E];
        for (int k = 0; k < Nvars; k++) {
            const double x0 = params[(size_t)ch * Np + index_to_relax[k]];
            volatile double xp = x0 + hstep[k];
            const double happ = xp - x0;  // the step actually applied (the device adds the same two doubles)
            double g = dlogL_of(k + 1) / happ;
            if (prior_class != 0) {
                if (!std::isfinite(g)) g = 0.0;
                const double prp = lpp[(size_t)ch * E + k + 1], prm = lpm[(size_t)ch * E + k + 1];
                double gp;
                if (std::isfinite(prp)) gp = (prp - pr0) / happ;
                else gp = std::isfinite(prm) ? (pr0 - prm) / happ : 0.0;  // forward point outside the support: backward, else flat
                g += gp;
                if (grad_prior) grad_prior[(size_t)ch * Nv + k] = gp;
            }
            grad[(size_t)ch * Nv + k] = g;
        }
    }
    return first_err;
}

extern "C" {

int tamcmc_hip_fd_gradient(tamcmc_hip_ctx *c, int model_id, int C, const double *params, int64_t Nparams,
                           const int32_t *plength, const int32_t *index_to_relax, int Nvars, const double *hstep,
                           const double *Tcoefs, double p, double *logL0, double *grad) {
    return fd_run(c, model_id, 0, C, params, Nparams, plength, index_to_relax, Nvars, hstep, Tcoefs, p, nullptr, nullptr, nullptr,
                  logL0, nullptr, grad, nullptr);
}

int tamcmc_hip_fd_gradient_posterior(tamcmc_hip_ctx *c, int model_id, int prior_class, int C, const double *params,
                                     int64_t Nparams, const int32_t *plength, const int32_t *index_to_relax, int Nvars,
                                     const double *hstep, const double *Tcoefs, double p, const double *priors,
                                     const int32_t *priors_switch, const double *extra_priors, double *logL0, double *logPr0,
                                     double *grad, double *grad_prior) {
    if (prior_class != 2 && prior_class != 3) return TAMCMC_ERR_BAD_MODEL;
    return fd_run(c, model_id, prior_class, C, params, Nparams, plength, index_to_relax, Nvars, hstep, Tcoefs, p, priors,
                  priors_switch, extra_priors, logL0, logPr0, grad, grad_prior);
}

}  // extern "C"
