// capi.hip -- the C-ABI layer of include/tamcmc_hip.h: context, resident spectrum, staging buffers,
// batched likelihood launches.  One context owns one HIP stream on one device; host<->device staging goes
// through pinned buffers so that the small per-call tables travel with hipMemcpyAsync on that stream.
// There is NO CPU fallback: without a HIP device every entry point fails with TAMCMC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tamcmc_hip.h"
#include "kernels.h"
#include "rgb_prestep.h"
#include "mode_tables.h"

#include "ctx.h"

using tamcmc::StageLayout;

// Measured best tile geometry per arithmetic mode, unless the caller chose one (tools/gpu_probe.py, tools/groups_probe.py).
// FAST (far field): one wave per tile of 512 bins; spectra below 32768 bins take 256-bin tiles -- on a coarse grid a tile is wide in
// units of the mode spacing, most multiplets fall into its near field and the tile's own chain of work sets the launch time (1e4 bins,
// 10-20 chains: 32.6 -> 21.8 us per iteration).  Per-bin loops (STRICT, FAST_DIRECT): four waves per 1024-bin tile.
static void default_geometry(tamcmc_hip_ctx *c) {
    if (c->geom_user_set) return;
    if (c->precision == TAMCMC_PRECISION_FAST) { c->wgs = 64; c->K = (c->Nx > 0 && c->Nx < 32768) ? 4 : 8; }
    else { c->wgs = 256; c->K = 4; }
}

extern "C" {

const char *tamcmc_hip_version(void) { return "tamcmc-c_amd 0.1 (gfx950)"; }

int tamcmc_hip_create(tamcmc_hip_ctx **out, int device) {
    if (!out) return TAMCMC_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TAMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return TAMCMC_ERR_BAD_ARG;
    tamcmc_hip_ctx *c = new tamcmc_hip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return TAMCMC_ERR_HIP;
    }
    *out = c;
    return TAMCMC_OK;
}

// (private, host_capi.cpp) a sampler borrows the context for its whole life
void tamcmc_hip_ctx_attach(tamcmc_hip_ctx *c) { if (c) c->attached++; }
void tamcmc_hip_ctx_detach(tamcmc_hip_ctx *c) {
    if (!c) return;
    c->attached--;
    if (c->attached <= 0 && c->zombie) { c->zombie = false; tamcmc_hip_destroy(c); }
}

void tamcmc_hip_destroy(tamcmc_hip_ctx *c) {
    if (!c) return;
    if (c->attached > 0) { c->zombie = true; return; }  // freed by the last sampler's destruction
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->dx.release(); c->dy.release(); c->dlogx.release();
    c->h_stage.release(); c->d_stage.release();
    c->d_part.release(); c->d_S.release(); c->d_model.release(); c->h_S.release();
    c->d_fd.release(); c->d_poly.release(); c->h_fd.release();
    c->d_rgb.release(); c->h_rgb.release(); c->d_bg.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *tamcmc_hip_last_error(const tamcmc_hip_ctx *c) { return c ? c->err.c_str() : "null context"; }

int tamcmc_hip_set_option(tamcmc_hip_ctx *c, int option, int64_t value) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    switch (option) {
    case TAMCMC_OPT_PRECISION:
        if (value != TAMCMC_PRECISION_STRICT && value != TAMCMC_PRECISION_FAST && value != TAMCMC_PRECISION_FAST_DIRECT)
            return TAMCMC_ERR_BAD_ARG;
        c->precision = (int)value;
        default_geometry(c);
        return TAMCMC_OK;
    case TAMCMC_OPT_TIMING: c->timing = value ? 1 : 0; return TAMCMC_OK;
    case TAMCMC_OPT_BINS_PER_THREAD:
        if (!tamcmc::valid_geometry(c->wgs, (int)value)) return TAMCMC_ERR_BAD_ARG;
        c->K = (int)value;
        c->geom_user_set = true;
        return TAMCMC_OK;
    case TAMCMC_OPT_FD_WINDOWED: c->fd_windowed = value ? 1 : 0; return TAMCMC_OK;
    case TAMCMC_OPT_STEP_SCHEME:
        if (value < 0 || value > 3) return TAMCMC_ERR_BAD_ARG;
        c->step_scheme = (int)value;
        return TAMCMC_OK;
    case TAMCMC_OPT_ARMM_DENSE_SCAN: c->armm_dense = value ? 1 : 0; return TAMCMC_OK;
    case TAMCMC_OPT_WORKGROUP:  // sets the workgroup size AND its default bins per thread
        if (value != 64 && value != 256) return TAMCMC_ERR_BAD_ARG;
        c->wgs = (int)value;
        c->K = (value == 64) ? 8 : 4;
        c->geom_user_set = true;
        return TAMCMC_OK;
    default: return TAMCMC_ERR_BAD_ARG;
    }
}

void *tamcmc_hip_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void tamcmc_hip_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int tamcmc_hip_set_spectrum(tamcmc_hip_ctx *c, const double *x, const double *y, int64_t Nx) {
    if (!c || !x || !y || Nx < 2 || Nx > 0x7fffffff) return TAMCMC_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->hx.assign(x, x + Nx);
    std::vector<double> lx((size_t)Nx);
    for (int64_t i = 0; i < Nx; i++) lx[(size_t)i] = std::log(x[i]);  // FAST Harvey terms: (a x)^p = exp(p (ln a + ln x))
    HIPCHK(c, c->dx.reserve((size_t)Nx));
    HIPCHK(c, c->dy.reserve((size_t)Nx));
    HIPCHK(c, c->dlogx.reserve((size_t)Nx));
    HIPCHK(c, hipMemcpyAsync(c->dx.p, x, (size_t)Nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dy.p, y, (size_t)Nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dlogx.p, lx.data(), (size_t)Nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->Nx = Nx;
    default_geometry(c);
    return TAMCMC_OK;
}

// Launch on the tables staged in c->h_stage (layout L).
// device_tables: the block is already in c->d_stage (built there by device kernels); tile_rot then comes from the caller.
static int run_staged(tamcmc_hip_ctx *c, int B, const StageLayout &L, int noise_stride, const double *Tcoefs, double p,
                      double *logL, double *model, bool device_tables = false, int tile_rot = 0) {
    const int Nx = (int)c->Nx;
    const int tb = tamcmc::tile_bins(c->wgs, c->K);
    const int ntiles = (Nx + tb - 1) / tb;
    HIPCHK(c, c->d_stage.reserve(L.bytes));
    HIPCHK(c, c->d_part.reserve((size_t)B * ntiles * 2));
    HIPCHK(c, c->d_S.reserve((size_t)B));
    HIPCHK(c, c->h_S.reserve((size_t)B));
    if (model) HIPCHK(c, c->d_model.reserve((size_t)B * Nx));
    hipStream_t st = c->stream;
    if (!device_tables) HIPCHK(c, hipMemcpyAsync(c->d_stage.p, c->h_stage.p, L.bytes, hipMemcpyHostToDevice, st));

    tamcmc::LoglikeArgs a;
    a.x = c->dx.p; a.y = c->dy.p; a.logx = c->dlogx.p; a.Nx = Nx; a.B = B; a.ntiles = ntiles;
    a.x0 = c->hx[0]; a.step = c->hx[1] - c->hx[0];
    a.mults = (const tamcmc_multiplet *)(c->d_stage.p + L.off_mults);
    a.offsets = (const int32_t *)(c->d_stage.p + L.off_pairs);
    a.noise = (const double *)(c->d_stage.p + L.off_noise);
    a.noise_stride = noise_stride;
    a.nharvey = (const int32_t *)(c->d_stage.p + L.off_nh);
    a.nnoise = (const int32_t *)(c->d_stage.p + L.off_nn);
    a.partials = c->d_part.p;
    a.model = model ? c->d_model.p : nullptr;
#ifdef TAMCMC_PROBE  // probe build only (tools/phase_probe.py)
    if (const char *ep = getenv("TAMCMC_PROBE_SKIP")) {
        a.probe = atoi(ep);
        static bool warned = false;
        if (a.probe && !warned) {
            warned = true;
            fprintf(stderr, "tamcmc_hip (PROBE BUILD): TAMCMC_PROBE_SKIP=%d skips kernel phases -- the log-likelihoods of this process are WRONG\n", a.probe);
        }
    }
#endif
#ifdef TAMCMC_PROBE  // phase stamps of the middle tile of evaluation 0 (wall_clock64: 100 MHz), printed after the launch
    static long *probe_dbg = nullptr;
    if (getenv("TAMCMC_PROBE_STAMPS")) {
        if (!probe_dbg) (void)hipMalloc((void **)&probe_dbg, 16 * sizeof(long));
        (void)hipMemsetAsync(probe_dbg, 0, 16 * sizeof(long), st);
        a.dbg = probe_dbg;
    }
#endif
    if (device_tables) a.tile_rot = tile_rot;
    else {
        const int32_t *pairs = (const int32_t *)(c->h_stage.p + L.off_pairs);
        const tamcmc_multiplet *hm = (const tamcmc_multiplet *)(c->h_stage.p + L.off_mults);
        a.tile_rot = tamcmc::pick_tile_rot(hm + pairs[0], pairs[1] - pairs[0], a.x0, a.step, tb, ntiles);
    }
    if (c->precision == TAMCMC_PRECISION_FAST) {  // background series per (evaluation, tile), built once instead of per workgroup
        HIPCHK(c, c->d_bg.reserve((size_t)B * ntiles * 8));
        HIPCHK(c, tamcmc::launch_bg_poly(a, c->wgs, c->K, c->d_bg.p, st));
        a.bg_poly = c->d_bg.p;
    }
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, st));
    HIPCHK(c, tamcmc::launch_loglike(a, c->precision, c->wgs, c->K, model != nullptr, st));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, tamcmc::launch_finalize(c->d_part.p, B, ntiles, c->d_S.p, st));
    HIPCHK(c, hipMemcpyAsync(c->h_S.p, c->d_S.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    if (model)
        HIPCHK(c, hipMemcpyAsync(model, c->d_model.p, (size_t)B * Nx * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
#ifdef TAMCMC_PROBE
    if (a.dbg) {
        long h[16];
        (void)hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "tile stamps (us since entry): setup %.2f | staged %.2f | near %.2f | far coef %.2f | poly %.2f | reduced+stored %.2f\n", (h[1] - h[0]) * 0.01,
                (h[2] - h[0]) * 0.01, (h[3] - h[0]) * 0.01, (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01, (h[6] - h[0]) * 0.01);
    }
#endif
    if (c->timing) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->kernel_ms += ms;
        c->launches += 1;
        c->evals += B;
    }
    // call_likelihood (model_def.cpp:399-401): f = -p*(sum1+sum2) in long double, then / Tcoefs[m]
    const long pl = (long)p;
    for (int b = 0; b < B; b++) {
        long double f = c->h_S.p[b];
        f = -pl * f;
        logL[b] = (double)(f / (Tcoefs ? Tcoefs[b] : 1.0));
    }
    return TAMCMC_OK;
}

int tamcmc_hip_loglike_batch(tamcmc_hip_ctx *c, int B, const tamcmc_multiplet *mults, const int32_t *offsets,
                             const double *noise, int noise_stride, const int32_t *nharvey, const int32_t *nnoise,
                             const double *Tcoefs, double p, double *logL, double *model) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (c->Nx <= 0) return TAMCMC_ERR_NO_SPECTRUM;
    if (B < 0 || !offsets || !noise || !nharvey || !nnoise || !logL || noise_stride < 1) return TAMCMC_ERR_BAD_ARG;
    if (B == 0) return TAMCMC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // validate what the kernel's indexing assumes before anything reaches the GPU
    if (offsets[0] != 0) return TAMCMC_ERR_BAD_ARG;
    for (int b = 0; b < B; b++) {
        if (offsets[b + 1] < offsets[b]) return TAMCMC_ERR_BAD_ARG;
        if (nnoise[b] < 1 || nnoise[b] > noise_stride) return TAMCMC_ERR_BAD_ARG;
        if (nharvey[b] < 0 || nharvey[b] > TAMCMC_MAX_HARVEY || 3 * nharvey[b] + 1 > nnoise[b]) return TAMCMC_ERR_BAD_ARG;
    }
    const size_t total = (size_t)offsets[B];
    if (total && !mults) return TAMCMC_ERR_BAD_ARG;
    for (size_t i = 0; i < total; i++) {
        const tamcmc_multiplet &m = mults[i];
        if (m.l < 0 || m.l > 3 || m.i0 < 0 || m.i1 > c->Nx || m.i1 <= m.i0) return TAMCMC_ERR_BAD_ARG;
    }
    const StageLayout L(B, noise_stride, total);
    HIPCHK(c, c->h_stage.reserve(L.bytes));
    unsigned char *h = c->h_stage.p;
    int32_t *pairs = (int32_t *)(h + L.off_pairs);
    for (int b = 0; b < B; b++) { pairs[2 * b] = offsets[b]; pairs[2 * b + 1] = offsets[b + 1]; }
    std::memcpy(h + L.off_nh, nharvey, (size_t)B * sizeof(int32_t));
    std::memcpy(h + L.off_nn, nnoise, (size_t)B * sizeof(int32_t));
    std::memcpy(h + L.off_noise, noise, (size_t)B * noise_stride * sizeof(double));
    if (total) std::memcpy(h + L.off_mults, mults, total * sizeof(tamcmc_multiplet));
    return run_staged(c, B, L, noise_stride, Tcoefs, p, logL, model);
}

// Build the B tables straight into the pinned staging block.  Every vector owns the fixed slot
// [b*per, (b+1)*per) of the multiplet area, so the builders are independent: host threads over parameter vectors
// (the reference's OpenMP-over-chains axis, MALA.cpp:648, moved to the scalar unpack).  Vectors whose table fails
// keep an empty table; their status is reported and their logL is forced to NaN afterwards.
static int stage_params(tamcmc_hip_ctx *c, int model_id, int B, const double *params, int64_t Nparams,
                        const int32_t *plength, int32_t *status, int *per_out, int *stride_out, int *first_err) {
    const int per = tamcmc::count_multiplets(model_id, plength);
    if (per < 0) return TAMCMC_ERR_BAD_MODEL;
    const int stride = plength[8] > 0 ? plength[8] : 1;
    if ((stride - 1) / 3 > TAMCMC_MAX_HARVEY) return TAMCMC_ERR_BAD_ARG;
    const StageLayout L(B, stride, (size_t)B * per);
    HIPCHK(c, c->h_stage.reserve(L.bytes));
    unsigned char *h = c->h_stage.p;
    int32_t *pairs = (int32_t *)(h + L.off_pairs), *h_nh = (int32_t *)(h + L.off_nh), *h_nn = (int32_t *)(h + L.off_nn);
    double *h_noise = (double *)(h + L.off_noise);
    tamcmc_multiplet *h_mults = (tamcmc_multiplet *)(h + L.off_mults);
    const double *hx = c->hx.data();
    const int64_t Nx = c->Nx;
    const int nthreads = B >= 8 ? (B < 16 ? B : 16) : 1;
    (void)nthreads;  // (only the host pass sees the OpenMP pragma)
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
    for (int b = 0; b < B; b++) {
        int n = 0, nh = 0, nn = 0;
        int st = tamcmc::build_mode_table(model_id, params + (size_t)b * Nparams, plength, hx, Nx,
                                          h_mults + (size_t)b * per, per, &n, h_noise + (size_t)b * stride, &nh, &nn);
        if (st == TAMCMC_OK && n > per) st = TAMCMC_ERR_BAD_ARG;
        status[b] = st;
        if (st != TAMCMC_OK) {
            n = 0; nh = 0; nn = 1;
            h_noise[(size_t)b * stride] = 1.0;  // harmless placeholder row; logL[b] is overwritten with NaN
        }
        pairs[2 * b] = (int32_t)((size_t)b * per);
        pairs[2 * b + 1] = (int32_t)((size_t)b * per + n);
        h_nh[b] = nh;
        h_nn[b] = nn;
    }
    *first_err = TAMCMC_OK;
    for (int b = 0; b < B; b++)
        if (status[b] != TAMCMC_OK && *first_err == TAMCMC_OK) *first_err = status[b];
    *per_out = per;
    *stride_out = stride;
    return TAMCMC_OK;
}

int tamcmc_hip_loglike_params_batch(tamcmc_hip_ctx *c, int model_id, int B, const double *params, int64_t Nparams,
                                    const int32_t *plength, const double *Tcoefs, double p, double *logL,
                                    double *model, int32_t *status) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (c->Nx <= 0) return TAMCMC_ERR_NO_SPECTRUM;
    if (B < 0 || !params || !plength || !logL || Nparams < 1) return TAMCMC_ERR_BAD_ARG;
    if (B == 0) return TAMCMC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int per = 0, stride = 1, first_err = TAMCMC_OK;
    std::vector<int32_t> st_local;
    if (!status) { st_local.resize((size_t)B); status = st_local.data(); }
    // the red-giant model needs its device pre-step (mixed-mode solver, zeta) before the rows can be written
    const bool rgb = (model_id == TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4_ID || model_id == TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4_ID);
    int rot = 0;
    int rc = rgb ? tamcmc::rgb_stage_params(c, model_id, B, params, Nparams, plength, status, &per, &stride, &first_err, &rot)
                 : stage_params(c, model_id, B, params, Nparams, plength, status, &per, &stride, &first_err);
    if (rc) return rc;
    rc = run_staged(c, B, StageLayout(B, stride, (size_t)B * per), stride, Tcoefs, p, logL, model, rgb, rot);
    if (rc) return rc;
    if (rgb) tamcmc::rgb_collect_status(c, B, status, &first_err);
    for (int b = 0; b < B; b++)
        if (status[b] != TAMCMC_OK) logL[b] = NAN;
    return first_err;
}

// The l=1 mixed modes of one red-giant parameter vector, as the device pre-step computes them for the likelihood table.
int tamcmc_hip_rgb_mixed_modes(tamcmc_hip_ctx *c, int model_id, const double *params, int64_t Nparams, const int32_t *plength,
                               int max_modes, double *nu_m, double *zeta, double *h1_h0, int *n_modes) {
    if (!c || !params || !plength || !n_modes || max_modes < 0 || Nparams < 1) return TAMCMC_ERR_BAD_ARG;
    if (c->Nx <= 0) return TAMCMC_ERR_NO_SPECTRUM;
    if (model_id != TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4_ID && model_id != TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4_ID) return TAMCMC_ERR_BAD_MODEL;
    HIPCHK(c, hipSetDevice(c->device));
    int per = 0, stride = 1, first_err = TAMCMC_OK, rot = 0;
    int32_t status = TAMCMC_OK;
    int rc = tamcmc::rgb_stage_params(c, model_id, 1, params, Nparams, plength, &status, &per, &stride, &first_err, &rot);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tamcmc::rgb_collect_status(c, 1, &status, &first_err);
    if (status != TAMCMC_OK) return status;
    std::vector<double> z((size_t)(max_modes > 0 ? max_modes : 1));
    rc = tamcmc::rgb_fetch_modes(c, 1, 0, max_modes, nu_m, z.data(), n_modes);
    if (rc) return rc;
    const int n = *n_modes < max_modes ? *n_modes : max_modes;
    // height ratio law h_l_rgb (bump_DP.cpp:235-254) with the vector's Hfactor (the l=1 block's 8th entry, models.cpp:4760)
    const double Hfactor = std::fabs(params[plength[0] + plength[1] + plength[2] + 7]);
    for (int i = 0; i < n; i++) {
        if (zeta) zeta[i] = z[(size_t)i];
        if (h1_h0) {
            double hr = std::sqrt(1. - Hfactor * z[(size_t)i]);
            if (hr > -1e-5 && hr < 1e-5) hr = 1e-10;
            h1_h0[i] = hr;
        }
    }
    return TAMCMC_OK;
}

int tamcmc_hip_get_kernel_stats(tamcmc_hip_ctx *c, double *kernel_ms_total, int64_t *launches, int64_t *evaluations) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (kernel_ms_total) *kernel_ms_total = c->kernel_ms;
    if (launches) *launches = c->launches;
    if (evaluations) *evaluations = c->evals;
    return TAMCMC_OK;
}

int tamcmc_hip_get_fd_stats(tamcmc_hip_ctx *c, int64_t *affected_bins, int64_t *delta_evaluations) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (affected_bins) *affected_bins = c->fd_bins;
    if (delta_evaluations) *delta_evaluations = c->fd_delta_evals;
    return TAMCMC_OK;
}

int tamcmc_hip_get_fd_full_tables(tamcmc_hip_ctx *c, int64_t *full_table_evaluations) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    if (full_table_evaluations) *full_table_evaluations = c->fd_full_evals;
    return TAMCMC_OK;
}

int tamcmc_hip_reset_kernel_stats(tamcmc_hip_ctx *c) {
    if (!c) return TAMCMC_ERR_BAD_ARG;
    c->kernel_ms = 0;
    c->launches = 0;
    c->evals = 0;
    c->fd_bins = 0;
    c->fd_delta_evals = 0;
    c->fd_full_evals = 0;
    return TAMCMC_OK;
}

}  // extern "C"
