// kernels.h -- launch interface between the C-ABI layer (capi.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tamcmc_hip.h"

#define TAMCMC_MAX_HARVEY 16

namespace tamcmc {

struct LoglikeArgs {
    const double *x;      // [Nx] frequencies, resident
    const double *y;      // [Nx] power, resident
    const double *logx;   // [Nx] ln(x) (FAST Harvey terms), resident
    int Nx;
    double x0, step;      // regular grid: x[i] = x0 + i*step (far-field tile geometry)
    int B;                // evaluations in this launch
    int ntiles;           // filled by launch_loglike
#ifdef TAMCMC_PROBE        // the probe build only (make probe -> libtamcmc_hip_probe.so, used by tools/): never in the product library
    int probe = 0;        // bit mask of kernel phases to skip -- results are then wrong (tools/phase_probe.py)
    long *dbg = nullptr;  // phase stamps of one workgroup / per-workgroup timeline
#endif
    int tile_rot = 0;     // tile dispatched first (launch order wraps around); any value in [0, ntiles) gives the same results
    int prio_b = -1;      // >= 0: evaluations prio_b and prio_b+1 are dispatched first (B >= 3); results do not depend on it
    const tamcmc_multiplet *mults;  // concatenated multiplet tables
    int per = 0, slot0 = 0;         // per > 0: evaluation b's table begins at row (slot0 + slot) * per (fixed-size slots, device-built tables)
    const int32_t *offsets;         // [2B] (begin,end) multiplet range per evaluation
    const double *noise;            // [B x noise_stride] |noise params|
    int noise_stride;
    const int32_t *nharvey;         // [B]
    const int32_t *nnoise;          // [B]
    double *partials;               // [B x ntiles x 2]
    double *model;                  // [B x Nx] or nullptr
    // FAST far field: [B x ntiles x 8] background series per (evaluation, tile) built with the table (bg_series.h);
    // nullptr -> every tile's workgroup computes its own (same arithmetic, same result)
    const double *bg_poly = nullptr;
    // DELTA launches (windowed finite differences, fd_batch.hip): evaluation b's table holds the CHANGED multiplets only,
    // new rows with +H*V and old rows with -H*V, so the kernel accumulates dM = M(theta + h e_k) - M(theta); the partial
    // sums are those of the log-likelihood DIFFERENCE against the base model row, over the affected bins only.
    const int32_t *d_range = nullptr;     // [2B] affected bin range [lo, hi) of evaluation b
    const int32_t *d_flags = nullptr;     // [B]  bit 0: the noise parameters changed (background difference on every bin); bit 1: the table
                                          //      holds EVERY row of the perturbed point (+H*V): dM = M - M0 against model0's third plane
    const double *d_noise_old = nullptr;  // [B x noise_stride] |noise params| of the base point
    const int32_t *d_row = nullptr;       // [B]  row of model0 holding the base point of evaluation b
    // base point of the finite differences, three planes of [rows x Nx], fd_plane doubles apart: 1/M0, y/M0 and M0.  Written by the base launch
    // (WRITE_MODEL with fd_rows set, instead of the model rows), read by the DELTA launch (model0).
    const double *model0 = nullptr;
    double *fd_rows = nullptr;
    size_t fd_plane = 0;
    // DELTA, tiles whose changed multiplets are all in the far field: moments of the base point per (row, tile), FD_MOM doubles each
    // (launch_fd_moments), or nullptr: every such tile walks its bins
    const double *fd_mom = nullptr;
    const double *fd_momT = nullptr;  // the same moments as [rows x FD_MOM x ntiles] (launch_fd_far: one lane per tile reads along the tiles)
    // DELTA: [B x ntiles] 1 = this (evaluation, tile) is already done (launch_fd_far wrote its partial sums): the workgroup leaves at once
    const unsigned char *d_done = nullptr;
};
constexpr int FD_MOM = 48;  // 16 first-order + 31 second-order moments + max 1/M0 (loglike_tile.h, DELTA far-only tiles)

// Tile to dispatch first: three tiles below the lowest multiplet centre of a representative table (so the near-field tiles lead
// the launch).  A hint only -- results do not depend on it.
inline int pick_tile_rot(const tamcmc_multiplet *m, int n, double x0, double step, int tile, int ntiles) {
    double fmin = 0.0;
    bool any = false;
    for (int i = 0; i < n; i++)
        if (m[i].fc == m[i].fc && (!any || m[i].fc < fmin)) { fmin = m[i].fc; any = true; }
    if (!any || !(step > 0.0) || ntiles < 1) return 0;
    const double t = (fmin - x0) / step / (double)tile - 3.0;
    if (!(t > 0.0)) return 0;
    return t >= (double)ntiles ? ntiles - 1 : (int)t;
}

int tile_bins(int wgs, int K);       // bins per workgroup = workgroup size x bins per thread
bool valid_geometry(int wgs, int K);  // (256; 1,2,4) or (64; 4,8,16)
bool delta_geometry(int wgs, int K);  // geometries the DELTA variant is instantiated for: (256,4), (64,8)
// mode = TAMCMC_PRECISION_* (0 strict, 1 fast = far-field expansion + direct near field, 2 fast without the far field)
hipError_t launch_loglike(LoglikeArgs a, int mode, int wgs, int K, bool write_model, hipStream_t st);
// DELTA variant (FAST modes only): a.d_* / a.model0 must be set
hipError_t launch_loglike_delta(LoglikeArgs a, int mode, int wgs, int K, hipStream_t st);
// fills bg[B x ntiles x 8] for launch_loglike(a with a.bg_poly = bg, FAST mode, same wgs/K): one thread per (evaluation, tile)
hipError_t launch_bg_poly(const LoglikeArgs &a, int wgs, int K, double *bg, hipStream_t st);
hipError_t launch_finalize(const double *partials, int B, int ntiles, double *S, hipStream_t st);
// moments of the base points of a finite-difference batch for the DELTA launch's far-only tiles: a.fd_rows (the planes the base launch
// left), a.B rows, geometry (wgs, K) of the DELTA launch -> mom[B x ntiles x FD_MOM] and its transpose momT[B x FD_MOM x ntiles]
hipError_t launch_fd_moments(const LoglikeArgs &a, int wgs, int K, double *mom, double *momT, hipStream_t st);
// DELTA launch arguments `d` (delta tables, a.fd_mom set): every (evaluation, tile) whose changed multiplets are ALL in the tile's far
// field is evaluated here, one lane per tile -- its partial sums written, done[b x ntiles] set to 1 -- for the evaluations with few
// rows (a perturbed frequency, width or height: most tiles of their windows); everything else gets done = 0 and is the DELTA launch's.
hipError_t launch_fd_far(const LoglikeArgs &d, int wgs, int K, unsigned char *done, hipStream_t st);

}  // namespace tamcmc
