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
    const tamcmc_multiplet *mults;  // concatenated multiplet tables
    const int32_t *offsets;         // [2B] (begin,end) multiplet range per evaluation
    const double *noise;            // [B x noise_stride] |noise params|
    int noise_stride;
    const int32_t *nharvey;         // [B]
    const int32_t *nnoise;          // [B]
    double *partials;               // [B x ntiles x 2]
    double *model;                  // [B x Nx] or nullptr
    long *dbg = nullptr;            // optional phase stamps of one workgroup (TAMCMC_DEBUG_STAMPS)
};

int tile_bins(int wgs, int K);       // bins per workgroup = workgroup size x bins per thread
bool valid_geometry(int wgs, int K);  // (256; 1,2,4) or (64; 4,8,16)
// mode = TAMCMC_PRECISION_* (0 strict, 1 fast = far-field expansion + direct near field, 2 fast without the far field)
hipError_t launch_loglike(LoglikeArgs a, int mode, int wgs, int K, bool write_model, hipStream_t st);
hipError_t launch_finalize(const double *partials, int B, int ntiles, double *S, hipStream_t st);

}  // namespace tamcmc
