// rgb_prestep.h -- red-giant models (ids 25, 27) table builder with its device pre-step (rgb_prestep.hip).
#pragma once
#include <cstdint>

#include "ctx.h"
#if defined(__HIPCC__)
#include "rgb_unpack.h"
#endif

#define TAMCMC_MODEL_RGB_ASYMPT_AJ_APPWIDTH_V4_ID 25
#define TAMCMC_MODEL_RGB_ASYMPT_AJ_CTEWIDTH_V4_ID 27

namespace tamcmc {
// Builds the B variable-length tables in the DEVICE staging block c->d_stage (layout StageLayout(B, stride, B*per)); status[b] per vector;
// tile_rot = launch-order hint for k_loglike.
int rgb_stage_params(tamcmc_hip_ctx *c, int model_id, int B, const double *params, int64_t Nparams, const int32_t *plength, int32_t *status,
                     int *per_out, int *stride_out, int *first_err, int *tile_rot_out);
// after rgb_stage_params(B) + a stream synchronisation: vector b's mixed-mode frequencies and normalised zeta (at most max_modes each)
int rgb_fetch_modes(tamcmc_hip_ctx *c, int B, int b, int max_modes, double *nu_m, double *zeta, int *n_out);
// after the caller's stream synchronisation: device-side status words -> status[] / first_err
void rgb_collect_status(tamcmc_hip_ctx *c, int B, int32_t *status, int *first_err);
#if defined(__HIPCC__)
// Device engine: the pre-step on parameter vectors already in device memory, tables written into the engine's likelihood input block
// (all arrays indexed by chain).  status: each vector's final status (the row builder merges what the unpack left in Prep / RowIn).
struct RgbDeviceTables {
    tamcmc_multiplet *mults;
    int *pairs, *nh, *nn;
    double *noise;
    int *status;
    int stride = 1;
    double *bg = nullptr;  // [chain][tile][8] background series of the FAST far field, or nullptr
    int ntiles = 0, tile_bins = 0;
};
// prepare: workspace for `slices` chain groups of at most Bmax vectors each (once, outside the iteration loop).
// slice:   what the proposal kernel needs to run the scalar unpack of its chain into workspace slice `slice` (rgb_unpack.h).
// stage:   chains [b0, b0 + B) through workspace slice `slice`: solver + zeta normalisation, then sort / zeta / rows / background, on st.
int rgb_device_prepare(tamcmc_hip_ctx *c, int Bmax, int slices, const int32_t *plength, int *per_out, int *stride_out);
rgb::Slice rgb_device_slice(tamcmc_hip_ctx *c, int Bmax, int slice);
int rgb_device_stage(tamcmc_hip_ctx *c, int b0, int B, int Bmax, int slice, int per, const RgbDeviceTables &T, hipStream_t st);
#endif
}  // namespace tamcmc
