// rgb_prestep.h -- red-giant models (ids 25, 27) table builder with its device pre-step (rgb_prestep.hip).
#pragma once
#include <cstdint>

#include "ctx.h"

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
}  // namespace tamcmc
