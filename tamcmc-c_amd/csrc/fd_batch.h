// fd_batch.h -- one finite-difference batch (C chains x (Nvars + 1) evaluations) as an object: layout of its device block, launch from
// parameter vectors already on the device (fd_batch.hip).  Used by the host entry points (tamcmc_hip_fd_gradient*) and by the
// device-resident Langevin step (dev_mala.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ctx.h"

namespace tamcmc {

struct FdBatch {
    int model_id = 0, prior_class = 0, C = 0, E = 0, B = 0, Nvars = 0, per = 0, stride = 1, ntiles = 0;
    int64_t Np = 0;
    bool windowed = false;
    // offsets inside the device block: host-filled constants [0, in_bytes), results [in_bytes, in_bytes + out_bytes), tables after
    size_t o_params = 0, o_h = 0, o_pr = 0, o_ex = 0, o_pl = 0, o_idx = 0, o_sw = 0, in_bytes = 0;
    size_t o_lpp = 0, o_lpm = 0, o_st = 0, out_bytes = 0;
    size_t o_tab = 0, o_dtab = 0, o_btab = 0, o_drange = 0, o_dflags = 0, o_drow = 0, o_dnold = 0, total_bytes = 0;
    size_t nS = 0;  // sums the batch produces: C base sums + B differences (windowed) or B full sums
    int layout(tamcmc_hip_ctx *c, int model_id, int prior_class, int C, int64_t Nparams, const int32_t *plength, int Nvars);
    int enqueue(tamcmc_hip_ctx *c, unsigned char *block, const double *d_params, double *part, double *S, double *model, double *bgbuf,
                hipEvent_t ev0, hipEvent_t ev1);
    // [B x ntiles] flags of the last enqueue (device memory, inside `model`): 1 = that (evaluation, tile) was a far-only tile taken from
    // the base point's moments -- no bin of it was read; nullptr: no such pass (roofline bookkeeping: bins_not_walked)
    const unsigned char *d_done = nullptr;
    int tile_bins_ = 0;
    long bins_not_walked() const;  // (synchronous copy; call after the batch has finished)
};
int fd_ensure_poly(tamcmc_hip_ctx *c);  // Pslm/Qlm tables in c->d_poly

}  // namespace tamcmc
