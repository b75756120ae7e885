// ctx.h -- private definition of the opaque context of include/tamcmc_hip.h (shared by capi.hip and dev_sampler.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/tamcmc_hip.h"

namespace tamcmc {

template <typename T>
struct DevBuf {  // grow-only device buffer
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 2 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinBuf {  // grow-only pinned host buffer
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 2 + 64;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace tamcmc

struct tamcmc_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // options
    int precision = TAMCMC_PRECISION_STRICT;
    int timing = 0;
    int wgs = 256;  // workgroup size of k_loglike: 256 (4 waves per tile) or 64 (one wave per tile)
    int K = 4;      // bins per thread: tile = wgs*K bins
    bool geom_user_set = false;
    int fd_windowed = 1;  // FAST modes: finite differences through delta tables (changed multiplets on their windows only)
    int step_scheme = 0;  // device sampler: 0 = fused steps where possible (one or two launches per iteration: automatic), 1 = lockstep kernels only,
                          // 2 = fused, always one launch per iteration, 3 = fused, two chain groups whenever the chain count allows
    int armm_dense = 0;   // red-giant pre-step: 1 = dense grid walk
    // resident spectrum
    int64_t Nx = 0;
    std::vector<double> hx;  // host copy of x (table builders need x[0], x[Nx-1], step)
    tamcmc::DevBuf<double> dx, dy, dlogx;
    // ONE pinned staging block and its device image per call (a single H2D copy):
    //   [int32 begin/end pairs 2B | int32 nharvey B | int32 nnoise B | pad] [double noise B*stride] [multiplets]
    tamcmc::PinBuf<unsigned char> h_stage;
    tamcmc::DevBuf<unsigned char> d_stage;
    tamcmc::DevBuf<double> d_part, d_S, d_model;
    tamcmc::DevBuf<unsigned char> d_rgb;  // red-giant pre-step workspace (rgb_prestep.hip)
    tamcmc::PinBuf<unsigned char> h_rgb;  // its pinned host image: [Prep B | RowIn B] going up, [int status B] coming back
    tamcmc::DevBuf<double> d_bg;  // FAST far field: background series per (evaluation, tile) (bg_series.h)
    // finite-difference batches built on the device (fd_batch.hip)
    tamcmc::DevBuf<unsigned char> d_fd, d_poly;
    tamcmc::PinBuf<unsigned char> h_fd;
    bool poly_ready = false;
    tamcmc::PinBuf<double> h_S;
    // samplers created on this context (they borrow its stream and buffers): tamcmc_hip_destroy with samplers still attached only
    // marks the context; the last tamcmc_sampler_destroy frees it -- any destruction order is safe
    int attached = 0;
    bool zombie = false;
    // stats
    double kernel_ms = 0;
    int64_t launches = 0, evals = 0;
    int64_t fd_full_evals = 0;                // ... of which "full table" evaluations (fd_batch.hip)
    int64_t fd_bins = 0, fd_delta_evals = 0;  // windowed finite differences (timing on): bins inside the affected ranges, delta evaluations
};

namespace tamcmc {
struct StageLayout {
    size_t off_pairs, off_nh, off_nn, off_noise, off_mults, bytes;
    StageLayout(int B, int stride, size_t total_mults) {
        off_pairs = 0;
        off_nh = off_pairs + (size_t)2 * B * sizeof(int32_t);
        off_nn = off_nh + (size_t)B * sizeof(int32_t);
        off_noise = (off_nn + (size_t)B * sizeof(int32_t) + 15) & ~(size_t)15;
        off_mults = (off_noise + (size_t)B * stride * sizeof(double) + 15) & ~(size_t)15;
        bytes = off_mults + (total_mults + 1) * sizeof(tamcmc_multiplet);
    }
};
}  // namespace tamcmc

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return TAMCMC_ERR_HIP;                                                            \
        }                                                                                     \
    } while (0)
