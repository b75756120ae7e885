// ubench.hip -- fp64 instruction micro-benchmarks on gfx950 (scratch tool, not product code):
// accuracy of v_rcp_f64, and issue rates of v_fma_f64 / v_rcp_f64 / IEEE divide / log / exp per wave.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>

__global__ void k_rcp_acc(const double* in, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcp(in[i]);
}
template <int MODE>
__global__ void k_rate(double* out, double seed, int iters) {
    double a = seed + threadIdx.x * 1e-3, b = 1.0000001, c = 0.5, d = a * 0.3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) { a = fma(a, b, c); d = fma(d, b, c); }
            if (MODE == 1) { a = __builtin_amdgcn_rcp(a) + 1.0; d = __builtin_amdgcn_rcp(d) + 1.0; }
            if (MODE == 2) { a = c / a + 1.0; d = c / d + 1.0; }
            if (MODE == 3) { a = log(a) + 2.0; d = log(d) + 2.0; }
            if (MODE == 4) { a = exp(a) * 1e-3 + 0.1; d = exp(d) * 1e-3 + 0.1; }
            if (MODE == 5) { a = a + b; d = d + b; }
            if (MODE == 6) { a = a * b; d = d * b; }
            if (MODE == 7) { a = pow(a, 1.7) * 1e-2 + 1.0; d = pow(d, 1.7) * 1e-2 + 1.0; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
}
template <int MODE>
void rate(const char* name, int ops_per_iter) {
    double* out; hipMalloc(&out, 256 * 8 * 256 * 8 * sizeof(double));
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.3, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.3, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 8 * ops_per_iter;
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double wave_instr = ops / 64.0;
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;
    printf("%-10s %8.3f ms  %.3e lane-ops/s  ~%.1f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, ms, ops / (ms * 1e-3), cyc);
    hipFree(out);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), r(n);
    std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-300, 300);
    for (auto& v : h) v = std::pow(10.0, u(g) * 0.5) * (g() & 1 ? 1 : -1);
    double *di, *dr; hipMalloc(&di, n * 8); hipMalloc(&dr, n * 8);
    hipMemcpy(di, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp_acc, dim3(n / 256), dim3(256), 0, 0, di, dr, n);
    hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    double maxrel = 0;
    for (int i = 0; i < n; i++) { double e = std::fabs(r[i] * h[i] - 1.0); if (e > maxrel) maxrel = e; }
    printf("v_rcp_f64 max |x*rcp(x)-1| over %d samples: %.3e (2^%.1f)\n", n, maxrel, std::log2(maxrel));
    rate<0>("fma", 2); rate<5>("add", 2); rate<6>("mul", 2); rate<1>("rcp+add", 2); rate<2>("div+add", 2);
    rate<3>("log+add", 2); rate<4>("exp+fma", 2); rate<7>("pow+fma", 2);
    return 0;
}
