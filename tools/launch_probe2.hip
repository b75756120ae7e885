// launch_probe2.hip -- can two kernels of ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (AQL barrier bit
// clear)?  Pattern per iteration: C (ordered: waits for everything before it), T (any-order: may start while C runs).  (scratch tool)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(long ticks, long *sink, int slot) {
    const long t0 = wall_clock64();
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[slot] = t0;
    while ((long)wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[slot + 1] = (long)wall_clock64();
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int NIT = argc > 1 ? atoi(argv[1]) : 2000;
    const long TL = argc > 2 ? atol(argv[2]) : 2700, TB = argc > 3 ? atol(argv[3]) : 1500;
    hipStream_t s1;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    long *sink;
    CK(hipMalloc(&sink, 4096 * sizeof(long)));
    CK(hipMemset(sink, 0, 4096 * sizeof(long)));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_spin, dim3(3920), dim3(64), 0, s1, 100, (long *)nullptr, 0);
    CK(hipStreamSynchronize(s1));
    for (int mode = 0; mode < 2; mode++) {
        double t0 = now();
        for (int i = 0; i < NIT; i++) {
            const bool rec = (i >= 100 && i < 104);
            hipExtLaunchKernelGGL(k_spin, dim3(20), dim3(256), 0, s1, nullptr, nullptr, 0, TB, rec ? sink : (long *)nullptr, (i - 100) * 4);
            hipExtLaunchKernelGGL(k_spin, dim3(3920), dim3(64), 0, s1, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, TL, rec ? sink : (long *)nullptr,
                                  (i - 100) * 4 + 2);
        }
        double t1 = now();
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        long h[16];
        CK(hipMemcpy(h, sink, sizeof h, hipMemcpyDeviceToHost));
        printf("%s: %.2f us/iter (host enqueue %.2f), C %.1f us, T %.1f us\n", mode ? "C ordered + T any-order" : "both ordered          ", (t2 - t0) / NIT * 1e6,
               (t1 - t0) / NIT * 1e6, TB * 0.01, TL * 0.01);
        for (int k = 0; k < 4; k++)
            printf("    it %d: C [%.2f, %.2f]  T [%.2f, %.2f] us (relative to the first C start)\n", k, (h[4 * k] - h[0]) * 0.01, (h[4 * k + 1] - h[0]) * 0.01,
                   (h[4 * k + 2] - h[0]) * 0.01, (h[4 * k + 3] - h[0]) * 0.01);
    }
    return 0;
}
