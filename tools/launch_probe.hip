// launch_probe.hip -- how much does the HOST / the command processor cost per kernel for the launch patterns an MCMC iteration
// could use?  (scratch tool, not product code.)  Kernels spin for a fixed time on the 100 MHz wall clock.
//   A  one stream, back-to-back launches of L (20 us, 3920 single-wave workgroups)
//   B  two streams with cross events: Br(j) after L(j-2); L(j) after Br(j)   (6 runtime calls per iteration)
//   C  pattern B captured once into a graph of NIT iterations, replayed
//   D  one-stream graph of NIT sequential L
//   E  one stream, ONE fused launch per iteration (L and Br workgroups in the same grid)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(long ticks, long *sink) {
    const long t0 = wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = t0;
}
__global__ void k_fused(long ticksL, long ticksB, int nL, long *sink) {
    const long t0 = wall_clock64();
    const long ticks = ((int)blockIdx.x < nL) ? ticksL : ticksB;
    while ((long)wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = t0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int NIT = argc > 1 ? atoi(argv[1]) : 2000;
    const long TL = argc > 2 ? atol(argv[2]) : 2000, TB = argc > 3 ? atol(argv[3]) : 1200;  // ticks of 10 ns
    const int GL = 3920, GB = 44;
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    long *sink;
    CK(hipMalloc(&sink, 64));
    hipEvent_t eL[4], eB[4];
    for (int i = 0; i < 4; i++) { CK(hipEventCreateWithFlags(&eL[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eB[i], hipEventDisableTiming)); }
    auto warm = [&]() { for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_spin, dim3(GL), dim3(64), 0, s1, 100, sink); CK(hipStreamSynchronize(s1)); };
    warm();
    // A
    {
        double t0 = now();
        for (int i = 0; i < NIT; i++) hipLaunchKernelGGL(k_spin, dim3(GL), dim3(64), 0, s1, TL, sink);
        double t1 = now();
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        printf("A one stream          : %.2f us/iter (host enqueue %.2f us/iter), kernel %.1f us\n", (t2 - t0) / NIT * 1e6, (t1 - t0) / NIT * 1e6, TL * 0.01);
    }
    // B
    auto patternB = [&](int n, hipStream_t a, hipStream_t b) {
        for (int j = 0; j < n; j++) {
            if (j >= 2) CK(hipStreamWaitEvent(b, eL[(j - 2) & 3], 0));
            hipLaunchKernelGGL(k_spin, dim3(GB), dim3(256), 0, b, TB, sink + 1);
            CK(hipEventRecord(eB[j & 3], b));
            CK(hipStreamWaitEvent(a, eB[j & 3], 0));
            hipLaunchKernelGGL(k_spin, dim3(GL), dim3(64), 0, a, TL, sink);
            CK(hipEventRecord(eL[j & 3], a));
        }
    };
    {
        warm();
        double t0 = now();
        patternB(NIT, s1, s2);
        double t1 = now();
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        double t2 = now();
        printf("B two streams + events : %.2f us/iter (host enqueue %.2f us/iter)\n", (t2 - t0) / NIT * 1e6, (t1 - t0) / NIT * 1e6);
    }
    // C: graph of G iterations of pattern B
    for (int G : {8, 32}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipEvent_t ef, ej;
        CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        CK(hipEventRecord(ef, s1));
        CK(hipStreamWaitEvent(s2, ef, 0));
        patternB(G, s1, s2);
        CK(hipEventRecord(ej, s2));
        CK(hipStreamWaitEvent(s1, ej, 0));
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s1));
        CK(hipStreamSynchronize(s1));
        const int reps = NIT / G;
        double t0 = now();
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s1));
        double t1 = now();
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        printf("C graph(%2d) of pattern B: %.2f us/iter (host enqueue %.2f us/iter)\n", G, (t2 - t0) / (reps * G) * 1e6, (t1 - t0) / (reps * G) * 1e6);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // D: one-stream graph
    for (int G : {8, 32}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        for (int j = 0; j < G; j++) hipLaunchKernelGGL(k_spin, dim3(GL), dim3(64), 0, s1, TL, sink);
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s1));
        CK(hipStreamSynchronize(s1));
        const int reps = NIT / G;
        double t0 = now();
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s1));
        double t1 = now();
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        printf("D graph(%2d) one stream : %.2f us/iter (host enqueue %.2f us/iter)\n", G, (t2 - t0) / (reps * G) * 1e6, (t1 - t0) / (reps * G) * 1e6);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // E: fused launch (L workgroups as 256-thread blocks of four tiles + Br blocks)
    {
        warm();
        double t0 = now();
        for (int i = 0; i < NIT; i++) hipLaunchKernelGGL(k_fused, dim3(GL / 4 + GB), dim3(256), 0, s1, TL, TB, GL / 4, sink);
        double t1 = now();
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        printf("E fused, one stream    : %.2f us/iter (host enqueue %.2f us/iter)\n", (t2 - t0) / NIT * 1e6, (t1 - t0) / NIT * 1e6);
    }
    // F: short windows: 20 iterations + sync, repeated (what a 20-step bench call sees)
    {
        warm();
        double best = 1e9;
        for (int r = 0; r < 20; r++) {
            double t0 = now();
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_fused, dim3(GL / 4 + GB), dim3(256), 0, s1, TL, TB, GL / 4, sink);
            CK(hipStreamSynchronize(s1));
            double t2 = now();
            if (t2 - t0 < best) best = t2 - t0;
        }
        printf("F 20 fused launches + sync: best %.2f us/iter\n", best / 20 * 1e6);
    }
    return 0;
}
