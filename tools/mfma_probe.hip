// How does v_mfma_f64_16x16x4_f64 round?  D = C + sum_k A[i][k] B[k][j], k = 0..3, compared bit for bit with candidate host-style
// evaluation orders (sequential fused multiply-adds in k order starting from C, the reverse order, unfused).  hipcc --offload-arch=gfx950
// tools/mfma_probe.hip -o tools/mfma_probe && ./tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef LAYOUT
#define LAYOUT 1
#endif
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k(const double *A, const double *B, const double *C, double *D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // A[i][k], i = l % 16, k = l / 16
    const double b = B[(l / 16) * 16 + l % 16];  // B[k][j], k = l / 16, j = l % 16
    double4_t c;
    for (int r = 0; r < 4; r++) c[r] = C[(LAYOUT ? 4 * r + l / 16 : 4 * (l / 16) + r) * 16 + l % 16];
    double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(LAYOUT ? 4 * r + l / 16 : 4 * (l / 16) + r) * 16 + l % 16] = d[r];
}

int main() {
    std::vector<double> A(64), B(64), C(256), D(256);
    srand(1);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 8 - 4); };
    long bad_seq = 0, bad_rev = 0, bad_unf = 0, bad_pair = 0, n = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
    for (int trial = 0; trial < 200; trial++) {
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        for (auto &v : C) v = rnd();
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double s = C[i * 16 + j], r = C[i * 16 + j], u = C[i * 16 + j];
                for (int kk = 0; kk < 4; kk++) s = std::fma(A[i * 4 + kk], B[kk * 16 + j], s);
                for (int kk = 3; kk >= 0; kk--) r = std::fma(A[i * 4 + kk], B[kk * 16 + j], r);
                for (int kk = 0; kk < 4; kk++) { volatile double p = A[i * 4 + kk] * B[kk * 16 + j]; u = u + p; }
                double p2 = std::fma(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]) + std::fma(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]);
                p2 = p2 + C[i * 16 + j];
                const double d = D[i * 16 + j];
                n++;
                if (d != s) bad_seq++;
                if (d != r) bad_rev++;
                if (d != u) bad_unf++;
                if (d != p2) bad_pair++;
            }
    }
    printf("elements %ld: differs from sequential fma (k ascending from C) %ld, descending %ld, unfused %ld, pairwise %ld\n", n, bad_seq, bad_rev, bad_unf, bad_pair);
    return 0;
}
