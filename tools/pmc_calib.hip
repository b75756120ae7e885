// pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS path's access width (MI355X_MICROARCH.md, HBM section:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  k_loglike reads x and y as 8-byte-per-lane coalesced loads: this kernel
// streams a known number of bytes the same way (and, second kernel, 16 B/lane for comparison).  (scratch tool, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_calib_read8(const double *p, size_t n, double *out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void k_calib_read16(const double2 *p, size_t n, double *out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB: far beyond L2 and the 256 MiB Infinity Cache
    double *p, *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_calib_read8, dim3(4096), dim3(256), 0, 0, p, bytes / 8, out);
        hipLaunchKernelGGL(k_calib_read16, dim3(4096), dim3(256), 0, 0, (const double2 *)p, bytes / 16, out);
    }
    hipDeviceSynchronize();
    printf("calib: each launch reads %zu bytes\n", bytes);
    return 0;
}
