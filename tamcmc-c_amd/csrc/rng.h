// rng.h -- seedable counter-based RNG (Philox4x32-10, Salmon et al. 2011) shared by host and device code.
// Replaces the reference's libc rand()/time(NULL) seeding and its racy static Box-Muller state
// (MALA.cpp:62-63, random_JB.cpp:99-105,255): every draw is a pure function of (seed, stream, counter),
// so host and device samplers consume identical random numbers and runs are reproducible.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TAMCMC_HD __host__ __device__ inline
#else
#define TAMCMC_HD inline
#endif

namespace tamcmc {

struct Philox4 { uint32_t v[4]; };

TAMCMC_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

// counter = (c0..c3), key = (k0,k1); 10 rounds
TAMCMC_HD Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        philox_mulhilo(M0, c0, hi0, lo0);
        philox_mulhilo(M1, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// 53-bit uniform in (0,1): never 0, never 1
TAMCMC_HD double u01_from_bits(uint32_t hi, uint32_t lo) {
    const uint64_t b = (((uint64_t)hi << 32) | lo) >> 11;  // 53 bits
    return ((double)b + 0.5) * (1.0 / 9007199254740992.0);
}

// Draw addressing: (seed) x (purpose, chain) x (iteration, index).
enum RngPurpose : uint32_t { RNG_PROPOSAL = 1, RNG_ACCEPT = 2, RNG_SWAP = 3, RNG_NOISE = 4 };

// two independent uniforms for (purpose, chain, iteration, pair index)
TAMCMC_HD void rng_uniform2(uint64_t seed, uint32_t purpose, uint32_t chain, uint64_t iter, uint32_t idx, double &u0,
                            double &u1) {
    const Philox4 r = philox4x32((uint32_t)iter, (uint32_t)(iter >> 32), idx, (purpose << 16) | (chain & 0xffffu),
                                 (uint32_t)seed, (uint32_t)(seed >> 32));
    u0 = u01_from_bits(r.v[0], r.v[1]);
    u1 = u01_from_bits(r.v[2], r.v[3]);
}

// two independent standard normals (Box-Muller on one Philox block)
TAMCMC_HD void rng_normal2(uint64_t seed, uint32_t purpose, uint32_t chain, uint64_t iter, uint32_t idx, double &z0,
                           double &z1) {
    double u0, u1;
    rng_uniform2(seed, purpose, chain, iter, idx, u0, u1);
    const double r = sqrt(-2.0 * log(u0));
    const double a = 6.283185307179586476925286766559 * u1;
    z0 = r * cos(a);
    z1 = r * sin(a);
}

}  // namespace tamcmc
