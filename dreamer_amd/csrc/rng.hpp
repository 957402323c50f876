// Counter-based random numbers for the draws inside the imagination kernels (lib/world_model.py:459-461 and
// lib/trainer.py:190-191 draw Exp(1) noise through torch's generator; here the same distribution comes from
// Philox-4x32-10, Salmon et al. 2011, keyed by a seed and addressed by (call counter, kind of draw, element)).
// A value depends on nothing but its address, so every workgroup that repeats a trajectory's draw obtains the same
// numbers, and a recorded graph draws fresh numbers on every replay because the call counter lives in device memory.
#pragma once
#include "common.hpp"

namespace drm {

struct RngArgs {
    uint32_t seed_lo, seed_hi;
    const uint64_t* state;      // device memory holding the call counter, or nullptr: `counter` below
    uint64_t counter;
    uint32_t t;                 // time step of the launch
    int32_t rows_total, row0;   // rows of a time step in the whole call, first row of this launch's trajectories
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}

// Exp(1) from 23 random bits: u = (bits + 1/2) 2^-23 lies in [2^-24, 1 - 2^-24], so the result is in [6e-8, 16.7]:
// never zero (the draws divide by it) and never infinite
// (hardware base-2 logarithm: the draw sits on the latency-critical path of its kernel, and a sample of a noise
// distribution needs no more than the unit's ~1 ulp)
__device__ __forceinline__ float exp1_from_bits(uint32_t x) {
    const float u = ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return fmaxf(__builtin_amdgcn_logf(u) * -0.69314718055994531f, 1.0e-30f);
}

// four consecutive Exp(1) values: group `idx` of stream `kind` (0: latent draws, 1: action draws) of this call
__device__ __forceinline__ float4 rng_exp4(const RngArgs& R, uint32_t kind, uint64_t idx) {
    const uint64_t ctr = R.state != nullptr ? *R.state : R.counter;
    const uint4 r = philox4x32_10(make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32) ^ (kind << 31)),
                                  R.seed_lo, R.seed_hi);
    return make_float4(exp1_from_bits(r.x), exp1_from_bits(r.y), exp1_from_bits(r.z), exp1_from_bits(r.w));
}

}  // namespace drm
