// Shared helpers for libdreamer_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <string>

#include "../../include/dreamer_amd.h"

namespace drm {

void set_error(const std::string& msg);

#define DRM_REQUIRE(cond, code, ...)                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            char _b[512];                                              \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                     \
            ::drm::set_error(std::string(__func__) + ": " + _b);       \
            return (code);                                             \
        }                                                              \
    } while (0)

#define DRM_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::drm::set_error(std::string(__func__) + ": " #expr " -> " + hipGetErrorString(_e)); \
            return DRM_EHIP;                                                               \
        }                                                                                  \
    } while (0)

#define DRM_LAUNCH_CHECK() DRM_HIP(hipGetLastError())

// kernel classes for the optional per-launch timing
enum ProfClass { PC_LINEAR = 0, PC_GRU, PC_DYN_SAMPLE, PC_ACTOR_HEAD, PC_SCALAR_HEADS, PC_CRITIC_VALUE, PC_LAMBDA,
                 PC_GATHER, PC_SAMPLE, PC_PACK };
bool prof_on();
void prof_begin(int cls, hipStream_t s, double flops, double bytes);
void prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(int cls, hipStream_t st, double flops, double bytes) : s(st), on(prof_on()) { if (on) prof_begin(cls, s, flops, bytes); }
    ~ProfScope() { if (on) prof_end(s); }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float LN_EPS = 1e-5f;

// ---- device math, written to round the way the fp32 reference ops do --------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float symexp_f(float x) { return copysignf(expm1f(fabsf(x)), x); }

// log(exp(a) + exp(b))
__device__ __forceinline__ float logaddexp_f(float a, float b) {
    float m = fmaxf(a, b);
    return m + log1pf(expf(-fabsf(a - b)));
}

// reductions over the `width` consecutive lanes this lane belongs to (width a power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WIDTH));
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WIDTH);
    return v;
}
// argmax with first-index tie break over a lane group
template <int WIDTH>
__device__ __forceinline__ void group_argmax(float& v, int& i) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, WIDTH);
        int oi = __shfl_xor(i, o, WIDTH);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

}  // namespace drm
