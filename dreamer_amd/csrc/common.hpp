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
                 PC_GATHER, PC_SAMPLE, PC_PACK, PC_LINEAR_BIG, PC_ELEMENTWISE };
bool prof_on();
void prof_begin(int cls, hipStream_t s, double flops, double bytes);
void prof_end(hipStream_t s);
void prof_fresh();   // the next launch starts a new back-to-back run (gets a start event of its own)
struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(int cls, hipStream_t st, double flops, double bytes) : s(st), on(prof_on()) { if (on) prof_begin(cls, s, flops, bytes); }
    ~ProfScope() { if (on) prof_end(s); }
};

// ---- diagnostic build only (-DDRM_STAMPS): wall-clock stamps of kernel phases ------------------
// Block (0,0), lane 0 of wave 0 appends {kernel id, 100 MHz wall-clock ticks per phase} to a buffer of
// its own that nothing else reads.  Never compiled into the shipped library.
#ifdef DRM_STAMPS
static __device__ unsigned long long* g_stamp_buf = nullptr;   // per translation unit; [0] = record count, then 8 words per record
#define STAMP_DECL unsigned long long _st[6]; int _sn = 0; const bool _stamper = (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
#define STAMP() do { if (_stamper && _sn < 6) _st[_sn++] = wall_clock64(); } while (0)
#define STAMP_FLUSH(kid) do { if (_stamper && g_stamp_buf) { unsigned long long r = atomicAdd(&g_stamp_buf[0], 1ull); \
        if (r < 100000) { unsigned long long* o = g_stamp_buf + 1 + r * 8; o[0] = (kid); o[1] = _sn | ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 8); for (int i = 0; i < 6; ++i) o[2 + i] = i < _sn ? _st[i] : 0; } } } while (0)
#else
#define STAMP_DECL
#define STAMP() do {} while (0)
#define STAMP_FLUSH(kid) do {} while (0)
#endif

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float LN_EPS = 1e-5f;

// ---- device math, written to round the way the fp32 reference ops do --------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
#ifdef DRM_ACCURATE_SILU
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
#else
// SiLU on the hardware exp2 / reciprocal units (each within 1 ulp): x * rcp(1 + 2^(-x*log2 e)).
// The activation is recomputed by every consumer workgroup of a layer, so its instruction count
// sits on the latency-critical path; the error (a few 1e-7 relative) is far inside the 1e-5 budget.
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
#endif
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
