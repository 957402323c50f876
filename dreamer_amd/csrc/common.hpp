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

// After every kernel launch.  With DRM_SYNC_CHECK=1 in the environment the device is synchronised behind each launch and a
// failure is reported on stderr with the launch site, so an asynchronous memory fault names the kernel that caused it
// (diagnostic mode: serialises everything); DRM_SYNC_CHECK=2 also prints every launch site before waiting for it.
int launch_check(const char* func, const char* file, int line);
// DRM_SYNC_CHECK (0: off).  A synchronising launch check cannot run inside a recording: the entry points that record
// graphs launch directly while it is on.
int sync_check_mode();
#define DRM_LAUNCH_CHECK()                                                       \
    do {                                                                         \
        if (int _rc = ::drm::launch_check(__func__, __FILE__, __LINE__)) return _rc; \
    } while (0)

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
#define STAMP_DECL_AT(cond) unsigned long long _st[6]; int _sn = 0; const bool _stamper = (cond) && threadIdx.x == 0;
#define STAMP_DECL STAMP_DECL_AT(blockIdx.x == 0 && blockIdx.y == 0)
#define STAMP() do { if (_stamper && _sn < 6) _st[_sn++] = wall_clock64(); } while (0)
#define STAMP_FLUSH(kid) do { if (_stamper && g_stamp_buf) { unsigned long long r = atomicAdd(&g_stamp_buf[0], 1ull); \
        if (r < 100000) { unsigned long long* o = g_stamp_buf + 1 + r * 8; o[0] = (kid); o[1] = _sn | ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 8) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 16) | ((unsigned long long)blockIdx.y << 48); for (int i = 0; i < 6; ++i) o[2 + i] = i < _sn ? _st[i] : 0; } } } while (0)
#else
#define STAMP_DECL_AT(cond)
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

// ---- cross-lane exchange ---------------------------------------------------------------------------
// Value of lane (this ^ M).  Partners up to 8 lanes away are reached with DPP modifiers on a plain
// vector move (no LDS crossbar round trip): quad permutes for 1 and 2, half-row mirror followed by a
// quad reverse for 4 (i -> 7-i -> (7-i)^3 = i^4), a row rotation by 8 for 8.  16 and 32 go through
// ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int M>
__device__ __forceinline__ float xor_lane(float v) {
    if constexpr (M == 1) return dpp_mov<0xB1>(v);                   // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);              // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // row_half_mirror, quad_perm [3,2,1,0]
    else if constexpr (M == 8) return dpp_mov<0x128>(v);             // row_ror:8
    else return __shfl_xor(v, M, 64);
}
template <int M>
__device__ __forceinline__ int xor_lane_i(int v) { return __builtin_bit_cast(int, xor_lane<M>(__builtin_bit_cast(float, v))); }

// reductions over the `width` consecutive lanes this lane belongs to (width a power of two <= 64);
// every lane of the group receives the result
#define DRM_LEVELS(OP)                                  \
    if constexpr (WIDTH >= 2) { OP(1) }                 \
    if constexpr (WIDTH >= 4) { OP(2) }                 \
    if constexpr (WIDTH >= 8) { OP(4) }                 \
    if constexpr (WIDTH >= 16) { OP(8) }                \
    if constexpr (WIDTH >= 32) { OP(16) }               \
    if constexpr (WIDTH >= 64) { OP(32) }
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#define DRM_OP(m) v = fmaxf(v, xor_lane<m>(v));
    DRM_LEVELS(DRM_OP)
#undef DRM_OP
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#define DRM_OP(m) v += xor_lane<m>(v);
    DRM_LEVELS(DRM_OP)
#undef DRM_OP
    return v;
}
// argmax with first-index tie break over a lane group
template <int WIDTH>
__device__ __forceinline__ void group_argmax(float& v, int& i) {
#define DRM_OP(m) { const float ov = xor_lane<m>(v); const int oi = xor_lane_i<m>(i); if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; } }
    DRM_LEVELS(DRM_OP)
#undef DRM_OP
}
#undef DRM_LEVELS

}  // namespace drm
