// Adam on one flat parameter buffer, with the global-norm gradient clip folded in
// (lib/trainer.py:170-173: clip_grad_norm_ then Adam(fused) step; same arithmetic as torch's fused Adam).
//
// The parameters of a model, their gradients and both moment estimates each live in one contiguous
// fp32 buffer, so an optimiser step is two HBM-bound launches instead of one multi-tensor launch per
// few dozen tensors plus the norm / scale passes of the clip:
//   k_adam_sumsq : partial sums of g^2 (4 bytes per element read), step counter += 1
//   k_adam_flat  : total norm from the partials, clip coefficient, moments and parameters updated in one
//                  pass (16 bytes read + 12 bytes written per element; clipped gradients are never
//                  written back, nothing reads them before the next zeroing)
// Everything the kernels need at run time (step count, norm) is in device memory, so the pair can be
// recorded into a graph and replayed.
#include "common.hpp"
#include "../../include/dreamer_amd.h"

namespace drm {

constexpr int ADAM_MAX_PARTIALS = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
    v = group_sum<64>(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    const float t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void k_adam_sumsq(const float* __restrict__ g, int64_t n, float* __restrict__ partials,
                                                    float* __restrict__ step) {
    __shared__ float s_red[4];
    const int64_t n4 = n >> 2;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
    const float t = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0) step[0] += 1.0f;
    }
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; int64_t n;
    double lr, beta1, beta2, eps; float max_norm;
    const float* partials; int n_partials;
    const float* step; float* norm_out;
};

// One element.  The hyper-parameters are doubles and the moment / parameter expressions are evaluated in
// double and rounded once, as the mixed double / float expressions of torch's fused kernel are (1 - beta2
// formed in float would already be off by 1e-5 relative).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float coef, double w1, double beta2, double w2,
                                         double step_size, double sqrt_bc2, double eps) {
    g *= coef;
    m = (float)((double)m + w1 * ((double)g - (double)m));
    v = (float)(beta2 * (double)v + w2 * (double)g * (double)g);
    const double denom = (double)sqrtf(v) / sqrt_bc2 + eps;
    p = (float)((double)p - step_size * (double)m / denom);
}

__global__ __launch_bounds__(256) void k_adam_flat(const AdamArgs A) {
    __shared__ float s_red[4];
    float part = 0.f;
    for (int i = threadIdx.x; i < A.n_partials; i += 256) part += A.partials[i];
    const float norm = sqrtf(block_sum_256(part, s_red));
    float coef = 1.0f;
    if (A.max_norm > 0.f) coef = fminf(A.max_norm / (norm + 1e-6f), 1.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.norm_out) A.norm_out[0] = norm;
    const double t = (double)A.step[0];
    const double bc1 = 1.0 - pow(A.beta1, t), bc2 = 1.0 - pow(A.beta2, t);
    const double step_size = A.lr / bc1, sqrt_bc2 = sqrt(bc2);
    const double w1 = 1.0 - A.beta1, w2 = 1.0 - A.beta2;
    const int64_t n4 = A.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 p = reinterpret_cast<float4*>(A.p)[i];
        const float4 g = reinterpret_cast<const float4*>(A.g)[i];
        float4 m = reinterpret_cast<float4*>(A.m)[i];
        float4 v = reinterpret_cast<float4*>(A.v)[i];
        adam_one(p.x, g.x, m.x, v.x, coef, w1, A.beta2, w2, step_size, sqrt_bc2, A.eps);
        adam_one(p.y, g.y, m.y, v.y, coef, w1, A.beta2, w2, step_size, sqrt_bc2, A.eps);
        adam_one(p.z, g.z, m.z, v.z, coef, w1, A.beta2, w2, step_size, sqrt_bc2, A.eps);
        adam_one(p.w, g.w, m.w, v.w, coef, w1, A.beta2, w2, step_size, sqrt_bc2, A.eps);
        reinterpret_cast<float4*>(A.p)[i] = p;
        reinterpret_cast<float4*>(A.m)[i] = m;
        reinterpret_cast<float4*>(A.v)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (A.n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float p = A.p[i], m = A.m[i], v = A.v[i];
        adam_one(p, A.g[i], m, v, coef, w1, A.beta2, w2, step_size, sqrt_bc2, A.eps);
        A.p[i] = p; A.m[i] = m; A.v[i] = v;
    }
}

}  // namespace drm

extern "C" size_t drm_adam_workspace_bytes(void) { return drm::ADAM_MAX_PARTIALS * sizeof(float); }

extern "C" int drm_adam_flat_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                                  double beta1, double beta2, double eps, double max_norm, float* step, float* workspace,
                                  float* norm_out, drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(params && grads && exp_avg && exp_avg_sq && step && workspace, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(n >= 0, DRM_EINVAL, "n=%lld", (long long)n);
    DRM_REQUIRE(aligned16(params) && aligned16(grads) && aligned16(exp_avg) && aligned16(exp_avg_sq), DRM_EINVAL,
                "buffers must be 16-byte aligned");
    DRM_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps > 0.0, DRM_EINVAL, "bad hyper-parameters");
    if (n == 0) return DRM_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = (n + 3) / 4;
    int64_t blocks = (n4 + 256 * 4 - 1) / (256 * 4);     // four 16-byte loads per thread and pass
    const int nb1 = (int)(blocks < 1 ? 1 : (blocks > ADAM_MAX_PARTIALS ? ADAM_MAX_PARTIALS : blocks));
    {
        ProfScope prof(PC_ELEMENTWISE, s, 2.0 * n, 4.0 * n);
        k_adam_sumsq<<<dim3(nb1), dim3(256), 0, s>>>(grads, n, workspace, step);
        DRM_LAUNCH_CHECK();
    }
    const int nb2 = (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
    AdamArgs a{params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)max_norm, workspace, nb1, step, norm_out};
    ProfScope prof(PC_ELEMENTWISE, s, 12.0 * n, 28.0 * n);
    k_adam_flat<<<dim3(nb2), dim3(256), 0, s>>>(a);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
