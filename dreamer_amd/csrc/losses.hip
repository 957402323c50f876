// Loss-side row kernels of the world-model update (lib/world_model.py:463-528): the two KL terms between
// the unimix'ed posterior and prior categoricals, and the per-frame squared reconstruction error.
// Both are streams over (B*T)-row tensors; in eager tensor ops each is a few dozen launches of a few
// microseconds (or, for the image term, six passes over 50 MB tensors).
#include "common.hpp"
#include "../../include/dreamer_amd.h"

namespace drm {

// ---- KL(q || p) per row, q = unimix(softmax(post)), p = unimix(softmax(prior)), K = 32 classes ------------
// kl_l = sum_k q_k (log q_k - log p_k);  row value = sum_l max(kl_l, free)   (lib/world_model.py:463-475)
// The dynamics and the representation loss have the same value and differ in where the gradient goes
// (prior only / posterior only), so the forward writes the row sums twice and the backward takes one
// upstream gradient per use.
// One workgroup per row; a half-wave (32 lanes) per latent, lane = class.
struct Cat32 {
    float soft;    // softmax probability
    float mix;     // (1 - eps) * soft + eps / K
    float logmix;  // log of it, formed as the reference forms it: logaddexp(log_softmax + log(1 - eps), log(eps / K))
};

__device__ __forceinline__ Cat32 unimix32(float logit, float log1m_eps, float log_eps_over_k) {
    const float mx = group_max<32>(logit);
    const float ex = expf(logit - mx);
    const float se = group_sum<32>(ex);
    const float ls = (logit - mx) - logf(se);
    Cat32 c;
    c.soft = ex / se;
    c.logmix = logaddexp_f(ls + log1m_eps, log_eps_over_k);
    c.mix = expf(c.logmix);
    return c;
}

__global__ __launch_bounds__(256) void k_kl_pair_fwd(const float* __restrict__ post, const float* __restrict__ prior, int L,
                                                      float log1m_eps, float log_eps_over_k, float free_nats,
                                                      float* __restrict__ out_a, float* __restrict__ out_b) {
    __shared__ float s_part[8];
    const int row = blockIdx.x, grp = threadIdx.x >> 5, k = threadIdx.x & 31;
    float acc = 0.f;
    for (int l = grp; l < L; l += 8) {   // uniform trip count per group (inactive groups idle at the barrier below)
        const int64_t o = ((int64_t)row * L + l) * 32 + k;
        const Cat32 q = unimix32(post[o], log1m_eps, log_eps_over_k);
        const Cat32 p = unimix32(prior[o], log1m_eps, log_eps_over_k);
        const float kl = group_sum<32>(q.mix * (q.logmix - p.logmix));
        acc += fmaxf(kl, free_nats);
    }
    if (k == 0) s_part[grp] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 8; ++i) t += s_part[i];
        out_a[row] = t;
        out_b[row] = t;
    }
}

__global__ __launch_bounds__(256) void k_kl_pair_bwd(const float* __restrict__ post, const float* __restrict__ prior, int L,
                                                      float log1m_eps, float log_eps_over_k, float one_m_eps, float free_nats,
                                                      const float* __restrict__ g_dyn, const float* __restrict__ g_rep,
                                                      float* __restrict__ d_post, float* __restrict__ d_prior) {
    const int row = blockIdx.x, grp = threadIdx.x >> 5, k = threadIdx.x & 31;
    const float gd = g_dyn ? g_dyn[row] : 0.f, gr = g_rep ? g_rep[row] : 0.f;
    for (int l = grp; l < L; l += 8) {
        const int64_t o = ((int64_t)row * L + l) * 32 + k;
        const Cat32 q = unimix32(post[o], log1m_eps, log_eps_over_k);
        const Cat32 p = unimix32(prior[o], log1m_eps, log_eps_over_k);
        const float diff = q.logmix - p.logmix;
        const float kl = group_sum<32>(q.mix * diff);
        const float on = kl >= free_nats ? 1.f : 0.f;       // clamp_min passes the gradient where kl >= free
        // d kl / d log p_k = -q_k ; d kl / d log q_k = q_k (log q_k - log p_k + 1)
        const float wp = one_m_eps * p.soft / p.mix, wq = one_m_eps * q.soft / q.mix;   // d logmix / d log_softmax
        const float Gp = gd * on * (-q.mix) * wp;
        const float Gq = gr * on * (q.mix * (diff + 1.f)) * wq;
        const float sp = group_sum<32>(Gp), sq = group_sum<32>(Gq);
        d_prior[o] = Gp - p.soft * sp;
        d_post[o] = Gq - q.soft * sq;
    }
}

// ---- 0.5 * mean over a frame of (x_hat - x)^2 (lib/world_model.py:483-487), one workgroup per frame ----------
__global__ __launch_bounds__(256) void k_mse_rows_fwd(const float* __restrict__ x_hat, const float* __restrict__ x, int E,
                                                       float* __restrict__ out) {
    __shared__ float s_part[4];
    const int row = blockIdx.x;
    const float4* a = reinterpret_cast<const float4*>(x_hat + (int64_t)row * E);
    const float4* b = reinterpret_cast<const float4*>(x + (int64_t)row * E);
    float acc = 0.f;
    for (int i = threadIdx.x; i < E / 4; i += 256) {
        const float4 u = a[i], v = b[i];
        const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
        acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = 0.5f * (((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)E);
}

__global__ __launch_bounds__(256) void k_mse_rows_bwd(const float* __restrict__ x_hat, const float* __restrict__ x, int E,
                                                       const float* __restrict__ g, float* __restrict__ d_x_hat) {
    const int row = blockIdx.x;
    const float s = g[row] / (float)E;
    const float4* a = reinterpret_cast<const float4*>(x_hat + (int64_t)row * E);
    const float4* b = reinterpret_cast<const float4*>(x + (int64_t)row * E);
    float4* o = reinterpret_cast<float4*>(d_x_hat + (int64_t)row * E);
    for (int i = threadIdx.x; i < E / 4; i += 256) {
        const float4 u = a[i], v = b[i];
        o[i] = make_float4((u.x - v.x) * s, (u.y - v.y) * s, (u.z - v.z) * s, (u.w - v.w) * s);
    }
}

}  // namespace drm

extern "C" int drm_kl_pair_forward(const float* post, const float* prior, int64_t rows, int32_t L, int32_t K, float unimix_eps,
                                   float free_nats, float* dyn_out, float* rep_out, drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(post && prior && dyn_out && rep_out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(K == 32, DRM_ENOTSUP, "classes per latent must be 32, got %d", K);
    DRM_REQUIRE(rows >= 0 && L >= 1 && unimix_eps > 0.f && unimix_eps < 1.f, DRM_EINVAL, "bad shape or eps");
    if (rows == 0) return DRM_OK;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 60.0 * rows * L * 32, 8.0 * rows * L * 32);
    k_kl_pair_fwd<<<dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream>>>(
        post, prior, L, (float)log1p(-(double)unimix_eps), (float)(log((double)unimix_eps) - log(32.0)), free_nats, dyn_out, rep_out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_kl_pair_backward(const float* post, const float* prior, int64_t rows, int32_t L, int32_t K, float unimix_eps,
                                    float free_nats, const float* g_dyn, const float* g_rep, float* d_post, float* d_prior,
                                    drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(post && prior && d_post && d_prior, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(K == 32, DRM_ENOTSUP, "classes per latent must be 32, got %d", K);
    DRM_REQUIRE(rows >= 0 && L >= 1 && unimix_eps > 0.f && unimix_eps < 1.f, DRM_EINVAL, "bad shape or eps");
    if (rows == 0) return DRM_OK;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 90.0 * rows * L * 32, 16.0 * rows * L * 32);
    k_kl_pair_bwd<<<dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream>>>(
        post, prior, L, (float)log1p(-(double)unimix_eps), (float)(log((double)unimix_eps) - log(32.0)), 1.0f - unimix_eps,
        free_nats, g_dyn, g_rep, d_post, d_prior);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_mse_rows_forward(const float* x_hat, const float* x, int64_t rows, int32_t E, float* out, drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(x_hat && x && out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(rows >= 0 && E >= 4 && E % 4 == 0 && aligned16(x_hat) && aligned16(x), DRM_EINVAL, "bad shape or alignment");
    if (rows == 0) return DRM_OK;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 3.0 * rows * E, 8.0 * rows * E);
    k_mse_rows_fwd<<<dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream>>>(x_hat, x, E, out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_mse_rows_backward(const float* x_hat, const float* x, int64_t rows, int32_t E, const float* g, float* d_x_hat,
                                     drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(x_hat && x && g && d_x_hat, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(rows >= 0 && E >= 4 && E % 4 == 0 && aligned16(x_hat) && aligned16(x) && aligned16(d_x_hat), DRM_EINVAL,
                "bad shape or alignment");
    if (rows == 0) return DRM_OK;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 3.0 * rows * E, 12.0 * rows * E);
    k_mse_rows_bwd<<<dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream>>>(x_hat, x, E, g, d_x_hat);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
