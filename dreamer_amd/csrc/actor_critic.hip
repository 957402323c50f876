// Row kernels of the critic / actor updates (lib/critic.py:103-161, lib/actor.py:16-33):
//   k_critic_loss_fwd / _bwd : two-hot target of symlog(return), cross-entropy against the critic's bucket
//                              logits, and the KL regulariser towards the slow critic's distribution, per row
//   k_ema_multi              : slow <- decay * slow + (1 - decay) * fast over a list of tensors in one launch
//   k_percentile_scale       : 5 % / 95 % quantiles of the returns (sorted in LDS, linear interpolation as
//                              torch.quantile), their EMA, and S = max(p95 - p05, limit)
// Each replaces a chain of a dozen or more small eager launches per update.
#include "common.hpp"
#include "../../include/dreamer_amd.h"

namespace drm {

// ---- critic loss rows ---------------------------------------------------------------------------------
// One wave per row, up to 256 buckets (four per lane).
//   x = clamp(symlog(ret), lo, hi); pos = (x - lo) / delta; a = floor(pos); b = min(a + 1, NB - 1); w = clamp(pos - a, 0, 1)
//   y[a] = 1 - w; y[b] += w                                         (lib/critic.py:103-129)
//   logp = log_softmax(logits); s = softmax(slow_logits)
//   row = -sum_k y_k logp_k + reg * sum_k (s_k log s_k - s_k logp_k)  (lib/critic.py:140-161; xlogy(0, 0) = 0)
struct TwoHot { int a, b; float wa, wb; };

__device__ __forceinline__ TwoHot twohot_of(float ret, float lo, float hi, float delta, int NB) {
    const float sl = copysignf(log1pf(fabsf(ret)), ret);          // symlog; sign(0) * log1p(0) = 0 either way
    const float x = fminf(fmaxf(sl, lo), hi);
    const float pos = (x - lo) / delta;
    const float fl = floorf(pos);
    TwoHot t;
    t.a = (int)fl;
    t.b = min(t.a + 1, NB - 1);
    t.wb = fminf(fmaxf(pos - fl, 0.f), 1.f);
    t.wa = 1.0f - t.wb;
    return t;
}

__device__ __forceinline__ float target_at(const TwoHot& t, int k) {
    return (k == t.a ? t.wa : 0.f) + (k == t.b ? t.wb : 0.f);
}

__global__ __launch_bounds__(256) void k_critic_loss_fwd(const float* __restrict__ logits, const float* __restrict__ slow,
                                                          const float* __restrict__ ret, int64_t rows, int NB, float lo, float hi,
                                                          float delta, float reg, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float NEG = -3.0e38f;
    float v[4], sv[4];
    float mx = NEG, smx = NEG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        v[j] = k < NB ? logits[row * NB + k] : NEG;
        sv[j] = (k < NB && slow) ? slow[row * NB + k] : NEG;
        mx = fmaxf(mx, v[j]); smx = fmaxf(smx, sv[j]);
    }
    mx = group_max<64>(mx); smx = group_max<64>(smx);
    float se = 0.f, sse = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k < NB) { se += expf(v[j] - mx); if (slow) sse += expf(sv[j] - smx); }
    }
    const float lse = logf(group_sum<64>(se));
    const float slse = slow ? logf(group_sum<64>(sse)) : 0.f;
    const TwoHot t = twohot_of(ret[row], lo, hi, delta, NB);
    float ce = 0.f, kl = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k >= NB) continue;
        const float logp = (v[j] - mx) - lse;
        ce += target_at(t, k) * logp;
        if (slow) {
            const float ls = (sv[j] - smx) - slse;
            const float s = expf(ls);
            kl += (s > 0.f ? s * ls : 0.f) - s * logp;
        }
    }
    ce = group_sum<64>(ce); kl = group_sum<64>(kl);
    if (lane == 0) out[row] = -ce + reg * kl;
}

// d row / d logits_k = (softmax_k - y_k) + reg * (softmax_k - s_k)      (sum y = sum s = 1)
__global__ __launch_bounds__(256) void k_critic_loss_bwd(const float* __restrict__ logits, const float* __restrict__ slow,
                                                          const float* __restrict__ ret, const float* __restrict__ g, int64_t rows,
                                                          int NB, float lo, float hi, float delta, float reg,
                                                          float* __restrict__ d_logits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float NEG = -3.0e38f;
    float v[4], sv[4];
    float mx = NEG, smx = NEG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        v[j] = k < NB ? logits[row * NB + k] : NEG;
        sv[j] = (k < NB && slow) ? slow[row * NB + k] : NEG;
        mx = fmaxf(mx, v[j]); smx = fmaxf(smx, sv[j]);
    }
    mx = group_max<64>(mx); smx = group_max<64>(smx);
    float e[4], es[4], se = 0.f, sse = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        e[j] = k < NB ? expf(v[j] - mx) : 0.f;
        es[j] = (k < NB && slow) ? expf(sv[j] - smx) : 0.f;
        se += e[j]; sse += es[j];
    }
    se = group_sum<64>(se); sse = group_sum<64>(sse);
    const TwoHot t = twohot_of(ret[row], lo, hi, delta, NB);
    const float gr = g[row];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k >= NB) continue;
        const float p = e[j] / se;
        float d = p - target_at(t, k);
        if (slow) d += reg * (p - es[j] / sse);
        d_logits[row * NB + k] = gr * d;
    }
}

// ---- multi-tensor EMA ------------------------------------------------------------------------------------
// dst <- dst * decay; dst <- dst + src * one_minus (two separately rounded steps, as the in-place
// tensor ops of lib/critic.py:131-138 round them)
constexpr int EMA_MAX_TENSORS = 32;
struct EmaJobs { const float* src[EMA_MAX_TENSORS]; float* dst[EMA_MAX_TENSORS]; int64_t n[EMA_MAX_TENSORS];
                 int block_begin[EMA_MAX_TENSORS + 1]; int count; };

__global__ __launch_bounds__(256) void k_ema_multi(const EmaJobs J, float decay, float one_minus) {
    int t = 0;
    while (t + 1 < J.count && (int)blockIdx.x >= J.block_begin[t + 1]) ++t;
    const int64_t base = ((int64_t)blockIdx.x - J.block_begin[t]) * 1024;
    const float* __restrict__ src = J.src[t];
    float* __restrict__ dst = J.dst[t];
    const int64_t n = J.n[t];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + threadIdx.x + 256 * j;
        if (i < n) {
            const float s = dst[i] * decay;
            dst[i] = s + src[i] * one_minus;
        }
    }
}

// ---- return percentiles ----------------------------------------------------------------------------------
// One workgroup: bitonic sort of up to 16384 values in LDS, then torch.quantile's linear interpolation
//   rank = q * (n - 1) (fp32); lo = floor(rank); w = rank - lo; value = lerp(x[lo], x[ceil(rank)], w)
// with torch's lerp (w < 0.5: a + w (b - a), else b - (b - a)(1 - w)); hi = max(q95, q05 + 1e-8);
// training: p <- p * decay + (1 - decay) * q (lib/actor.py:16-33); S = max(p95 - p05, limit).
constexpr int PCT_MAX = 16384;

__device__ __forceinline__ float lerp_torch(float a, float b, float w) {
    const float d = b - a;
    return w < 0.5f ? a + w * d : b - d * (1.0f - w);
}

__global__ __launch_bounds__(1024) void k_percentile_scale(const float* __restrict__ x, int n, int npow2, float q_lo, float q_hi,
                                                            float decay, float one_minus, float limit, int training,
                                                            float* __restrict__ p05, float* __restrict__ p95,
                                                            float* __restrict__ scale_out) {
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < npow2; i += 1024) s[i] = i < n ? x[i] : __builtin_inff();
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = s[i], b = s[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        const float last = (float)(n - 1);
        float q[2];
        const float qs[2] = {q_lo, q_hi};
        for (int i = 0; i < 2; ++i) {
            const float rank = qs[i] * last;
            const float below = floorf(rank);
            const int ib = (int)below, ia = (int)ceilf(rank);
            q[i] = lerp_torch(s[ib], s[ia < n ? ia : n - 1], rank - below);
        }
        q[1] = fmaxf(q[1], q[0] + 1e-8f);
        float a = p05[0], b = p95[0];
        if (training) {
            a = a * decay + one_minus * q[0];
            b = b * decay + one_minus * q[1];
            p05[0] = a; p95[0] = b;
        }
        scale_out[0] = fmaxf(b - a, limit);
    }
}

}  // namespace drm

using namespace drm;

static int check_critic_loss(const float* logits, const float* ret, int64_t rows, int32_t NB, double lo, double hi) {
    DRM_REQUIRE(logits && ret, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(rows >= 0 && NB >= 2 && NB <= 256 && hi > lo, DRM_EINVAL, "rows=%lld buckets=%d", (long long)rows, NB);
    return DRM_OK;
}

extern "C" int drm_critic_loss_forward(const float* logits, const float* slow_logits, const float* returns, int64_t rows,
                                       int32_t NB, double bucket_min, double bucket_max, double reg, float* row_loss,
                                       drm_stream_t stream) {
    if (int rc = check_critic_loss(logits, returns, rows, NB, bucket_min, bucket_max)) return rc;
    DRM_REQUIRE(row_loss, DRM_EINVAL, "null output");
    if (rows == 0) return DRM_OK;
    const float delta = (float)((bucket_max - bucket_min) / (double)(NB - 1));
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 30.0 * rows * NB, 8.0 * rows * NB);
    k_critic_loss_fwd<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        logits, reg > 0.0 ? slow_logits : nullptr, returns, rows, NB, (float)bucket_min, (float)bucket_max, delta, (float)reg, row_loss);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_critic_loss_backward(const float* logits, const float* slow_logits, const float* returns, const float* g_rows,
                                        int64_t rows, int32_t NB, double bucket_min, double bucket_max, double reg,
                                        float* d_logits, drm_stream_t stream) {
    if (int rc = check_critic_loss(logits, returns, rows, NB, bucket_min, bucket_max)) return rc;
    DRM_REQUIRE(g_rows && d_logits, DRM_EINVAL, "null pointer");
    if (rows == 0) return DRM_OK;
    const float delta = (float)((bucket_max - bucket_min) / (double)(NB - 1));
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 30.0 * rows * NB, 12.0 * rows * NB);
    k_critic_loss_bwd<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        logits, reg > 0.0 ? slow_logits : nullptr, returns, g_rows, rows, NB, (float)bucket_min, (float)bucket_max, delta, (float)reg,
        d_logits);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_ema_multi(int32_t count, const float* const* src, float* const* dst, const int64_t* numel, double decay,
                             drm_stream_t stream) {
    DRM_REQUIRE(count >= 0 && (count == 0 || (src && dst && numel)), DRM_EINVAL, "bad tensor list");
    hipStream_t s = (hipStream_t)stream;
    for (int first = 0; first < count; first += EMA_MAX_TENSORS) {
        EmaJobs J{};
        int blocks = 0;
        double total = 0;
        J.count = count - first < EMA_MAX_TENSORS ? count - first : EMA_MAX_TENSORS;
        for (int i = 0; i < J.count; ++i) {
            DRM_REQUIRE(src[first + i] && dst[first + i] && numel[first + i] >= 0, DRM_EINVAL, "bad tensor %d", first + i);
            J.src[i] = src[first + i]; J.dst[i] = dst[first + i]; J.n[i] = numel[first + i];
            J.block_begin[i] = blocks;
            blocks += (int)((numel[first + i] + 1023) / 1024);
            total += (double)numel[first + i];
        }
        J.block_begin[J.count] = blocks;
        if (blocks == 0) continue;
        ProfScope prof(PC_ELEMENTWISE, s, 3.0 * total, 12.0 * total);
        k_ema_multi<<<dim3(blocks), dim3(256), 0, s>>>(J, (float)decay, (float)(1.0 - decay));
        DRM_LAUNCH_CHECK();
    }
    return DRM_OK;
}

extern "C" int drm_percentile_scale(const float* x, int32_t n, double q_lo, double q_hi, double decay, double limit,
                                    int32_t training, float* p_lo, float* p_hi, float* scale_out, drm_stream_t stream) {
    DRM_REQUIRE(x && p_lo && p_hi && scale_out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(n >= 1 && n <= PCT_MAX, DRM_ENOTSUP, "n=%d (1..%d values)", n, PCT_MAX);
    DRM_REQUIRE(q_lo >= 0.0 && q_lo <= q_hi && q_hi <= 1.0, DRM_EINVAL, "quantiles %g, %g", q_lo, q_hi);
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 20.0 * npow2, 4.0 * n);
    k_percentile_scale<<<dim3(1), dim3(1024), npow2 * sizeof(float), (hipStream_t)stream>>>(
        x, n, npow2, (float)q_lo, (float)q_hi, (float)decay, (float)(1.0 - decay), (float)limit, training, p_lo, p_hi, scale_out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
