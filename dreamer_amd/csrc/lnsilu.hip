// LayerNorm over a whole sample (normalized_shape = [C, H, W], elementwise affine) fused with SiLU,
// forward and backward: the stage nonlinearity of the encoder / decoder conv towers
// (lib/world_model.py:94-108, :290-302).  HBM-bound; E = C*H*W elements per sample, N samples.
//
//   forward : y = SiLU((x - mean) * rstd * gamma + beta); mean, rstd saved per sample
//   backward: u = xhat*gamma + beta, s = sigmoid(u), du = dy * s * (1 + u (1 - s)), g = du * gamma
//             dx = rstd * (g - mean_E(g) - xhat * mean_E(g xhat))
//             dgamma[e] = sum_n du xhat,  dbeta[e] = sum_n du
//
// One 1024-thread workgroup per sample keeps the sample in registers (<= 8 float4 per thread) between
// the statistics and the output pass, so x (and dy) are read once.  The parameter gradients are a
// reduction over samples: a second kernel walks the samples with one thread per four elements and
// writes one partial row per sample slice (no atomics; the caller adds the slices).
//
// Optional channel bias: the producing convolution's bias b[c] (c = e / (H*W)) is added on the way in
// (x + b is what gets normalised), so the convolution runs without its own bias pass over the output;
// its gradient, the sum of dx over samples and positions, comes out of the parameter kernel as a third
// partial row per slice (dx is re-formed there from two per-sample sums the dx kernel leaves behind).
#include "common.hpp"
#include <cstdlib>

namespace drm {

constexpr int LNS_THREADS = 1024;

__device__ __forceinline__ float block_sum_1024(float v, float* s16) {
    v = group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s16[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s16[i];
    return t;
}

// Channels-last samples: the data of a sample is laid out (position, channel) while gamma / beta (and the
// parameter-gradient rows) keep the module's (channel, position) order.  clC = number of channels (0: the
// data is in parameter order), hw = positions per channel.  Four consecutive data elements are four
// consecutive channels of one position.
__device__ __forceinline__ float4 lns_param4(const float* __restrict__ p, int e, int clC, int hw) {
    if (clC == 0) return *reinterpret_cast<const float4*>(p + e);
    const int c = e % clC, pos = e / clC;
    const float* q = p + (int64_t)c * hw + pos;
    return make_float4(q[0], q[hw], q[2 * hw], q[3 * hw]);
}
__device__ __forceinline__ void lns_store_param4(float* __restrict__ p, int e, int clC, int hw, float4 v) {
    if (clC == 0) { *reinterpret_cast<float4*>(p + e) = v; return; }
    const int c = e % clC, pos = e / clC;
    float* q = p + (int64_t)c * hw + pos;
    q[0] = v.x; q[hw] = v.y; q[2 * hw] = v.z; q[3 * hw] = v.w;
}
__device__ __forceinline__ float4 lns_cbias4(const float* __restrict__ cbias, int e, int clC, int hw) {
    if (cbias == nullptr) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (clC != 0) return *reinterpret_cast<const float4*>(cbias + e % clC);
    const float b = cbias[e / hw];
    return make_float4(b, b, b, b);
}

template <int NV>
__global__ __launch_bounds__(LNS_THREADS) void k_lnsilu_fwd(const float* __restrict__ X, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ Y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int E, float eps, const float* __restrict__ cbias, int hw, int clC) {
    __shared__ float s16[16];
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x;
    const float* x = X + n * E;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * LNS_THREADS + tid) * 4;
        v[i] = e < E ? *reinterpret_cast<const float4*>(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (cbias != nullptr && e < E) { const float4 b = lns_cbias4(cbias, e, clC, hw); v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w; }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = block_sum_1024(s, s16) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * LNS_THREADS + tid) * 4;
        if (e < E) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(block_sum_1024(q, s16) / (float)E + eps);
    if (tid == 0) { mean_out[n] = mean; rstd_out[n] = rstd; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * LNS_THREADS + tid) * 4;
        if (e < E) {
            const float4 g = lns_param4(gamma, e, clC, hw);
            const float4 b = lns_param4(beta, e, clC, hw);
            float4 o;
            o.x = silu_f((v[i].x - mean) * rstd * g.x + b.x);
            o.y = silu_f((v[i].y - mean) * rstd * g.y + b.y);
            o.z = silu_f((v[i].z - mean) * rstd * g.z + b.z);
            o.w = silu_f((v[i].w - mean) * rstd * g.w + b.w);
            *reinterpret_cast<float4*>(Y + n * E + e) = o;
        }
    }
}

// ---- many long samples: walk form --------------------------------------------------------------------
// The kernels above hold a sample in the registers of one 1024-thread workgroup and fetch gamma and beta per sample:
// x, gamma, beta and y are four tensor-sized streams through the vector caches per forward launch (seven per backward
// pass), and the caches' throughput, not HBM, bounds them (a plain copy of x takes 37 us where the forward took 97).
// With many samples the passes below keep a thread on its four elements instead and walk the samples: gamma, beta and the
// channel bias stay in registers, a launch moves only the data tensors.
//   forward : k_lns_stats (a 256-thread workgroup streams a sample: mean, rstd)  ->  k_lns_apply_walk
//   backward: k_lns_bwd_walk1 (d gamma / d beta partial rows; per sample and 256 elements the sums of g and g xhat)
//             ->  k_lns_bwd_merge (those sums per sample, fixed order)  ->  k_lns_bwd_walk2 (dx, bias-gradient row)
// Statistics as (count, mean, M2) triples merged pairwise (Chan et al.): each batch is reduced exactly (mean first, then
// squared deviations from it), batches and threads are merged with  M2 = M2a + M2b + d^2 na nb / n,  d = mean_b - mean_a.
// sigmoid from the hardware's exp2 and reciprocal (each within 1 ulp): 5 instructions where expf and the IEEE division
// take about 25 -- at four tensor-sized streams per launch the walk kernels are otherwise bound by the vector ALUs
__device__ __forceinline__ float lns_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896341f));
}
__device__ __forceinline__ float lns_silu(float u) { return u * lns_sigmoid(u); }
__device__ __forceinline__ float lns_dsilu_times(float dy, float u) {
    const float s = lns_sigmoid(u);
    return dy * s * (1.0f + u * (1.0f - s));
}
struct LnsMom { float n, mean, m2; };
__device__ __forceinline__ LnsMom lns_merge(const LnsMom a, const LnsMom b) {
    const float n = a.n + b.n;
    const float f = n > 0.f ? b.n / n : 0.f;
    const float d = b.mean - a.mean;
    return LnsMom{n, fmaf(d, f, a.mean), a.m2 + b.m2 + d * d * a.n * f};
}
template <int OFF>
__device__ __forceinline__ LnsMom lns_merge_lanes(const LnsMom a) {
    return lns_merge(a, LnsMom{xor_lane<OFF>(a.n), xor_lane<OFF>(a.mean), xor_lane<OFF>(a.m2)});
}
// the whole workgroup's triple, the same bits in every thread (fixed merge order over the four waves)
__device__ __forceinline__ LnsMom lns_block_merge_256(LnsMom a, float (*s)[3]) {
    a = lns_merge_lanes<1>(a); a = lns_merge_lanes<2>(a); a = lns_merge_lanes<4>(a);
    a = lns_merge_lanes<8>(a); a = lns_merge_lanes<16>(a); a = lns_merge_lanes<32>(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6][0] = a.n; s[threadIdx.x >> 6][1] = a.mean; s[threadIdx.x >> 6][2] = a.m2; }
    __syncthreads();
    LnsMom t{s[0][0], s[0][1], s[0][2]};
#pragma unroll
    for (int w = 1; w < 4; ++w) t = lns_merge(t, LnsMom{s[w][0], s[w][1], s[w][2]});
    return t;
}

template <int U>
__global__ __launch_bounds__(256) void k_lns_stats(const float* __restrict__ X, float* __restrict__ mean_out,
                                                   float* __restrict__ rstd_out, int E, float eps,
                                                   const float* __restrict__ cbias, int hw, int clC) {
    __shared__ float s_m[4][3];
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x;
    const float* x = X + n * E;
    const int E4 = E >> 2;
    LnsMom acc{0.f, 0.f, 0.f};
    for (int base = 0; base < E4; base += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i4 = base + u * 256 + tid;
            v[u] = *reinterpret_cast<const float4*>(x + 4 * (i4 < E4 ? i4 : 0));
        }
        float cnt = 0.f, sm = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i4 = base + u * 256 + tid;
            if (cbias != nullptr) {
                const float4 b = lns_cbias4(cbias, 4 * (i4 < E4 ? i4 : 0), clC, hw);
                v[u].x += b.x; v[u].y += b.y; v[u].z += b.z; v[u].w += b.w;
            }
            if (i4 < E4) { sm += (v[u].x + v[u].y) + (v[u].z + v[u].w); cnt += 4.f; }
        }
        const float bm = cnt > 0.f ? sm / cnt : 0.f;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i4 = base + u * 256 + tid;
            const float a = v[u].x - bm, b = v[u].y - bm, c = v[u].z - bm, d = v[u].w - bm;
            if (i4 < E4) q += (a * a + b * b) + (c * c + d * d);
        }
        acc = lns_merge(acc, LnsMom{cnt, bm, q});
    }
    const LnsMom tot = lns_block_merge_256(acc, s_m);
    if (tid == 0) { mean_out[n] = tot.mean; rstd_out[n] = 1.0f / sqrtf(tot.m2 / (float)E + eps); }
}

// grid (ceil(E / 1024), slices): thread = four elements, walks the samples of its slice four at a time
__global__ __launch_bounds__(256) void k_lns_apply_walk(const float* __restrict__ X, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mean_in,
                                                        const float* __restrict__ rstd_in, float* __restrict__ Y, int N, int E,
                                                        int per_slice, const float* __restrict__ cbias, int hw, int clC) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    const float4 cb = lns_cbias4(cbias, e, clC, hw);
    const float4 ga = lns_param4(gamma, e, clC, hw);
    const float4 be = lns_param4(beta, e, clC, hw);
    const int n0 = blockIdx.y * per_slice, n1 = min(n0 + per_slice, N);
    for (int n = n0; n < n1; n += 4) {
        float4 xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const float4*>(X + (int64_t)min(n + j, n1 - 1) * E + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= n1) break;
            const float mean = mean_in[n + j], rstd = rstd_in[n + j];
            float4 o;
            o.x = lns_silu((xv[j].x + cb.x - mean) * rstd * ga.x + be.x);
            o.y = lns_silu((xv[j].y + cb.y - mean) * rstd * ga.y + be.y);
            o.z = lns_silu((xv[j].z + cb.z - mean) * rstd * ga.z + be.z);
            o.w = lns_silu((xv[j].w + cb.w - mean) * rstd * ga.w + be.w);
            *reinterpret_cast<float4*>(Y + (int64_t)(n + j) * E + e) = o;
        }
    }
}

__device__ __forceinline__ float dsilu_times(float dy, float u) {
    const float s = 1.0f / (1.0f + expf(-u));
    return dy * s * (1.0f + u * (1.0f - s));
}

template <int NV>
__global__ __launch_bounds__(LNS_THREADS) void k_lnsilu_bwd_dx(const float* __restrict__ X, const float* __restrict__ dY,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                               float* __restrict__ dX, int E, const float* __restrict__ cbias, int hw,
                                                               float* __restrict__ sums_out, int clC) {
    __shared__ float s16[16];
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x;
    const float mean = mean_in[n], rstd = rstd_in[n];
    float4 xh[NV], g[NV];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * LNS_THREADS + tid) * 4;
        if (e < E) {
            float4 xv = *reinterpret_cast<const float4*>(X + n * E + e);
            const float4 dy = *reinterpret_cast<const float4*>(dY + n * E + e);
            const float4 ga = lns_param4(gamma, e, clC, hw);
            const float4 be = lns_param4(beta, e, clC, hw);
            if (cbias != nullptr) { const float4 b = lns_cbias4(cbias, e, clC, hw); xv.x += b.x; xv.y += b.y; xv.z += b.z; xv.w += b.w; }
            xh[i].x = (xv.x - mean) * rstd; xh[i].y = (xv.y - mean) * rstd; xh[i].z = (xv.z - mean) * rstd; xh[i].w = (xv.w - mean) * rstd;
            g[i].x = dsilu_times(dy.x, xh[i].x * ga.x + be.x) * ga.x;
            g[i].y = dsilu_times(dy.y, xh[i].y * ga.y + be.y) * ga.y;
            g[i].z = dsilu_times(dy.z, xh[i].z * ga.z + be.z) * ga.z;
            g[i].w = dsilu_times(dy.w, xh[i].w * ga.w + be.w) * ga.w;
            sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            sgx += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        } else {
            xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = xh[i];
        }
    }
    const float mg = block_sum_1024(sg, s16) / (float)E;
    const float mgx = block_sum_1024(sgx, s16) / (float)E;
    if (sums_out != nullptr && tid == 0) { sums_out[2 * n] = mg; sums_out[2 * n + 1] = mgx; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * LNS_THREADS + tid) * 4;
        if (e < E) {
            float4 o;
            o.x = rstd * (g[i].x - mg - xh[i].x * mgx);
            o.y = rstd * (g[i].y - mg - xh[i].y * mgx);
            o.z = rstd * (g[i].z - mg - xh[i].z * mgx);
            o.w = rstd * (g[i].w - mg - xh[i].w * mgx);
            *reinterpret_cast<float4*>(dX + n * E + e) = o;
        }
    }
}

// ---- the same two kernels for short samples (rows of an MLP, E <= 2048): one wave per sample, four
// samples per workgroup, no LDS and no barrier ------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void k_lnsilu_row_fwd(const float* __restrict__ X, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ Y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int64_t N, int E, float eps, const float* __restrict__ cbias, int hw) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * 64 + lane) * 4;
        v[i] = e < E ? *reinterpret_cast<const float4*>(X + n * E + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (cbias != nullptr && e < E) { const float b = cbias[e / hw]; v[i].x += b; v[i].y += b; v[i].z += b; v[i].w += b; }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = group_sum<64>(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((i * 64 + lane) * 4 < E) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(group_sum<64>(q) / (float)E + eps);
    if (lane == 0) { mean_out[n] = mean; rstd_out[n] = rstd; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * 64 + lane) * 4;
        if (e < E) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + e);
            const float4 b = *reinterpret_cast<const float4*>(beta + e);
            float4 o;
            o.x = silu_f((v[i].x - mean) * rstd * g.x + b.x);
            o.y = silu_f((v[i].y - mean) * rstd * g.y + b.y);
            o.z = silu_f((v[i].z - mean) * rstd * g.z + b.z);
            o.w = silu_f((v[i].w - mean) * rstd * g.w + b.w);
            *reinterpret_cast<float4*>(Y + n * E + e) = o;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void k_lnsilu_row_bwd_dx(const float* __restrict__ X, const float* __restrict__ dY,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           float* __restrict__ dX, int64_t N, int E, const float* __restrict__ cbias,
                                                           int hw, float* __restrict__ sums_out) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float mean = mean_in[n], rstd = rstd_in[n];
    float4 xh[NV], g[NV];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * 64 + lane) * 4;
        if (e < E) {
            float4 xv = *reinterpret_cast<const float4*>(X + n * E + e);
            const float4 dy = *reinterpret_cast<const float4*>(dY + n * E + e);
            const float4 ga = *reinterpret_cast<const float4*>(gamma + e);
            const float4 be = *reinterpret_cast<const float4*>(beta + e);
            if (cbias != nullptr) { const float b = cbias[e / hw]; xv.x += b; xv.y += b; xv.z += b; xv.w += b; }
            xh[i].x = (xv.x - mean) * rstd; xh[i].y = (xv.y - mean) * rstd; xh[i].z = (xv.z - mean) * rstd; xh[i].w = (xv.w - mean) * rstd;
            g[i].x = dsilu_times(dy.x, xh[i].x * ga.x + be.x) * ga.x;
            g[i].y = dsilu_times(dy.y, xh[i].y * ga.y + be.y) * ga.y;
            g[i].z = dsilu_times(dy.z, xh[i].z * ga.z + be.z) * ga.z;
            g[i].w = dsilu_times(dy.w, xh[i].w * ga.w + be.w) * ga.w;
            sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            sgx += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        } else {
            xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = xh[i];
        }
    }
    const float mg = group_sum<64>(sg) / (float)E;
    const float mgx = group_sum<64>(sgx) / (float)E;
    if (sums_out != nullptr && lane == 0) { sums_out[2 * n] = mg; sums_out[2 * n + 1] = mgx; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * 64 + lane) * 4;
        if (e < E) {
            float4 o;
            o.x = rstd * (g[i].x - mg - xh[i].x * mgx);
            o.y = rstd * (g[i].y - mg - xh[i].y * mgx);
            o.z = rstd * (g[i].z - mg - xh[i].z * mgx);
            o.w = rstd * (g[i].w - mg - xh[i].w * mgx);
            *reinterpret_cast<float4*>(dX + n * E + e) = o;
        }
    }
}

// grid (ceil(E / 1024), slices): thread = four elements, walks the samples of its slice
__global__ __launch_bounds__(256) void k_lnsilu_bwd_params(const float* __restrict__ X, const float* __restrict__ dY,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           float* __restrict__ part, int N, int E, int per_slice,
                                                           const float* __restrict__ cbias, int hw, const float* __restrict__ sums,
                                                           int clC) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    const float4 cb = lns_cbias4(cbias, e, clC, hw);
    float4 dxs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n0 = blockIdx.y * per_slice, n1 = min(n0 + per_slice, N);
    const float4 ga = lns_param4(gamma, e, clC, hw);
    const float4 be = lns_param4(beta, e, clC, hw);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    for (int n = n0; n < n1; ++n) {
        const float mean = mean_in[n], rstd = rstd_in[n];
        float4 xv = *reinterpret_cast<const float4*>(X + (int64_t)n * E + e);
        const float4 dy = *reinterpret_cast<const float4*>(dY + (int64_t)n * E + e);
        xv.x += cb.x; xv.y += cb.y; xv.z += cb.z; xv.w += cb.w;
        const float x0 = (xv.x - mean) * rstd, x1 = (xv.y - mean) * rstd, x2 = (xv.z - mean) * rstd, x3 = (xv.w - mean) * rstd;
        const float d0 = dsilu_times(dy.x, x0 * ga.x + be.x), d1 = dsilu_times(dy.y, x1 * ga.y + be.y);
        const float d2 = dsilu_times(dy.z, x2 * ga.z + be.z), d3 = dsilu_times(dy.w, x3 * ga.w + be.w);
        dg.x += d0 * x0; dg.y += d1 * x1; dg.z += d2 * x2; dg.w += d3 * x3;
        db.x += d0; db.y += d1; db.z += d2; db.w += d3;
        if (sums != nullptr) {   // dx of this sample, as the dx kernel formed it
            const float mg = sums[2 * n], mgx = sums[2 * n + 1];
            dxs.x += rstd * (d0 * ga.x - mg - x0 * mgx); dxs.y += rstd * (d1 * ga.y - mg - x1 * mgx);
            dxs.z += rstd * (d2 * ga.z - mg - x2 * mgx); dxs.w += rstd * (d3 * ga.w - mg - x3 * mgx);
        }
    }
    const int rows = sums != nullptr ? 3 : 2;
    float* row = part + (int64_t)blockIdx.y * rows * E;
    lns_store_param4(row, e, clC, hw, dg);          // parameter order in every row
    lns_store_param4(row + E, e, clC, hw, db);
    if (sums != nullptr) lns_store_param4(row + 2 * E, e, clC, hw, dxs);
}

// Walk form of the backward pass (whole waves only: E a multiple of 256).  wpart: (N, E / 256) x (sum of g, sum of g xhat)
// over the 256 elements of a wave, g = dy SiLU'(u) gamma.
__global__ __launch_bounds__(256) void k_lns_bwd_walk1(const float* __restrict__ X, const float* __restrict__ dY,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                       float* __restrict__ part, int rows, float* __restrict__ wpart, int N, int E,
                                                       int per_slice, const float* __restrict__ cbias, int hw, int clC) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int lane = threadIdx.x & 63;
    const int wchunk = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = E >> 8;
    const float4 cb = lns_cbias4(cbias, e, clC, hw);
    const float4 ga = lns_param4(gamma, e, clC, hw);
    const float4 be = lns_param4(beta, e, clC, hw);
    const int n0 = blockIdx.y * per_slice, n1 = min(n0 + per_slice, N);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    for (int n = n0; n < n1; n += 4) {
        float4 xv[4], dy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t off = (int64_t)min(n + j, n1 - 1) * E + e;
            xv[j] = *reinterpret_cast<const float4*>(X + off);
            dy[j] = *reinterpret_cast<const float4*>(dY + off);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= n1) break;                 // (uniform)
            const float mean = mean_in[n + j], rstd = rstd_in[n + j];
            const float x0 = (xv[j].x + cb.x - mean) * rstd, x1 = (xv[j].y + cb.y - mean) * rstd;
            const float x2 = (xv[j].z + cb.z - mean) * rstd, x3 = (xv[j].w + cb.w - mean) * rstd;
            const float d0 = lns_dsilu_times(dy[j].x, x0 * ga.x + be.x), d1 = lns_dsilu_times(dy[j].y, x1 * ga.y + be.y);
            const float d2 = lns_dsilu_times(dy[j].z, x2 * ga.z + be.z), d3 = lns_dsilu_times(dy[j].w, x3 * ga.w + be.w);
            dg.x += d0 * x0; dg.y += d1 * x1; dg.z += d2 * x2; dg.w += d3 * x3;
            db.x += d0; db.y += d1; db.z += d2; db.w += d3;
            const float g0 = d0 * ga.x, g1 = d1 * ga.y, g2 = d2 * ga.z, g3 = d3 * ga.w;
            const float sg = group_sum<64>((g0 + g1) + (g2 + g3));
            const float sgx = group_sum<64>((g0 * x0 + g1 * x1) + (g2 * x2 + g3 * x3));
            if (lane == 0) *reinterpret_cast<float2*>(wpart + ((int64_t)(n + j) * nw + wchunk) * 2) = make_float2(sg, sgx);
        }
    }
    float* row = part + (int64_t)blockIdx.y * rows * E;
    lns_store_param4(row, e, clC, hw, dg);          // parameter order in every row
    lns_store_param4(row + E, e, clC, hw, db);
}

// sums[n] = (mean over the sample of g, of g xhat): one wave per sample, every lane a fixed stride of the wave sums
__global__ __launch_bounds__(256) void k_lns_bwd_merge(const float* __restrict__ wpart, int nw, int N, int E, float* __restrict__ sums) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float a = 0.f, b = 0.f;
    for (int w = lane; w < nw; w += 64) {
        const float2 v = *reinterpret_cast<const float2*>(wpart + ((int64_t)n * nw + w) * 2);
        a += v.x; b += v.y;
    }
    a = group_sum<64>(a);
    b = group_sum<64>(b);
    if (lane == 0) { sums[2 * n] = a / (float)E; sums[2 * n + 1] = b / (float)E; }
}

__global__ __launch_bounds__(256) void k_lns_bwd_walk2(const float* __restrict__ X, const float* __restrict__ dY,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                       const float* __restrict__ sums, float* __restrict__ dX,
                                                       float* __restrict__ part, int rows, int N, int E, int per_slice,
                                                       const float* __restrict__ cbias, int hw, int clC) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    const float4 cb = lns_cbias4(cbias, e, clC, hw);
    const float4 ga = lns_param4(gamma, e, clC, hw);
    const float4 be = lns_param4(beta, e, clC, hw);
    const int n0 = blockIdx.y * per_slice, n1 = min(n0 + per_slice, N);
    float4 dxs = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = n0; n < n1; n += 4) {
        float4 xv[4], dy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t off = (int64_t)min(n + j, n1 - 1) * E + e;
            xv[j] = *reinterpret_cast<const float4*>(X + off);
            dy[j] = *reinterpret_cast<const float4*>(dY + off);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= n1) break;
            const float mean = mean_in[n + j], rstd = rstd_in[n + j];
            const float mg = sums[2 * (n + j)], mgx = sums[2 * (n + j) + 1];
            const float x0 = (xv[j].x + cb.x - mean) * rstd, x1 = (xv[j].y + cb.y - mean) * rstd;
            const float x2 = (xv[j].z + cb.z - mean) * rstd, x3 = (xv[j].w + cb.w - mean) * rstd;
            const float d0 = lns_dsilu_times(dy[j].x, x0 * ga.x + be.x), d1 = lns_dsilu_times(dy[j].y, x1 * ga.y + be.y);
            const float d2 = lns_dsilu_times(dy[j].z, x2 * ga.z + be.z), d3 = lns_dsilu_times(dy[j].w, x3 * ga.w + be.w);
            float4 o;
            o.x = rstd * (d0 * ga.x - mg - x0 * mgx); o.y = rstd * (d1 * ga.y - mg - x1 * mgx);
            o.z = rstd * (d2 * ga.z - mg - x2 * mgx); o.w = rstd * (d3 * ga.w - mg - x3 * mgx);
            *reinterpret_cast<float4*>(dX + (int64_t)(n + j) * E + e) = o;
            dxs.x += o.x; dxs.y += o.y; dxs.z += o.z; dxs.w += o.w;
        }
    }
    if (rows == 3) lns_store_param4(part + ((int64_t)blockIdx.y * rows + 2) * E, e, clC, hw, dxs);
}

// many long samples whose waves are whole: the walk form
static bool lns_walk(int64_t N, int E) {
    return N >= 256 && E >= 4096 && E % 1024 == 0 && std::getenv("DRM_NO_LN_WALK") == nullptr;
}

static int lns_check(int64_t N, int E, const void* a, const void* b, const void* c) {
    DRM_REQUIRE(N >= 1 && N < (1 << 30) && E >= 4 && E % 4 == 0 && E <= 32768, DRM_ENOTSUP,
                "N=%lld, E=%d: E must be a multiple of 4, at most 32768", (long long)N, E);
    DRM_REQUIRE(a && b && c && aligned16(a) && aligned16(b) && aligned16(c), DRM_EINVAL, "null or misaligned pointer");
    return DRM_OK;
}

}  // namespace drm

using namespace drm;

extern "C" int32_t drm_ln_silu_slices(int64_t N, int32_t E);

extern "C" int drm_ln_silu_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int64_t N, int32_t E, float eps, const float* channel_bias, int32_t hw, int32_t channels_last,
                                   drm_stream_t stream) {
    if (int rc = lns_check(N, E, x, gamma, beta)) return rc;
    DRM_REQUIRE(y && mean && rstd && aligned16(y), DRM_EINVAL, "null or misaligned output");
    DRM_REQUIRE(!channel_bias || channels_last || (hw >= 4 && hw % 4 == 0 && E % hw == 0), DRM_EINVAL,
                "channel bias needs hw %% 4 == 0 and E %% hw == 0");
    DRM_REQUIRE(!channels_last || (hw >= 1 && E % hw == 0 && (E / hw) % 4 == 0 && (!channel_bias || aligned16(channel_bias))),
                DRM_EINVAL, "channels-last samples need hw | E and a channel count that is a multiple of 4");
    const float* cb = channel_bias;
    const int clC = channels_last ? E / hw : 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PC_ELEMENTWISE, s, 14.0 * N * E, 8.0 * N * E + 8.0 * E);
    if (E <= 2048 && clC == 0) {   // short samples: one wave each
        const dim3 rgrid((unsigned)((N + 3) / 4)), rblock(256);
        if (E <= 256) k_lnsilu_row_fwd<1><<<rgrid, rblock, 0, s>>>(x, gamma, beta, y, mean, rstd, N, E, eps, cb, hw);
        else if (E <= 512) k_lnsilu_row_fwd<2><<<rgrid, rblock, 0, s>>>(x, gamma, beta, y, mean, rstd, N, E, eps, cb, hw);
        else if (E <= 1024) k_lnsilu_row_fwd<4><<<rgrid, rblock, 0, s>>>(x, gamma, beta, y, mean, rstd, N, E, eps, cb, hw);
        else k_lnsilu_row_fwd<8><<<rgrid, rblock, 0, s>>>(x, gamma, beta, y, mean, rstd, N, E, eps, cb, hw);
        DRM_LAUNCH_CHECK();
        return DRM_OK;
    }
    if (lns_walk(N, E)) {
        if (E >= 8192) k_lns_stats<8><<<dim3((unsigned)N), dim3(256), 0, s>>>(x, mean, rstd, E, eps, cb, hw, clC);
        else k_lns_stats<4><<<dim3((unsigned)N), dim3(256), 0, s>>>(x, mean, rstd, E, eps, cb, hw, clC);
        DRM_LAUNCH_CHECK();
        const int slices = drm_ln_silu_slices(N, E);
        const int per_slice = (int)((N + slices - 1) / slices);
        k_lns_apply_walk<<<dim3(E / 1024, slices), dim3(256), 0, s>>>(x, gamma, beta, mean, rstd, y, (int)N, E, per_slice, cb, hw, clC);
        DRM_LAUNCH_CHECK();
        return DRM_OK;
    }
    const dim3 grid((unsigned)N), block(LNS_THREADS);
    if (E <= 4096) k_lnsilu_fwd<1><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, E, eps, cb, hw, clC);
    else if (E <= 8192) k_lnsilu_fwd<2><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, E, eps, cb, hw, clC);
    else if (E <= 16384) k_lnsilu_fwd<4><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, E, eps, cb, hw, clC);
    else k_lnsilu_fwd<8><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, E, eps, cb, hw, clC);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int32_t drm_ln_silu_slices(int64_t N, int32_t E) {
    // enough (element block, sample slice) workgroups to fill the chip, at least 8 samples per slice
    const int64_t eb = (E / 4 + 255) / 256;
    int64_t slices = (1024 + eb - 1) / eb;
    if (slices > N / 8) slices = N / 8;
    if (slices < 1) slices = 1;
    return (int32_t)slices;
}

extern "C" size_t drm_ln_silu_sums_floats(int64_t N, int32_t E) {
    if (N < 1 || E < 4) return 0;
    return (size_t)(2 * N) + (lns_walk(N, E) ? (size_t)N * (size_t)(E / 256) * 2 : 0);
}

extern "C" int drm_ln_silu_backward(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                    const float* rstd, float* dx, float* param_partials, int64_t N, int32_t E,
                                    const float* channel_bias, int32_t hw, int32_t channels_last, float* sample_sums,
                                    drm_stream_t stream) {
    if (int rc = lns_check(N, E, x, gamma, beta)) return rc;
    DRM_REQUIRE(!channel_bias || (sample_sums && (channels_last || (hw >= 4 && hw % 4 == 0 && E % hw == 0))), DRM_EINVAL,
                "channel bias needs hw %% 4 == 0, E %% hw == 0 and the per-sample sums buffer");
    DRM_REQUIRE(!channels_last || (hw >= 1 && E % hw == 0 && (E / hw) % 4 == 0 && (!channel_bias || aligned16(channel_bias))),
                DRM_EINVAL, "channels-last samples need hw | E and a channel count that is a multiple of 4");
    const float* cb = channel_bias;
    const int clC = channels_last ? E / hw : 0;
    DRM_REQUIRE(dy && mean && rstd && dx && param_partials && aligned16(dy) && aligned16(dx) && aligned16(param_partials),
                DRM_EINVAL, "null or misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    if (sample_sums != nullptr && lns_walk(N, E)) {
        const int slices = drm_ln_silu_slices(N, E);
        const int per_slice = (int)((N + slices - 1) / slices);
        float* wpart = sample_sums + 2 * N;       // behind the per-sample sums: drm_ln_silu_sums_floats(N, E) floats in all
        ProfScope prof(PC_ELEMENTWISE, s, 60.0 * N * E, 20.0 * N * E + 12.0 * slices * E);
        const dim3 wgrid(E / 1024, slices), wblock(256);
        k_lns_bwd_walk1<<<wgrid, wblock, 0, s>>>(x, dy, gamma, beta, mean, rstd, param_partials, 3, wpart, (int)N, E, per_slice, cb, hw, clC);
        DRM_LAUNCH_CHECK();
        k_lns_bwd_merge<<<dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s>>>(wpart, E / 256, (int)N, E, sample_sums);
        DRM_LAUNCH_CHECK();
        k_lns_bwd_walk2<<<wgrid, wblock, 0, s>>>(x, dy, gamma, beta, mean, rstd, sample_sums, dx, param_partials, 3, (int)N, E, per_slice,
                                                 cb, hw, clC);
        DRM_LAUNCH_CHECK();
        return DRM_OK;
    }
    const dim3 grid((unsigned)N), block(LNS_THREADS);
    {
        ProfScope prof(PC_ELEMENTWISE, s, 30.0 * N * E, 12.0 * N * E + 8.0 * E);
        const dim3 rgrid((unsigned)((N + 3) / 4)), rblock(256);
        if (clC != 0 && E <= 4096) k_lnsilu_bwd_dx<1><<<grid, block, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, E, cb, hw, sample_sums, clC);
        else if (E <= 256) k_lnsilu_row_bwd_dx<1><<<rgrid, rblock, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, N, E, cb, hw, sample_sums);
        else if (E <= 512) k_lnsilu_row_bwd_dx<2><<<rgrid, rblock, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, N, E, cb, hw, sample_sums);
        else if (E <= 1024) k_lnsilu_row_bwd_dx<4><<<rgrid, rblock, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, N, E, cb, hw, sample_sums);
        else if (E <= 2048) k_lnsilu_row_bwd_dx<8><<<rgrid, rblock, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, N, E, cb, hw, sample_sums);
        else if (E <= 4096) k_lnsilu_bwd_dx<1><<<grid, block, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, E, cb, hw, sample_sums, clC);
        else if (E <= 8192) k_lnsilu_bwd_dx<2><<<grid, block, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, E, cb, hw, sample_sums, clC);
        else if (E <= 16384) k_lnsilu_bwd_dx<4><<<grid, block, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, E, cb, hw, sample_sums, clC);
        else k_lnsilu_bwd_dx<8><<<grid, block, 0, s>>>(x, dy, gamma, beta, mean, rstd, dx, E, cb, hw, sample_sums, clC);
        DRM_LAUNCH_CHECK();
    }
    const int slices = drm_ln_silu_slices(N, E);
    const int per_slice = (int)((N + slices - 1) / slices);
    ProfScope prof(PC_ELEMENTWISE, s, 24.0 * N * E, 8.0 * N * E + 8.0 * slices * E);
    k_lnsilu_bwd_params<<<dim3((E / 4 + 255) / 256, slices), dim3(256), 0, s>>>(x, dy, gamma, beta, mean, rstd, param_partials,
                                                                                 (int)N, E, per_slice, cb, hw, sample_sums, clC);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
