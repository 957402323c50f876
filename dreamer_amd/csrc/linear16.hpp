// Small-row fused linear block on fp32 MFMA (v_mfma_f32_16x16x4_f32), for row counts from one
// 16-row tile up to a few hundred rows, where the chain of dependent layers is latency-bound and
// the output columns have to be spread over as many CUs as possible.
//
// One workgroup = 4 waves = one 16-row x (NT x 16)-column output tile.  The K dimension is cut
// into 32-wide chunks dealt round-robin to the 4 waves (split-K inside the workgroup); each wave
// reads its A and B fragments straight from global memory / L2 in the tensors' natural layouts:
//     lane (r = lane & 15, q = lane >> 4) loads X[m0 + r][kc + 8q .. 8q+7] and W[n0 + r][same k]
// (two 16-byte loads each; a wave instruction covers 16 rows x 128 contiguous bytes).  Element s
// of those eight is the operand of MFMA s of the chunk: the instruction's k index is q, so the
// products are summed over k = kc + 8q + s for q = 0..3, the same permutation for A and B.  The
// four waves' partial tiles are added in a fixed order through LDS, so results are reproducible.
//
// Optional prologue: the LayerNorm + SiLU of the producing layer, applied to the A fragment as
// it is loaded (row statistics are computed once per workgroup with a two-pass mean / variance).
// Epilogues: store (+ bias, + one-hot row gather), GRU gate combine, and categorical sampling.
#pragma once
#include "common.hpp"

namespace drm {

constexpr int MAX_PROBLEMS = 4;

// One GEMM of a launch: Y[M,N] = f(X)[M,K] W[N,K]^T (+ bias, + one-hot table rows)
struct LinProblem {
    const float* X; int64_t ldx;
    const float* ln_g; const float* ln_b;     // LayerNorm+SiLU on X rows when ln_g != nullptr
    const float* W; int64_t ldw;
    const float* bias;
    const float* tab; int64_t ldt;            // one-hot gather table [(l*classes + idx)][n]
    float* Y; int64_t ldy;
    int N, K;
    int tile_begin;                            // first blockIdx.x of this problem
};

struct LinLaunch {
    LinProblem p[MAX_PROBLEMS];
    int np;
    int M;
    const uint8_t* zidx; int64_t ldz; int latents, classes;   // shared by all problems that gather
};

struct GruEpilogue {       // h_new = GRU(gi = acc + b_ih, gh (precomputed, bias included), h_prev)
    const float* b_ih;     // (3D)
    const float* gh; int64_t ldgh;   // (M, 3D)
    const float* h_prev; int64_t ldh;
    float* h_new; int64_t ldhn;
    int D;
};

struct SampleEpilogue {    // one latent (32 classes) per workgroup
    const float* expo; int64_t lde;          // (M, L*32) Exp(1) noise, row stride lde
    const uint8_t* forced; int64_t ldf;      // (M, L) or nullptr
    uint8_t* idx_used; int64_t ldu;          // (M, L)
    uint8_t* idx_sampled; int64_t lds_;      // (M, L) or nullptr
    float* z_dense; int64_t ldzd;            // (M, L*32) or nullptr
    float* logits_out; int64_t ldlo;         // (M, L*32) or nullptr
    float log1m_eps, log_eps_over_k;
};

enum { EPI_STORE = 0, EPI_GRU = 1, EPI_SAMPLE = 2 };

__device__ __forceinline__ const LinProblem& pick_problem(const LinLaunch& L, int bx, int& tile) {
    int i = L.np - 1;
    while (i > 0 && bx < L.p[i].tile_begin) --i;
    tile = bx - L.p[i].tile_begin;
    return L.p[i];
}

// Row statistics of the 16 rows of this m-tile over K columns: 16 lanes per row, two passes.
__device__ __forceinline__ void tile_row_stats(const float* X, int64_t ldx, int m0, int M, int K, float* s_mean,
                                               float* s_rstd) {
    const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int m = min(m0 + row, M - 1);
    const float* x = X + (int64_t)m * ldx;
    float s = 0.f;
    for (int k = c * 4; k < K; k += 64) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = group_sum<16>(s) / (float)K;
    float q = 0.f;
    for (int k = c * 4; k < K; k += 64) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
    }
    const float var = group_sum<16>(q) / (float)K;
    if (c == 0) {
        s_mean[row] = mean;
        s_rstd[row] = 1.0f / sqrtf(var + LN_EPS);
    }
}

template <int NT, int EPI>
__global__ __launch_bounds__(256) void k_linear16(const LinLaunch L, const GruEpilogue G, const SampleEpilogue S) {
    __shared__ __attribute__((aligned(16))) float s_part[4][NT][16][16];
    __shared__ float s_mean[16], s_rstd[16];

    int tile;
    const LinProblem& P = pick_problem(L, blockIdx.x, tile);
    const int m0 = blockIdx.y * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int K = P.K;

    // column base of each of this workgroup's NT tiles
    int nbase[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (EPI == EPI_GRU) nbase[t] = tile * 16 + t * G.D;        // gates r,z,n of the same unit
        else nbase[t] = (tile * NT + t) * 16;
    }

    const bool ln = P.ln_g != nullptr;
    if (ln) {
        tile_row_stats(P.X, P.ldx, m0, L.M, K, s_mean, s_rstd);
        __syncthreads();
    }
    const float mean = ln ? s_mean[r] : 0.f, rstd = ln ? s_rstd[r] : 1.f;

    const float* xrow = P.X + (int64_t)min(m0 + r, L.M - 1) * P.ldx + q * 8;
    const float* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = P.W + (int64_t)min(nbase[t] + r, P.N - 1) * P.ldw + q * 8;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
    for (int kc = wave * 32; kc < K; kc += 128) {
        float a[8];
        {
            const float4 a0 = *reinterpret_cast<const float4*>(xrow + kc);
            const float4 a1 = *reinterpret_cast<const float4*>(xrow + kc + 4);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
            a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        }
        float b[NT][8];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b0 = *reinterpret_cast<const float4*>(wrow[t] + kc);
            const float4 b1 = *reinterpret_cast<const float4*>(wrow[t] + kc + 4);
            b[t][0] = b0.x; b[t][1] = b0.y; b[t][2] = b0.z; b[t][3] = b0.w;
            b[t][4] = b1.x; b[t][5] = b1.y; b[t][6] = b1.z; b[t][7] = b1.w;
        }
        if (ln) {
            const float4 g0 = *reinterpret_cast<const float4*>(P.ln_g + kc + q * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(P.ln_g + kc + q * 8 + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(P.ln_b + kc + q * 8);
            const float4 e1 = *reinterpret_cast<const float4*>(P.ln_b + kc + q * 8 + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int s = 0; s < 8; ++s) a[s] = silu_f((a[s] - mean) * rstd * g[s] + e[s]);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
        }
    }

    // split-K combine: C layout of the 16x16x4 MFMA is row = 4*(lane>>4) + reg, col = lane & 15
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wave][t][q * 4 + i][r] = acc[t][i];
    __syncthreads();

    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const int m = m0 + row;
    float v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        v[t] = ((s_part[0][t][row][col] + s_part[1][t][row][col]) + s_part[2][t][row][col]) + s_part[3][t][row][col];

    if (EPI == EPI_STORE) {
        if (m >= L.M) return;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = nbase[t] + col;
            if (n >= P.N) continue;
            float y = v[t] + (P.bias ? P.bias[n] : 0.f);
            if (P.tab != nullptr) {
                const uint8_t* zi = L.zidx + (int64_t)m * L.ldz;
                for (int l = 0; l < L.latents; ++l)
                    y += P.tab[(int64_t)(l * L.classes + zi[l]) * P.ldt + n];
            }
            P.Y[(int64_t)m * P.ldy + n] = y;
        }
    } else if (EPI == EPI_GRU) {
        if (m >= L.M) return;
        const int j = tile * 16 + col;
        if (j >= G.D) return;
        const float* gh = G.gh + (int64_t)m * G.ldgh;
        const float gr = v[0] + G.b_ih[j] + gh[j];
        const float gu = v[1] + G.b_ih[G.D + j] + gh[G.D + j];
        const float rr = sigmoid_f(gr), uu = sigmoid_f(gu);
        const float nn = tanhf((v[2] + G.b_ih[2 * G.D + j]) + rr * gh[2 * G.D + j]);
        const float hp = G.h_prev[(int64_t)m * G.ldh + j];
        G.h_new[(int64_t)m * G.ldhn + j] = (1.0f - uu) * nn + uu * hp;
    } else {  // EPI_SAMPLE: NT == 2, this workgroup holds the 32 logits of latent `tile` for 16 rows
        const int lat = tile;
        const int mm = min(m, L.M - 1);
        float lg[2], sc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) lg[t] = v[t] + P.bias[lat * 32 + t * 16 + col];
        const float mx = group_max<16>(fmaxf(lg[0], lg[1]));
        const float lse = logf(group_sum<16>(expf(lg[0] - mx) + expf(lg[1] - mx)));
        const float* ex = S.expo + (int64_t)mm * S.lde + lat * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float lp = logaddexp_f((lg[t] - mx - lse) + S.log1m_eps, S.log_eps_over_k);
            sc[t] = lp - logf(ex[t * 16 + col]);
        }
        const float smax = group_max<16>(fmaxf(sc[0], sc[1]));
        float y0 = expf(sc[0] - smax), y1 = expf(sc[1] - smax);
        const float ssum = group_sum<16>(y0 + y1);
        y0 /= ssum; y1 /= ssum;
        float best = y0; int bi = col;
        if (y1 > y0) { best = y1; bi = 16 + col; }
        group_argmax<16>(best, bi);
        if (m >= L.M) return;
        int used = bi;
        if (S.forced != nullptr) used = S.forced[(int64_t)m * S.ldf + lat];
        if (col == 0) {
            S.idx_used[(int64_t)m * S.ldu + lat] = (uint8_t)used;
            if (S.idx_sampled) S.idx_sampled[(int64_t)m * S.lds_ + lat] = (uint8_t)bi;
        }
        if (S.z_dense) {
            float* zd = S.z_dense + (int64_t)m * S.ldzd + lat * 32;
            zd[col] = used == col ? 1.f : 0.f;
            zd[16 + col] = used == 16 + col ? 1.f : 0.f;
        }
        if (S.logits_out) {
            float* lo = S.logits_out + (int64_t)m * S.ldlo + lat * 32;
            lo[col] = lg[0];
            lo[16 + col] = lg[1];
        }
    }
}

}  // namespace drm
