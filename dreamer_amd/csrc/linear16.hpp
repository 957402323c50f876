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
// The kernel is latency-bound, so every global load of a wave (all of its A / B fragments, the
// LayerNorm parameters, the one-hot indices) is issued before the first use of any of them:
// CH = chunks per wave is a template parameter and the fragments live in registers.
//
// Optional prologue: the LayerNorm + SiLU of the producing layer, applied to the A fragments in
// registers; row statistics come from the fragments themselves (two-pass mean / variance, partial
// sums combined across the four waves through LDS), so the input is read exactly once.
// Epilogues: store (+ bias, + one-hot row gather), GRU gate combine, and categorical sampling.
#pragma once
#include "common.hpp"

namespace drm {

constexpr int MAX_PROBLEMS = 4;
constexpr int MAX_LATENTS = 64;   // one-hot indices staged in LDS per row

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
    // store epilogue extras: Y = (acc + bias + table rows + add[m][n]) * rowscale[m]
    const float* add; int64_t ldadd;
    const float* rowscale; int64_t ldrs;
};

struct LinLaunch {
    LinProblem p[MAX_PROBLEMS];
    int np;
    int M;
    const uint8_t* zidx; int64_t ldz; int latents, classes;   // shared by all problems that gather
};

struct GruEpilogue {       // h_new = GRU(gi = acc + b_ih, gh, h_prev)
    const float* b_ih;     // (3D)
    const float* gh; int64_t ldgh;   // (M, 3D): W_hh h (+ b_hh when b_hh == nullptr)
    const float* h_prev; int64_t ldh;
    float* h_new; int64_t ldhn;
    int D;
    // episode-boundary form (observe): gh_eff = row_scale[m] * gh + b_hh, h_prev_eff = row_scale[m] * h_prev
    const float* row_scale; int64_t ldrs;   // nullptr: no scaling
    const float* b_hh;                      // nullptr: gh already holds the bias
    float* gates_out; int64_t ldgo;         // (M, 4D) = r, u, n, gh_n saved for the backward pass, or nullptr
};

struct SampleEpilogue {    // one latent (32 classes) per workgroup
    const float* add; int64_t ldadd;         // (M, L*32) added to the logits (hoisted encoder part) or nullptr
    float* y_out; int64_t ldyo;              // (M, L*32) soft sample probabilities (for the backward pass) or nullptr
    const float* expo; int64_t lde;          // (M, L*32) Exp(1) noise, row stride lde
    const uint8_t* forced; int64_t ldf;      // (M, L) or nullptr
    uint8_t* idx_used; int64_t ldu;          // (M, L)
    uint8_t* idx_sampled; int64_t lds_;      // (M, L) or nullptr
    float* z_dense; int64_t ldzd;            // (M, L*32) or nullptr
    float* logits_out; int64_t ldlo;         // (M, L*32) or nullptr
    float log1m_eps, log_eps_over_k;
};

// Epilogues of the reverse-time (backward) observe recurrence.
struct BwdEpilogue {
    // EPI_GRUBWD (NT = 1, columns = GRU units j): dh = acc + g_h + dh_next, then the GRU cell's
    // elementwise backward -> d(gi), d(gh) for the three gates and dh * u
    const float* g_h; int64_t ldgh;        // upstream gradient wrt h_t, (M, D)
    const float* dh_next; int64_t lddn;    // gradient arriving from step t+1, (M, D)
    const float* gates; int64_t ldg;       // (M, 4D): r, u, n, gh_n saved by the forward pass
    const float* h_in; int64_t ldhi;       // (M, D)
    float* dgi; float* dgh; int64_t lddg;  // (M, 3D) each
    float* dhu; int64_t lddhu;             // (M, D)
    int D;
    // EPI_STBWD (NT = 2, one latent per workgroup): dz = acc * rowscale + g_z, then the
    // straight-through sample / unimix / log-softmax backward -> d(posterior logits)
    const float* rowscale; int64_t ldrs;   // (M) or nullptr (acc ignored when skip_acc)
    const float* g_z; const float* y; const float* wmix; const float* psoft; const float* g_post; int64_t ldl;  // (M, L*32)
    float* dpost; int64_t lddp;            // (M, L*32)
    int skip_acc;
};

enum { EPI_STORE = 0, EPI_GRU = 1, EPI_SAMPLE = 2, EPI_GRUBWD = 3, EPI_STBWD = 4 };

__device__ __forceinline__ const LinProblem& pick_problem(const LinLaunch& L, int bx, int& tile) {
    int i = L.np - 1;
    while (i > 0 && bx < L.p[i].tile_begin) --i;
    tile = bx - L.p[i].tile_begin;
    return L.p[i];
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <int NT, int EPI, int CH>
__global__ __launch_bounds__(256) void k_linear16(const LinLaunch L, const GruEpilogue G, const SampleEpilogue S,
                                                   const BwdEpilogue Bw) {
    __shared__ __attribute__((aligned(16))) float s_part[4][NT][16][16];
    __shared__ float s_red[2][4][16];
    __shared__ uint8_t s_zi[16][MAX_LATENTS];

    STAMP_DECL
    STAMP();
    int tile;
    const LinProblem& P = pick_problem(L, blockIdx.x, tile);
    const int m0 = blockIdx.y * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int K = P.K;
    const bool ln = P.ln_g != nullptr;

    int nbase[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (EPI == EPI_GRU) nbase[t] = tile * 16 + t * G.D;        // gates r,z,n of the same unit
        else nbase[t] = (tile * NT + t) * 16;
    }

    const float* xrow = P.X + (int64_t)min(m0 + r, L.M - 1) * P.ldx + q * 8;
    const float* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = P.W + (int64_t)min(nbase[t] + r, P.N - 1) * P.ldw + q * 8;
    if (EPI == EPI_STORE && P.tab != nullptr) {
        for (int i = threadIdx.x; i < 16 * L.latents; i += 256) {
            const int row = i / L.latents, l = i - row * L.latents;
            s_zi[row][l] = L.zidx[(int64_t)min(m0 + row, L.M - 1) * L.ldz + l];
        }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K is walked in passes of CH chunks per wave; inside a pass every load is issued before the first
    // use of any of them (the kernel is latency-bound).  LayerNorm needs the whole row: one pass only.
    for (int k0 = 0; k0 < K; k0 += CH * 128) {
        float a[CH][8], b[CH][NT][8], g[CH][8], e[CH][8];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kc = k0 + (c * 4 + wave) * 32;
            if (kc < K) {
                load8(xrow + kc, a[c]);
#pragma unroll
                for (int t = 0; t < NT; ++t) load8(wrow[t] + kc, b[c][t]);
                if (ln) { load8(P.ln_g + kc + q * 8, g[c]); load8(P.ln_b + kc + q * 8, e[c]); }
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) a[c][s] = 0.f;
            }
        }
#ifdef DRM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP();
#endif
        // ---- LayerNorm + SiLU of the producing layer, from the fragments in registers ----
        if (ln) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) s += a[c][i];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (q == 0) s_red[0][wave][r] = s;
            __syncthreads();
            const float mean = ((s_red[0][0][r] + s_red[0][1][r]) + (s_red[0][2][r] + s_red[0][3][r])) / (float)K;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((c * 4 + wave) * 32 < K) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float d = a[c][i] - mean; v += d * d; }
                }
            }
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) s_red[1][wave][r] = v;
            __syncthreads();
            const float var = ((s_red[1][0][r] + s_red[1][1][r]) + (s_red[1][2][r] + s_red[1][3][r])) / (float)K;
            const float rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((c * 4 + wave) * 32 < K) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[c][i] = silu_f((a[c][i] - mean) * rstd * g[c][i] + e[c][i]);
                }
            }
        }
#ifdef DRM_STAMPS
        STAMP();
#endif
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (k0 + (c * 4 + wave) * 32 < K) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s], b[c][t][s], acc[t], 0, 0, 0);
            }
        }
    }

    // split-K combine: C layout of the 16x16x4 MFMA is row = 4*(lane>>4) + reg, col = lane & 15
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wave][t][q * 4 + i][r] = acc[t][i];
    STAMP();
    __syncthreads();
    STAMP();

    STAMP_FLUSH((unsigned long long)(EPI * 100 + NT * 10 + (ln ? 1 : 0)));
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
                const float* tb = P.tab + n;
                int l = 0;
                for (; l + 8 <= L.latents; l += 8) {   // eight independent row reads in flight
                    float u[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) u[i] = tb[(int64_t)((l + i) * L.classes + s_zi[row][l + i]) * P.ldt];
#pragma unroll
                    for (int i = 0; i < 8; ++i) y += u[i];
                }
                for (; l < L.latents; ++l) y += tb[(int64_t)(l * L.classes + s_zi[row][l]) * P.ldt];
            }
            if (P.add != nullptr) y += P.add[(int64_t)m * P.ldadd + n];
            if (P.rowscale != nullptr) y *= P.rowscale[(int64_t)m * P.ldrs];
            P.Y[(int64_t)m * P.ldy + n] = y;
        }
    } else if (EPI == EPI_GRUBWD) {
        if (m >= L.M) return;
        const int j = tile * 16 + col;
        if (j >= Bw.D) return;
        const int D = Bw.D;
        const float* gt = Bw.gates + (int64_t)m * Bw.ldg;
        const float rr = gt[j], uu = gt[D + j], nn = gt[2 * D + j], ghn = gt[3 * D + j];
        const float hin = Bw.h_in[(int64_t)m * Bw.ldhi + j];
        const float dh = v[0] + Bw.g_h[(int64_t)m * Bw.ldgh + j] + Bw.dh_next[(int64_t)m * Bw.lddn + j];
        const float dn_pre = dh * (1.0f - uu) * (1.0f - nn * nn);
        const float du_pre = dh * (hin - nn) * uu * (1.0f - uu);
        const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
        float* gi = Bw.dgi + (int64_t)m * Bw.lddg;
        float* gh = Bw.dgh + (int64_t)m * Bw.lddg;
        gi[j] = dr_pre; gi[D + j] = du_pre; gi[2 * D + j] = dn_pre;
        gh[j] = dr_pre; gh[D + j] = du_pre; gh[2 * D + j] = dn_pre * rr;
        Bw.dhu[(int64_t)m * Bw.lddhu + j] = dh * uu;
    } else if (EPI == EPI_STBWD) {   // NT == 2: this workgroup holds the 32 classes of latent `tile` for 16 rows
        const int lat = tile;
        const int mm = min(m, L.M - 1);
        const int64_t o0 = (int64_t)mm * Bw.ldl + lat * 32 + col, o1 = o0 + 16;
        const float rs = Bw.rowscale ? Bw.rowscale[(int64_t)mm * Bw.ldrs] : 1.0f;
        const float dz0 = (Bw.skip_acc ? 0.f : v[0] * rs) + Bw.g_z[o0];
        const float dz1 = (Bw.skip_acc ? 0.f : v[1] * rs) + Bw.g_z[o1];
        const float y0 = Bw.y[o0], y1 = Bw.y[o1];
        const float s1 = group_sum<16>(y0 * dz0 + y1 * dz1);
        const float dl0 = y0 * (dz0 - s1) * Bw.wmix[o0], dl1 = y1 * (dz1 - s1) * Bw.wmix[o1];
        const float s2 = group_sum<16>(dl0 + dl1);
        if (m >= L.M) return;
        float* dp = Bw.dpost + (int64_t)m * Bw.lddp + lat * 32;
        dp[col] = dl0 - Bw.psoft[o0] * s2 + Bw.g_post[o0];
        dp[16 + col] = dl1 - Bw.psoft[o1] * s2 + Bw.g_post[o1];
    } else if (EPI == EPI_GRU) {
        if (m >= L.M) return;
        const int j = tile * 16 + col;
        if (j >= G.D) return;
        const float* gh = G.gh + (int64_t)m * G.ldgh;
        float ghr = gh[j], ghu = gh[G.D + j], ghn = gh[2 * G.D + j];
        const float bir = G.b_ih[j], biu = G.b_ih[G.D + j], bin = G.b_ih[2 * G.D + j];
        float hp = G.h_prev[(int64_t)m * G.ldh + j];
        if (G.row_scale != nullptr) {
            const float rs = G.row_scale[(int64_t)m * G.ldrs];
            ghr *= rs; ghu *= rs; ghn *= rs; hp *= rs;
        }
        if (G.b_hh != nullptr) { ghr += G.b_hh[j]; ghu += G.b_hh[G.D + j]; ghn += G.b_hh[2 * G.D + j]; }
        const float rr = sigmoid_f(v[0] + bir + ghr), uu = sigmoid_f(v[1] + biu + ghu);
        const float nn = tanhf((v[2] + bin) + rr * ghn);
        G.h_new[(int64_t)m * G.ldhn + j] = (1.0f - uu) * nn + uu * hp;
        if (G.gates_out != nullptr) {
            float* go = G.gates_out + (int64_t)m * G.ldgo;
            go[j] = rr; go[G.D + j] = uu; go[2 * G.D + j] = nn; go[3 * G.D + j] = ghn;
        }
    } else if (&P != &L.p[0]) {  // EPI_SAMPLE launch, companion problem: plain store of a 16 x 32 tile
        if (m >= L.M) return;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = nbase[t] + col;
            if (n < P.N) P.Y[(int64_t)m * P.ldy + n] = v[t] + (P.bias ? P.bias[n] : 0.f);
        }
    } else {  // EPI_SAMPLE: NT == 2, this workgroup holds the 32 logits of latent `tile` for 16 rows
        const int lat = tile;
        const int mm = min(m, L.M - 1);
        const float* ex = S.expo + (int64_t)mm * S.lde + lat * 32;
        const float e0 = ex[col], e1 = ex[16 + col];
        float lg[2], sc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            lg[t] = v[t] + (P.bias ? P.bias[lat * 32 + t * 16 + col] : 0.f);
            if (S.add != nullptr) lg[t] += S.add[(int64_t)mm * S.ldadd + lat * 32 + t * 16 + col];
        }
        const float mx = group_max<16>(fmaxf(lg[0], lg[1]));
        const float lse = logf(group_sum<16>(expf(lg[0] - mx) + expf(lg[1] - mx)));
        sc[0] = logaddexp_f((lg[0] - mx - lse) + S.log1m_eps, S.log_eps_over_k) - logf(e0);
        sc[1] = logaddexp_f((lg[1] - mx - lse) + S.log1m_eps, S.log_eps_over_k) - logf(e1);
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
        if (S.y_out) {
            float* yo = S.y_out + (int64_t)m * S.ldyo + lat * 32;
            yo[col] = y0;
            yo[16 + col] = y1;
        }
    }
}

// Store-epilogue variant for a few hundred rows (K <= 512, one pass): one workgroup = (MT x 16) rows x
// (NT x 16) columns.  Every A fragment feeds NT tiles and every B fragment MT tiles, so the L2 reads
// per output drop by 2 MT NT / (MT + NT), and the grid (a quarter of the 16 x 16 form's at MT = NT = 2)
// is resident in one round instead of queueing several latency-bound rounds per CU.
template <int MT, int NT>
__global__ __launch_bounds__(256) void k_linear16m(const LinLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_part[4][MT * NT][16][16];
    __shared__ float s_red[2][4][MT * 16];
    __shared__ uint8_t s_zi[MT * 16][MAX_LATENTS];

    STAMP_DECL_AT((blockIdx.x == 0 && blockIdx.y == 0) || (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1))
    STAMP();
    int tile;
    const LinProblem& P = pick_problem(L, blockIdx.x, tile);
    const int m0 = blockIdx.y * (MT * 16), n0 = tile * (NT * 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int K = P.K;
    const bool ln = P.ln_g != nullptr;

    const float* xrow[MT];
    const float* wrow[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xrow[i] = P.X + (int64_t)min(m0 + i * 16 + r, L.M - 1) * P.ldx + q * 8;
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = P.W + (int64_t)min(n0 + t * 16 + r, P.N - 1) * P.ldw + q * 8;
    if (P.tab != nullptr) {
        for (int i = threadIdx.x; i < MT * 16 * L.latents; i += 256) {
            const int row = i / L.latents, l = i - row * L.latents;
            s_zi[row][l] = L.zidx[(int64_t)min(m0 + row, L.M - 1) * L.ldz + l];
        }
    }

    // every load of the wave is issued before the first use of any of them
    float a[4][MT][8], b[4][NT][8], g[4][8], e[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kc = (c * 4 + wave) * 32;
        if (kc < K) {
#pragma unroll
            for (int i = 0; i < MT; ++i) load8(xrow[i] + kc, a[c][i]);
#pragma unroll
            for (int t = 0; t < NT; ++t) load8(wrow[t] + kc, b[c][t]);
            if (ln) { load8(P.ln_g + kc + q * 8, g[c]); load8(P.ln_b + kc + q * 8, e[c]); }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[c][i][j] = 0.f;
        }
    }
#ifdef DRM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
#endif
    if (ln) {   // LayerNorm + SiLU of the producing layer on the A fragments (two-pass statistics per row)
        float mean[MT], rstd[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) sm += a[c][i][j];
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            if (q == 0) s_red[0][wave][i * 16 + r] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rr = i * 16 + r;
            mean[i] = ((s_red[0][0][rr] + s_red[0][1][rr]) + (s_red[0][2][rr] + s_red[0][3][rr])) / (float)K;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if ((c * 4 + wave) * 32 < K) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = a[c][i][j] - mean[i]; v += d * d; }
                }
            }
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) s_red[1][wave][rr] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rr = i * 16 + r;
            const float var = ((s_red[1][0][rr] + s_red[1][1][rr]) + (s_red[1][2][rr] + s_red[1][3][rr])) / (float)K;
            rstd[i] = 1.0f / sqrtf(var + LN_EPS);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if ((c * 4 + wave) * 32 < K) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[c][i][j] = silu_f((a[c][i][j] - mean[i]) * rstd[i] * g[c][j] + e[c][j]);
            }
        }
    }
    STAMP();
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if ((c * 4 + wave) * 32 < K) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i][j], b[c][t][j], acc[i][t], 0, 0, 0);
        }
    }
    // split-K combine in a fixed order (C layout: row = 4*(lane>>4) + reg, col = lane & 15)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_part[wave][i * NT + t][q * 4 + j][r] = acc[i][t][j];
    STAMP();
    __syncthreads();
    STAMP();
    STAMP_FLUSH((unsigned long long)(5000 + (blockIdx.x == 0 ? 0 : 1)));

    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    float y[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int it = i * NT + t, n = min(n0 + t * 16 + col, P.N - 1);
            y[i][t] = ((s_part[0][it][row][col] + s_part[1][it][row][col]) + s_part[2][it][row][col]) + s_part[3][it][row][col];
            y[i][t] += P.bias ? P.bias[n] : 0.f;
        }
    if (P.tab != nullptr) {
        // one-hot rows of the table: the reads of a whole latent group for all of the thread's outputs are
        // issued together (the table is cold in L2, and a dependent chain of small batches pays the miss
        // latency once per batch)
        for (int l0 = 0; l0 < L.latents; l0 += 16) {
            float u[MT][NT][16];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint8_t* zi = s_zi[i * 16 + row];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float* tb = P.tab + min(n0 + t * 16 + col, P.N - 1);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int l = min(l0 + k, L.latents - 1);
                        u[i][t][k] = tb[(int64_t)(l * L.classes + zi[l]) * P.ldt];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 16; ++k) y[i][t] += (l0 + k < L.latents) ? u[i][t][k] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + row;
        if (m >= L.M) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 16 + col;
            if (n >= P.N) continue;
            float v = y[i][t];
            if (P.add != nullptr) v += P.add[(int64_t)m * P.ldadd + n];
            if (P.rowscale != nullptr) v *= P.rowscale[(int64_t)m * P.ldrs];
            P.Y[(int64_t)m * P.ldy + n] = v;
        }
    }
}

// CH (chunks of 32 per wave and pass) for a K: 1, 2 or 4; K > 512 takes several passes of 4
inline int chunks_for(int K) { return K <= 128 ? 1 : (K <= 256 ? 2 : 4); }

}  // namespace drm
