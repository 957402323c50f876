// Few-row linear block for the latency-bound regime (a handful of trajectories: the reference's default batch of 16).
//
// With 16 rows a 512 x 512 layer is 8 MFLOP against 1 MB of weights: nothing is compute-bound, and a
// launch costs what its slowest workgroup needs to FETCH.  A CU sustains only a few tens of GB/s on
// loads that miss L2 (weights are cold in L2 at every kernel start), so the layer is cut into as many
// workgroups as there are CUs: each workgroup owns CP output columns (2..8) and 4*RW rows, i.e. it
// fetches CP weight rows (a few KB, cold) plus its input rows (hot in L2: every workgroup of the launch
// reads the same ones).  The products run on the vector ALUs -- an MFMA tile would need 16 columns per
// workgroup, which is 8x fewer workgroups and 8x more cold bytes each.
//
// Work split: wave w of the 4 owns rows 4*RW*tile + RW*w .. +RW-1 (all CP columns, all of K); lane l
// owns k = 4l..4l+3 and 256+4l..256+4l+3 (K <= 512), so every global load is a full 16-byte-per-lane
// row read.  Waves never talk to each other: LayerNorm statistics and the K-reduction of the RW*CP
// partial sums are wave-level butterflies in which the number of live values halves per step, so a
// reduction of NV values costs NV-1 + (6 - log2 NV) shuffles.  No LDS, no barrier.
#pragma once
#include "common.hpp"

namespace drm {

constexpr int SKINNY_MAX_PROBLEMS = 4;

struct SkinnyProblem {
    const float* X; int64_t ldx;               // (M, K)
    const float* ln_g; const float* ln_b;     // LayerNorm + SiLU on X rows when ln_g != nullptr
    const float* W; int64_t ldw;               // (N, K) torch Linear layout
    const float* bias;                         // (N) or nullptr
    float* Y; int64_t ldy;
    int N, K;
    int wg_begin;                              // first blockIdx.x of this problem
    const float* add; int64_t ldadd;           // Y += add[m][n] or nullptr
};

struct SkinnyLaunch {
    SkinnyProblem p[SKINNY_MAX_PROBLEMS];
    int np, M;
};

// Sum over the 64 lanes of a wave of each of NV per-lane values (NV = 2^LG <= 64) with a butterfly in
// which the number of live values halves per level: level j pairs lane ^ 2^j, the lane whose bit j is
// set keeps the upper half of the values and sends the lower half (the other lane the reverse), both
// add what they receive.  The most exchanges happen at the levels with the cheapest partner
// (xor_lane<1>, <2>: one DPP move).  Returns, on every lane, the total of value number
//   idx(lane) = sum_{j < LG} bit_j(lane) << (LG - 1 - j);
// lanes 0 .. NV-1 hold one total each, the other lanes repeat them.
template <int NV>
__device__ __forceinline__ float halving_sum64(float (&v)[NV], int lane) {
    constexpr int LG = NV >= 64 ? 6 : NV >= 32 ? 5 : NV >= 16 ? 4 : NV >= 8 ? 3 : NV >= 4 ? 2 : NV >= 2 ? 1 : 0;
#define DRM_HALVE(j)                                                        \
    if constexpr (LG > (j)) {                                               \
        constexpr int n = NV >> (j);                                        \
        const bool up = (lane & (1 << (j))) != 0;                           \
        _Pragma("unroll") for (int i = 0; i < n / 2; ++i) {                 \
            const float keep = up ? v[i + n / 2] : v[i];                    \
            const float send = up ? v[i] : v[i + n / 2];                    \
            v[i] = keep + xor_lane<(1 << (j))>(send);                       \
        }                                                                   \
    } else {                                                                \
        v[0] += xor_lane<(1 << (j))>(v[0]);                                 \
    }
    DRM_HALVE(0) DRM_HALVE(1) DRM_HALVE(2) DRM_HALVE(3) DRM_HALVE(4) DRM_HALVE(5)
#undef DRM_HALVE
    return v[0];
}
template <int NV>
__device__ __forceinline__ int halving_index(int lane) {
    constexpr int LG = NV >= 64 ? 6 : NV >= 32 ? 5 : NV >= 16 ? 4 : NV >= 8 ? 3 : NV >= 4 ? 2 : NV >= 2 ? 1 : 0;
    int idx = 0;
#pragma unroll
    for (int j = 0; j < LG; ++j) idx |= ((lane >> j) & 1) << (LG - 1 - j);
    return idx;
}

// Sum over the 64 lanes, on every lane, of each of NV per-lane values (independent butterflies: six
// dependent levels whatever NV is).
template <int NV>
__device__ __forceinline__ void all_sum64(float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = group_sum<64>(v[i]);
}

// predicated 16-byte load (the zero is a prvalue on purpose: selecting between two lvalues would turn
// into a select of ADDRESSES followed by a generic-address load)
__device__ __forceinline__ float4 ld4(const float* p, bool on) {
    if (on) return *reinterpret_cast<const float4*>(p);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int RW, int CP>
__global__ __launch_bounds__(256) void k_skinny(const SkinnyLaunch L) {
    constexpr int NV = RW * CP;
    static_assert(NV >= 2 && NV <= 64 && (NV & (NV - 1)) == 0, "RW*CP must be a power of two in [2, 64]");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int pi = L.np - 1;
    while (pi > 0 && (int)blockIdx.x < L.p[pi].wg_begin) --pi;
    const SkinnyProblem& P = L.p[pi];
    const int cg = blockIdx.x - P.wg_begin;            // column group of this workgroup
    const int K = P.K;
    const int kA = 4 * lane, kB = 256 + 4 * lane;
    const bool vA = kA < K, vB = kB < K;
    const int row0 = blockIdx.y * (4 * RW) + wave * RW;
    const bool ln = P.ln_g != nullptr;
    STAMP_DECL
    STAMP();

    // ---- every load of the kernel is issued here, before the first use
    float x[RW][8];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float* xp = P.X + (int64_t)min(row0 + r, L.M - 1) * P.ldx;
        const float4 a = ld4(xp + kA, vA), b = ld4(xp + kB, vB);
        x[r][0] = a.x; x[r][1] = a.y; x[r][2] = a.z; x[r][3] = a.w;
        x[r][4] = b.x; x[r][5] = b.y; x[r][6] = b.z; x[r][7] = b.w;
    }
    float g[8], e[8];
    {
        const float4 ga = ld4(P.ln_g + kA, ln && vA), gb = ld4(P.ln_g + kB, ln && vB);
        const float4 ea = ld4(P.ln_b + kA, ln && vA), eb = ld4(P.ln_b + kB, ln && vB);
        g[0] = ga.x; g[1] = ga.y; g[2] = ga.z; g[3] = ga.w; g[4] = gb.x; g[5] = gb.y; g[6] = gb.z; g[7] = gb.w;
        e[0] = ea.x; e[1] = ea.y; e[2] = ea.z; e[3] = ea.w; e[4] = eb.x; e[5] = eb.y; e[6] = eb.z; e[7] = eb.w;
    }
    float w[CP][8];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int n = cg * CP + c;
        const bool ok = n < P.N;
        const float* wp = P.W + (int64_t)(ok ? n : 0) * P.ldw;
        const float4 a = ld4(wp + kA, ok && vA), b = ld4(wp + kB, ok && vB);
        w[c][0] = a.x; w[c][1] = a.y; w[c][2] = a.z; w[c][3] = a.w;
        w[c][4] = b.x; w[c][5] = b.y; w[c][6] = b.z; w[c][7] = b.w;
    }
    // the output element this lane will hold after the reduction
    const int idx = halving_index<NV>(lane);
    const int m = row0 + idx / CP;
    const int n = cg * CP + idx % CP;
    const bool writer = lane < NV && m < L.M && n < P.N;
    float bias = (writer && P.bias != nullptr) ? P.bias[n] : 0.f;
    if (writer && P.add != nullptr) bias += P.add[(int64_t)m * P.ldadd + n];

#ifdef DRM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
#endif
    // ---- LayerNorm + SiLU of the input rows.  One pass: sums of (x - c) and (x - c)^2 with c the
    // row's first element as a shift (keeps the variance free of cancellation), every row's two sums
    // reduced together.
    if (ln) {
        float st[2 * RW];
        float shift[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            shift[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x[r][0])));
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool on = i < 4 ? vA : vB;
                const float d = on ? x[r][i] - shift[r] : 0.f;
                s += d;
                q = fmaf(d, d, q);
            }
            st[2 * r] = s; st[2 * r + 1] = q;
        }
        all_sum64<2 * RW>(st);
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float md = st[2 * r] * inv_k;                       // mean - shift
            const float var = fmaxf(fmaf(-md, md, st[2 * r + 1] * inv_k), 0.f);
            const float rstd = __builtin_amdgcn_rsqf(var + LN_EPS);
            const float mean = shift[r] + md;
            // lanes beyond K hold g = e = 0, so they come out as SiLU(0) = 0 without a select
#pragma unroll
            for (int i = 0; i < 8; ++i) x[r][i] = silu_f(fmaf((x[r][i] - mean) * rstd, g[i], e[i]));
        }
    }

    STAMP();
    // ---- partial dot products over this lane's 8 k, then the wave-level K reduction
    float acc[NV];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) a = fmaf(x[r][i], w[c][i], a);
            acc[r * CP + c] = a;
        }
    const float tot = halving_sum64<NV>(acc, lane);
#ifdef DRM_STAMPS
    if (tot == 12345.678f) return;   // make the stamp wait for the reduction
    STAMP();
    STAMP_FLUSH((unsigned long long)(1000 + RW * 10 + CP));
#endif
    if (writer) P.Y[(int64_t)m * P.ldy + n] = tot + bias;
}

// ---- deeper K (up to 512 * KC) without LayerNorm: the reverse-time observe scan -----------------------
// Epilogue 0: Y = acc + bias.  Epilogue 1 (one problem, columns = GRU units j): the GRU cell's
// elementwise backward (what autograd computes for torch.nn.GRUCell):
//     dh = acc + g_h + next_scale * (dhu_next + dhw_next)        (the last two: the path through h_{t+1})
//     d(gi_n) = dh (1-u)(1-n^2),  d(gh_n) = d(gi_n) r,  d(gi_r) = d(gh_r) = d(gi_n) gh_n r (1-r),
//     d(gi_u) = d(gh_u) = dh (h_in - n) u (1-u),  dhu = dh u
struct SkinnyGruBwd {
    const float* g_h; int64_t ldgh;          // (M, D) upstream gradient wrt h_t
    const float* dhu_next;                   // (M, D) dh*u of step t+1 (row stride D) or nullptr at the last step
    const float* dhw_next;                   // (M, D) d(gh)[t+1] W_hh (row stride D)
    const float* next_scale; int64_t ldns;   // (M) continue flag of step t+1
    const float* gates; int64_t ldg;         // (M, 4D): r, u, n, gh_n
    const float* h_in; int64_t ldhi;         // (M, D)
    float* dgi; float* dgh; int64_t lddg;    // (M, 3D)
    float* dhu;                              // (M, D), row stride D (may alias dhu_next)
    int D;
};

template <int RW, int CP, int KC, int EPI>
__global__ __launch_bounds__(256) void k_skinny_deep(const SkinnyLaunch L, const SkinnyGruBwd G) {
    constexpr int NV = RW * CP;
    static_assert(NV >= 2 && NV <= 64 && (NV & (NV - 1)) == 0, "RW*CP must be a power of two in [2, 64]");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int pi = L.np - 1;
    while (pi > 0 && (int)blockIdx.x < L.p[pi].wg_begin) --pi;
    const SkinnyProblem& P = L.p[pi];
    const int cg = blockIdx.x - P.wg_begin;
    const int K = P.K;
    const int row0 = blockIdx.y * (4 * RW) + wave * RW;

    float x[RW][8 * KC], w[CP][8 * KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const int kA = kc * 512 + 4 * lane, kB = kA + 256;
        const bool vA = kA < K, vB = kB < K;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float* xp = P.X + (int64_t)min(row0 + r, L.M - 1) * P.ldx;
            const float4 a = ld4(xp + kA, vA), b = ld4(xp + kB, vB);
            x[r][8 * kc + 0] = a.x; x[r][8 * kc + 1] = a.y; x[r][8 * kc + 2] = a.z; x[r][8 * kc + 3] = a.w;
            x[r][8 * kc + 4] = b.x; x[r][8 * kc + 5] = b.y; x[r][8 * kc + 6] = b.z; x[r][8 * kc + 7] = b.w;
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const int n = cg * CP + c;
            const bool ok = n < P.N;
            const float* wp = P.W + (int64_t)(ok ? n : 0) * P.ldw;
            const float4 a = ld4(wp + kA, ok && vA), b = ld4(wp + kB, ok && vB);
            w[c][8 * kc + 0] = a.x; w[c][8 * kc + 1] = a.y; w[c][8 * kc + 2] = a.z; w[c][8 * kc + 3] = a.w;
            w[c][8 * kc + 4] = b.x; w[c][8 * kc + 5] = b.y; w[c][8 * kc + 6] = b.z; w[c][8 * kc + 7] = b.w;
        }
    }
    const int idx = halving_index<NV>(lane);
    const int m = row0 + idx / CP;
    const int n = cg * CP + idx % CP;
    const bool writer = lane < NV && m < L.M && n < P.N;
    // epilogue operands, requested with everything else
    float bias = 0.f, rr = 0.f, uu = 0.f, nn = 0.f, ghn = 0.f, hin = 0.f;
    if (EPI == 0) {
        bias = (writer && P.bias != nullptr) ? P.bias[n] : 0.f;
        if (writer && P.add != nullptr) bias += P.add[(int64_t)m * P.ldadd + n];
    } else if (writer) {
        const int D = G.D;
        bias = G.g_h[(int64_t)m * G.ldgh + n];
        if (G.dhu_next != nullptr)
            bias += G.next_scale[(int64_t)m * G.ldns] * (G.dhu_next[(int64_t)m * D + n] + G.dhw_next[(int64_t)m * D + n]);
        const float* gt = G.gates + (int64_t)m * G.ldg;
        rr = gt[n]; uu = gt[D + n]; nn = gt[2 * D + n]; ghn = gt[3 * D + n];
        hin = G.h_in[(int64_t)m * G.ldhi + n];
    }

    float acc[NV];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8 * KC; ++i) a = fmaf(x[r][i], w[c][i], a);
            acc[r * CP + c] = a;
        }
    const float tot = halving_sum64<NV>(acc, lane);
    if (!writer) return;
    if (EPI == 0) {
        P.Y[(int64_t)m * P.ldy + n] = tot + bias;
    } else {
        const int D = G.D;
        const float dh = tot + bias;
        const float dn_pre = dh * (1.0f - uu) * (1.0f - nn * nn);
        const float du_pre = dh * (hin - nn) * uu * (1.0f - uu);
        const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
        float* gi = G.dgi + (int64_t)m * G.lddg;
        float* gh = G.dgh + (int64_t)m * G.lddg;
        gi[n] = dr_pre; gi[D + n] = du_pre; gi[2 * D + n] = dn_pre;
        gh[n] = dr_pre; gh[D + n] = du_pre; gh[2 * D + n] = dn_pre * rr;
        G.dhu[(int64_t)m * D + n] = dh * uu;
    }
}

// ---- any K, few rows: wide models (recurrent width up to 4096) at the training / default batch -----------------
// The same work split as k_skinny (wave = RW rows x CP columns, lane = 8 of every 512 k), walked over K in chunks of
// 512 with the next chunk's loads in flight while the current one is multiplied; one wave-level reduction at the end.
// A 16 x 4096 . 4096 x 12288 product is 200 MB of weights against 1.6 GFLOP: memory-bound, so the launch is cut into
// N / CP workgroups (1536 at CP = 8) that each stream CP weight rows.  MODE 0: fp32 operands.  MODE 1 and 2: the bf16
// form of the product (both operands rounded to bf16, fp32 accumulation) with the weights read as fp32 and rounded here
// (1) or read from a bf16 image [N][K] (2: half the bytes).
struct SkinnyLoopProblem {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const uint16_t* Wh;                        // bf16 image [N][K] or nullptr
    const float* bias;
    float* Y; int64_t ldy;
    int N, K;
    int wg_begin;
};
struct SkinnyLoopLaunch {
    SkinnyLoopProblem p[SKINNY_MAX_PROBLEMS];
    int np, M;
};

__device__ __forceinline__ f32x4 round_bf16x4(f32x4 v) {
    return f32x4{(float)(__bf16)v.x, (float)(__bf16)v.y, (float)(__bf16)v.z, (float)(__bf16)v.w};
}

template <int RW, int CP, int MODE>
__global__ __launch_bounds__(256) void k_skinny_loop(const SkinnyLoopLaunch L) {
    constexpr int NV = RW * CP;
    constexpr bool WH = MODE == 2;
    static_assert(NV >= 2 && NV <= 64 && (NV & (NV - 1)) == 0, "RW*CP must be a power of two in [2, 64]");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int pi = L.np - 1;
    while (pi > 0 && (int)blockIdx.x < L.p[pi].wg_begin) --pi;
    const SkinnyLoopProblem& P = L.p[pi];
    const int cg = blockIdx.x - P.wg_begin;
    const int K = P.K;
    const int row0 = blockIdx.y * (4 * RW) + wave * RW;
    const float* xp[RW];
    const float* wp[CP];
    const uint16_t* hp[CP];
#pragma unroll
    for (int r = 0; r < RW; ++r) xp[r] = P.X + (int64_t)min(row0 + r, L.M - 1) * P.ldx;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int64_t n = min(cg * CP + c, P.N - 1);
        wp[c] = P.W + n * P.ldw;
        hp[c] = WH ? P.Wh + n * K : nullptr;
    }
    typedef unsigned int u32x2k __attribute__((ext_vector_type(2)));
    f32x4 xa[2][RW], xb[2][RW], wa[2][CP], wb[2][CP];
    u32x2k ha[2][CP], hb[2][CP];
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // chunk c covers k = 512 c .. 512 c + 511; this lane's k: 4 lane .. + 3 and 256 + 4 lane .. + 3 of it (clamped, masked)
#define SKL_LOAD(S, c)                                                                                 \
    {                                                                                                  \
        const int kA = (c) * 512 + 4 * lane, kB = kA + 256;                                            \
        const int cA = min(kA, K - 4), cB = min(kB, K - 4);                                            \
        _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                               \
            xa[S][r] = *reinterpret_cast<const f32x4*>(xp[r] + cA);                                    \
            xb[S][r] = *reinterpret_cast<const f32x4*>(xp[r] + cB);                                    \
        }                                                                                              \
        _Pragma("unroll") for (int q = 0; q < CP; ++q) {                                               \
            if (WH) {                                                                                  \
                ha[S][q] = *reinterpret_cast<const u32x2k*>(hp[q] + cA);                               \
                hb[S][q] = *reinterpret_cast<const u32x2k*>(hp[q] + cB);                               \
            } else {                                                                                   \
                wa[S][q] = *reinterpret_cast<const f32x4*>(wp[q] + cA);                                \
                wb[S][q] = *reinterpret_cast<const f32x4*>(wp[q] + cB);                                \
            }                                                                                          \
        }                                                                                              \
    }
#define SKL_WIDEN(h) f32x4{__builtin_bit_cast(float, (h).x << 16), __builtin_bit_cast(float, (h).x & 0xffff0000u), \
                           __builtin_bit_cast(float, (h).y << 16), __builtin_bit_cast(float, (h).y & 0xffff0000u)}
#define SKL_FMA(S, c)                                                                                  \
    {                                                                                                  \
        const int kA = (c) * 512 + 4 * lane, kB = kA + 256;                                            \
        const bool vA = kA < K, vB = kB < K;                                                           \
        _Pragma("unroll") for (int q = 0; q < CP; ++q) {                                               \
            f32x4 a = WH ? SKL_WIDEN(ha[S][q]) : wa[S][q];                                             \
            f32x4 b = WH ? SKL_WIDEN(hb[S][q]) : wb[S][q];                                             \
            if (MODE == 1) { a = round_bf16x4(a); b = round_bf16x4(b); }                               \
            if (!vA) a = zero;                                                                         \
            if (!vB) b = zero;                                                                         \
            _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                           \
                float t = acc[r * CP + q];                                                             \
                t = fmaf(xa[S][r].x, a.x, t); t = fmaf(xa[S][r].y, a.y, t);                            \
                t = fmaf(xa[S][r].z, a.z, t); t = fmaf(xa[S][r].w, a.w, t);                            \
                t = fmaf(xb[S][r].x, b.x, t); t = fmaf(xb[S][r].y, b.y, t);                            \
                t = fmaf(xb[S][r].z, b.z, t); t = fmaf(xb[S][r].w, b.w, t);                            \
                acc[r * CP + q] = t;                                                                   \
            }                                                                                          \
        }                                                                                              \
    }
#define SKL_ROUND_X(S)                                                                                 \
    if (MODE != 0) {                                                                                   \
        _Pragma("unroll") for (int r = 0; r < RW; ++r) {                                               \
            xa[S][r] = round_bf16x4(xa[S][r]);                                                         \
            xb[S][r] = round_bf16x4(xb[S][r]);                                                         \
        }                                                                                              \
    }
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const int nc = (K + 511) / 512;
    SKL_LOAD(0, 0)
    for (int c = 0; c < nc; c += 2) {
        SKL_LOAD(1, min(c + 1, nc - 1))
        SKL_ROUND_X(0)
        SKL_FMA(0, c)
        SKL_LOAD(0, min(c + 2, nc - 1))
        SKL_ROUND_X(1)
        if (c + 1 < nc) SKL_FMA(1, c + 1)
    }
#undef SKL_ROUND_X
#undef SKL_LOAD
#undef SKL_WIDEN
#undef SKL_FMA
    const int idx = halving_index<NV>(lane);
    const int m = row0 + idx / CP;
    const int n = cg * CP + idx % CP;
    const float tot = halving_sum64<NV>(acc, lane);
    if (lane < NV && m < L.M && n < P.N) P.Y[(int64_t)m * P.ldy + n] = tot + (P.bias != nullptr ? P.bias[n] : 0.f);
}

}  // namespace drm
