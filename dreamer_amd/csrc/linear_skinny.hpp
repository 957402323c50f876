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
// A 16 x 4096 . 4096 x 12288 product is 200 MB of weights against 1.6 GFLOP: memory-bound; the launch is cut into
// N / 16 workgroups that each stream 16 weight rows.  Weights come as fp32, or (bf16 form of the product: both operands
// rounded to bf16, fp32 accumulation) as fp32 rounded in the kernel, or from a bf16 image [N][K] (half the bytes).
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

// Matrix-core kernel.  Workgroup = 16 CT output columns x 16 RT rows; its four waves split K in
// interleaved blocks and add their partial tiles through LDS at the end.  Each wave moves its blocks of W and X
// through an LDS area of its own: the global loads are laid out for the memory system (a quarter-wave reads 256
// contiguous bytes of one row), the MFMA fragments (16 rows x 4 k-groups per register) are then read back from LDS with
// ds_read_b128; no workgroup barrier inside the K loop.  A weight element is requested from the vector cache once.
// The 16x16 tile of v_mfma_f32_16x16x4_f32 holds the default batch of 16 rows exactly.
//   MODE 0 / 1: fp32 MFMA (1: operands rounded to bf16 first); blocks of 64 k; lane (c = lane & 15, g = lane >> 4) takes
//               k = 16 j + 4 g .. + 3 for j = 0..3: MFMA number 4 j + e multiplies the lanes' element e.
//   MODE 2:     v_mfma_f32_16x16x32_bf16 on a bf16 weight image, X rounded to bf16 on its way into LDS; blocks of 128 k;
//               the lane takes k = 32 j + 8 g .. + 7.
// An LDS row is 256 bytes of payload + 16: the 16 lanes that a ds_read_b128 serves together start 4 banks apart per
// row and cover all 64 banks once.
typedef __bf16 skm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 skm_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int skm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int skm_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_lds_sync() {
    // orders this wave's LDS writes before its later LDS reads by other lanes (LDS operations of one wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int RT, int CT, int MODE>
__global__ __launch_bounds__(256) void k_skinny_mfma(const SkinnyLoopLaunch L) {
    constexpr int NT = RT * CT;
    static_assert(NT <= 4 && CT <= 2, "tile shapes: (1,1) (1,2) (2,1) (2,2) (4,1)");
    constexpr int KB = MODE == 2 ? 128 : 64;          // k per block
    constexpr int KW = MODE == 2 ? 8 : 4;             // k per 16-byte piece of a weight row
    constexpr int LDR = 68;                           // dwords per LDS row
    constexpr int NWL = CT * 4;                       // weight loads per lane and block (4 rows per instruction)
    constexpr int NXL = MODE == 2 ? RT * 8 : RT * 4;  // X loads (MODE 2: 2 rows of 128 fp32 per instruction)
    constexpr int WAVE_DW = (CT + RT) * 16 * LDR;
    static_assert(4 * WAVE_DW >= 4 * NT * 4 * 64, "the partial tiles fit the staging area");
    __shared__ __attribute__((aligned(16))) float smem[4 * WAVE_DW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, kg = lane >> 4;
    float* sw = smem + wave * WAVE_DW;
    float* sx = sw + CT * 16 * LDR;
    int pi = L.np - 1;
    while (pi > 0 && (int)blockIdx.x < L.p[pi].wg_begin) --pi;
    const SkinnyLoopProblem& P = L.p[pi];
    const int n0 = ((int)blockIdx.x - P.wg_begin) * (16 * CT);
    const int K = P.K;
    // loads: lane = (row within the instruction, 16-byte piece of the row's block)
    const float* wrow[NWL];
    const uint16_t* hrow[NWL];
    const float* xrow[NXL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
        const int64_t n = min(n0 + 4 * i + (lane >> 4), P.N - 1);
        wrow[i] = P.W + n * P.ldw;
        hrow[i] = MODE == 2 ? P.Wh + n * K : nullptr;
    }
    const int xq = MODE == 2 ? (lane & 31) : (lane & 15);            // 4-float piece of an X row's block
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int m = MODE == 2 ? 2 * i + (lane >> 5) : 4 * i + (lane >> 4);
        xrow[i] = P.X + (int64_t)min(m, L.M - 1) * P.ldx;
    }
    f32x4 wv[2][NWL], xv[2][NXL];
    skm_u32x4 hv[2][NWL];
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nblk = (K + KB - 1) / KB, last = nblk - 1;
    // workgroups start their walk over K at different blocks, so that at any moment the chip's requests are spread over
    // the whole width of the rows (and with them over the memory channels) instead of one 256-byte column of every row
    const int rot = (int)(blockIdx.x % (unsigned)nblk);
#define SKM_RB(blk) ((blk) < nblk ? ((blk) + rot >= nblk ? (blk) + rot - nblk : (blk) + rot) : last)
#define SKM_LOAD(S, blk)                                                                               \
    {                                                                                                  \
        const int kb = SKM_RB(blk) * KB;                                                               \
        const int ckw = min(kb + KW * (lane & 15), K - KW), ckx = min(kb + 4 * xq, K - 4);             \
        _Pragma("unroll") for (int i = 0; i < NWL; ++i) {                                              \
            if (MODE == 2) hv[S][i] = *reinterpret_cast<const skm_u32x4*>(hrow[i] + ckw);              \
            else wv[S][i] = *reinterpret_cast<const f32x4*>(wrow[i] + ckw);                            \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NXL; ++i) xv[S][i] = *reinterpret_cast<const f32x4*>(xrow[i] + ckx); \
    }
    // registers -> LDS; weights beyond K (or of a block past the end) become zeros, so what X holds there does not count
#define SKM_STAGE(S, blk)                                                                              \
    {                                                                                                  \
        const bool valid = (blk) < nblk && SKM_RB(blk) * KB + KW * (lane & 15) < K;                    \
        _Pragma("unroll") for (int i = 0; i < NWL; ++i) {                                              \
            float* dst = sw + (4 * i + (lane >> 4)) * LDR + 4 * (lane & 15);                           \
            if (MODE == 2) {                                                                           \
                *reinterpret_cast<skm_u32x4*>(dst) = valid ? hv[S][i] : skm_u32x4{0u, 0u, 0u, 0u};     \
            } else {                                                                                   \
                const f32x4 v = MODE == 1 ? round_bf16x4(wv[S][i]) : wv[S][i];                         \
                *reinterpret_cast<f32x4*>(dst) = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};                \
            }                                                                                          \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NXL; ++i) {                                              \
            const f32x4 v = xv[S][i];                                                                  \
            if (MODE == 2) {                                                                           \
                const skm_bf16x4 h = skm_bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};   \
                *reinterpret_cast<skm_u32x2*>(sx + (2 * i + (lane >> 5)) * LDR + 2 * (lane & 31)) = __builtin_bit_cast(skm_u32x2, h); \
            } else {                                                                                   \
                *reinterpret_cast<f32x4*>(sx + (4 * i + (lane >> 4)) * LDR + 4 * (lane & 15)) = MODE == 1 ? round_bf16x4(v) : v; \
            }                                                                                          \
        }                                                                                              \
    }
#define SKM_MMA()                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        f32x4 bw[CT], ax[RT];                                                                          \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) bw[c] = *reinterpret_cast<const f32x4*>(sw + (16 * c + lr) * LDR + 16 * j + 4 * kg); \
        _Pragma("unroll") for (int r = 0; r < RT; ++r) ax[r] = *reinterpret_cast<const f32x4*>(sx + (16 * r + lr) * LDR + 16 * j + 4 * kg); \
        if (MODE == 2) {                                                                               \
            _Pragma("unroll") for (int r = 0; r < RT; ++r)                                             \
                _Pragma("unroll") for (int c = 0; c < CT; ++c)                                         \
                    acc[r * CT + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(skm_bf16x8, ax[r]), \
                                          __builtin_bit_cast(skm_bf16x8, bw[c]), acc[r * CT + c], 0, 0, 0); \
        } else {                                                                                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
                _Pragma("unroll") for (int r = 0; r < RT; ++r)                                         \
                    _Pragma("unroll") for (int c = 0; c < CT; ++c)                                     \
                        acc[r * CT + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[r][e], bw[c][e], acc[r * CT + c], 0, 0, 0); \
        }                                                                                              \
    }
    // this wave's blocks: wave, wave + 4, ...; two blocks of loads in flight while a third is multiplied
    SKM_LOAD(0, wave)
    SKM_LOAD(1, wave + 4)
    for (int b = wave; b < nblk; b += 8) {
        SKM_STAGE(0, b)
        wave_lds_sync();
        SKM_LOAD(0, b + 8)
        SKM_MMA()
        wave_lds_sync();
        SKM_STAGE(1, b + 4)
        wave_lds_sync();
        SKM_LOAD(1, b + 12)
        SKM_MMA()
        wave_lds_sync();
    }
#undef SKM_LOAD
#undef SKM_RB
#undef SKM_STAGE
#undef SKM_MMA
    __syncthreads();
    float (*red)[NT * 4][64] = reinterpret_cast<float (*)[NT * 4][64]>(smem);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][t * 4 + r][lane] = acc[t][r];
    __syncthreads();
    // accumulator element r of lane l of tile (rt, ct): row 16 rt + 4 (l >> 4) + r, column 16 ct + (l & 15)
#pragma unroll
    for (int v = 0; v < NT; ++v) {
        const int idx = v * 256 + tid;
        const int e = idx >> 6, l = idx & 63;
        const int t = e >> 2, r = e & 3;
        const int m = (t / CT) * 16 + 4 * (l >> 4) + r;
        const int n = n0 + (t % CT) * 16 + (l & 15);
        const float tot = (red[0][e][l] + red[1][e][l]) + (red[2][e][l] + red[3][e][l]);
        if (m < L.M && n < P.N) P.Y[(int64_t)m * P.ldy + n] = tot + (P.bias != nullptr ? P.bias[n] : 0.f);
    }
}

}  // namespace drm
