// Large-row linear block on fp32 MFMA (v_mfma_f32_32x32x2_f32): the MFMA-bound regime
// (thousands of trajectories).
//
// Workgroup = 4 waves = a 64-row x 128-column output tile; waves are laid out 2 (rows) x 2 (columns),
// each owning a 32 x 64 sub-tile = two 32x32 accumulators.  K is walked in steps of 32 through two LDS
// buffers with two register stages of global loads in flight (one barrier per step; schedule at k_linear_big).
//
// LDS image: [row][k] with a row stride of 36 words for both operands (X rows and W rows are both
// K-contiguous in memory, so the same 128-byte-per-row coalesced load shape fills both).  A lane reads
// its operand as one 16-byte ds_read_b128 = four consecutive k; those four feed four MFMAs whose
// instruction-level k index is lane>>5, i.e. the products are summed over k = kb + 4*(lane>>5) + s --
// the same permutation of k for A and B, so the sum is unchanged.  With the 36-word stride the 16
// lanes a ds_read_b128 services together touch 16 different 4-bank groups: no bank conflicts.
//
// Epilogue: + bias, + one-hot table rows (the z part of a Linear over cat[h, z]), store.
#pragma once
#include "common.hpp"

namespace drm {

struct BigProblem {
    const float* X; int64_t ldx;     // (M, K)
    const float* W; int64_t ldw;     // (N, K) torch Linear layout
    const float* bias;               // (N) or nullptr
    const float* tab; int64_t ldt;   // one-hot gather table [(l*classes + idx)][n] or nullptr
    float* Y; int64_t ldy;
    int N, K;
    int tile_begin;                  // first blockIdx.x (column tile) of this problem
    // fp32 form only.  LayerNorm + SiLU of the X rows on their way into LDS (the consumer normalises), with the row
    // statistics taken from what the producer of X left behind: stats_in[row][part] = (sum, sum of squared deviations
    // from the slice's own mean) over a slice of the row, parts_in <= 16 slices of pw_in columns; the consumer merges the
    // slices with the pairwise update of Chan et al. (mean from the sums, M2 = sum of the slices' M2 plus
    // n_p (mean_p - mean)^2), so a row whose mean is large against its spread loses no digits to a subtraction of two
    // large numbers.  stats_out: the same for the Y rows this problem writes (slices of 32 NT columns when stored from
    // the accumulators, of 64 NT columns through LDS).
    const float* ln_g = nullptr; const float* ln_b = nullptr;
    const float* stats_in = nullptr; int parts_in = 0;
    float* stats_out = nullptr; int parts_out = 0;
    int pw_in = 0;                   // columns per slice of stats_in (the last slice of a row may be shorter)
    const uint16_t* Ws = nullptr;    // W as bf16 images [planes][N][K] made by k_split_planes (split form: 3 planes; bf16 form: 1), or nullptr
};

struct BigLaunch {
    BigProblem p[4];
    int np, M;
    const uint8_t* zidx; int64_t ldz; int latents, classes;
    int ntx = 0, nty = 0;            // fp32 form: column tiles (all problems) and row tiles; its grid is one-dimensional
};

constexpr int BIG_BM = 64, BIG_BN = 128, BIG_BN64 = 64, BIG_BK = 32, BIG_LD = 36;

// Sum over the 64 lanes of a wave of each of 32 per-lane values with a butterfly in which the number of live values
// halves per level (31 exchanges, then one across the wave halves); lane l ends up with the total of value number
// bit-reverse5(l & 31).
__device__ __forceinline__ float halving_sum32_big(float (&v)[32], int lane) {
#define DRM_HALVE_BIG(j)                                                    \
    {                                                                       \
        constexpr int n = 32 >> (j);                                        \
        const bool up = (lane & (1 << (j))) != 0;                           \
        _Pragma("unroll") for (int i = 0; i < n / 2; ++i) {                 \
            const float keep = up ? v[i + n / 2] : v[i];                    \
            const float send = up ? v[i] : v[i + n / 2];                    \
            v[i] = keep + xor_lane<(1 << (j))>(send);                       \
        }                                                                   \
    }
    DRM_HALVE_BIG(0) DRM_HALVE_BIG(1) DRM_HALVE_BIG(2) DRM_HALVE_BIG(3) DRM_HALVE_BIG(4)
#undef DRM_HALVE_BIG
    return v[0] + xor_lane<32>(v[0]);
}

// x * sigmoid(x) from the hardware's exp2 and reciprocal (each within 1 ulp): 6 instructions where the IEEE division
// sequence takes about 30; used where the activation runs between matrix instructions
__device__ __forceinline__ float big_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// Epilogue shared by the large-row kernels: the 64 x BN output tile lies in LDS (row stride BN + 4 words), every thread
// owns 4 consecutive columns of a row: + bias, + one-hot table rows (16-byte reads), 16-byte coalesced stores.
template <int BN>
__device__ __forceinline__ void big_epilogue(const BigLaunch& L, const BigProblem& P, const float (*sC)[BN + 4], int m0,
                                             int n0, int tid) {
    constexpr int TPR = BN / 4;                  // threads per output row
    const int c4 = (tid % TPR) * 4;
    const int n = n0 + c4;
    const bool live = n < P.N;                   // (no early return: the row sums below are taken over whole lane groups)
    const bool vec = (n + 3 < P.N) && ((P.ldy & 3) == 0) && ((P.ldt & 3) == 0);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias && live) {
        bias.x = P.bias[n];
        if (n + 1 < P.N) bias.y = P.bias[n + 1];
        if (n + 2 < P.N) bias.z = P.bias[n + 2];
        if (n + 3 < P.N) bias.w = P.bias[n + 3];
    }
    for (int r = tid / TPR; r < BIG_BM; r += 256 / TPR) {
        const int m = m0 + r;
        if (m >= L.M) break;
        float4 y = *reinterpret_cast<const float4*>(&sC[r][c4]);
        y.x += bias.x; y.y += bias.y; y.z += bias.z; y.w += bias.w;
        if (P.tab != nullptr && live) {
            const uint8_t* zi = L.zidx + (int64_t)m * L.ldz;
            if (vec) {
                for (int l = 0; l < L.latents; l += 8) {
                    float4 u[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ll = min(l + j, L.latents - 1);
                        u[j] = *reinterpret_cast<const float4*>(P.tab + (int64_t)(ll * L.classes + zi[ll]) * P.ldt + n);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (l + j < L.latents) { y.x += u[j].x; y.y += u[j].y; y.z += u[j].z; y.w += u[j].w; }
                }
            } else {
                for (int l = 0; l < L.latents; ++l) {
                    const float* tb = P.tab + (int64_t)(l * L.classes + zi[l]) * P.ldt + n;
                    y.x += tb[0];
                    if (n + 1 < P.N) y.y += tb[1];
                    if (n + 2 < P.N) y.z += tb[2];
                    if (n + 3 < P.N) y.w += tb[3];
                }
            }
        }
        if (live) {
            float* out = P.Y + (int64_t)m * P.ldy + n;
            if (vec) {
                *reinterpret_cast<float4*>(out) = y;
            } else {
                out[0] = y.x;
                if (n + 1 < P.N) out[1] = y.y;
                if (n + 2 < P.N) out[2] = y.z;
                if (n + 3 < P.N) out[3] = y.w;
            }
        }
        if (P.stats_out != nullptr) {       // sum and centred sum of squares over this tile's columns (uniform branch)
            const bool l1 = n + 1 < P.N, l2 = n + 2 < P.N, l3 = n + 3 < P.N;
            const float v0 = live ? y.x : 0.f, v1 = l1 ? y.y : 0.f, v2 = l2 ? y.z : 0.f, v3 = l3 ? y.w : 0.f;
            const float sm = group_sum<TPR>((v0 + v1) + (v2 + v3));
            const float mp = sm / (float)min(BN, P.N - n0);
            const float d0 = live ? y.x - mp : 0.f, d1 = l1 ? y.y - mp : 0.f, d2 = l2 ? y.z - mp : 0.f, d3 = l3 ? y.w - mp : 0.f;
            const float sq = group_sum<TPR>((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            if (tid % TPR == 0)
                *reinterpret_cast<float2*>(P.stats_out + ((int64_t)m * P.parts_out + n0 / BN) * 2) = make_float2(sm, sq);
        }
    }
}

// fp32 form.  NT = 32-column accumulators per wave: 2 -> 64 x 128 tile; 1 -> 64 x 64 tile, for launches whose
// 128-column grid would leave one workgroup per CU (e.g. 4096 rows x 512 columns = 256 tiles).
//
// Schedule of one K-step (32 deep = 4 fragment groups g0..g3 of 8 k, 4 NT MFMAs each) from LDS buffer b, with the
// next step's operands waiting in registers (loaded two steps ago) and the step after that in flight:
//
//     read g2 | MFMA g0 | read g3 | MFMA g1 | registers -> buffer b^1 | MFMA g2 | barrier |
//     read g0, g1 of buffer b^1 | global loads of step + 3 | MFMA g3
//
// so the LDS stores complete under MFMA g2, the barrier finds every wave's LDS traffic finished, and the first
// fragment reads of the next step and the global-load issue run under MFMA g3: the matrix pipe has work queued across
// the step boundary.
//
// A workgroup walks a sequence of tiles (grid = min(tiles, resident workgroups) when every K is an even number of
// steps, else one tile each): the K-steps of consecutive tiles form one stream, the loads of "step + 3" past the end
// of a tile fetch the first steps of the next one, and the finished tile is stored straight from the accumulators
// (nothing waits for the stores) while the next tile's first fragments are already in registers: no prologue and no
// drain between tiles.  Past the last tile the loads re-read its last step (no branch in the loop: the load counter
// the compiler waits on stays exact, only the loads a store needs are waited for).  Problems with a one-hot table
// take the epilogue through LDS and are launched one tile per workgroup.
// LNT: some problem of the launch normalises its X rows on load or leaves row statistics behind (the code for both
// costs ~50 registers).
template <int NT, bool LNT>
__global__ __launch_bounds__(256) void k_linear_big(const BigLaunch L) {
    constexpr int BN = 64 * NT, NB = 2 * NT;
    constexpr int STAGE_WORDS = 2 * (BIG_BM + BN) * BIG_LD, TILE_WORDS = BIG_BM * (BN + 4);
    // one array: operand stages during the K loop, the output tile during the epilogue through LDS
    __shared__ __attribute__((aligned(16))) float smem[STAGE_WORDS > TILE_WORDS ? STAGE_WORDS : TILE_WORDS];
    float (*sA)[BIG_BM][BIG_LD] = reinterpret_cast<float (*)[BIG_BM][BIG_LD]>(smem);
    float (*sB)[BN][BIG_LD] = reinterpret_cast<float (*)[BN][BIG_LD]>(smem + 2 * BIG_BM * BIG_LD);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1;            // wave position in the 2 x 2 grid
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid >> 3, skq = (tid & 7) * 4;     // global -> register staging: 8 threads cover one 128-byte row segment

    // Tile sequence of this workgroup: tile = tbase + tj, tj += tstride while tj < tlimit.  Workgroups are dealt
    // round-robin over the 8 XCDs; each XCD gets one contiguous eighth of the tiles (column tiles of a row tile are
    // consecutive, so they meet in one L2; a speed choice only).
    const int ntiles = L.ntx * L.nty, G = gridDim.x, w = blockIdx.x;
    const bool by_xcd = ((ntiles & 7) == 0) && ((G & 7) == 0);
    const int tbase = by_xcd ? (w & 7) * (ntiles >> 3) : 0;
    const int tstride = by_xcd ? (G >> 3) : G, tlimit = by_xcd ? (ntiles >> 3) : ntiles;
    int tj = by_xcd ? (w >> 3) : w;
    if (tj >= tlimit) return;

    // per-tile state: problem, origin, operand pointers of the thread's staging rows
    int pi, m0, n0, nk;
    const float* xa[2];
    const float* wb[NB];
    int pin, m0n, n0n, nkn;                               // the same for the next tile of the sequence
    const float* xan[2];
    const float* wbn[NB];
    // LayerNorm + SiLU of the X rows on load: mean / reciprocal deviation of the thread's two staging rows, scale and
    // shift vectors at the thread's k offset (a valid dummy address when the problem has none)
    bool ln, lnn;
    float mu[2], rs[2], mun[2], rsn[2];
    const float* gp; const float* bp; const float* gpn; const float* bpn;
#define BIG_TILE(tile, PI, M0, N0, NK, XA, WB, LN, MU, RS, GP, BP)                                    \
    {                                                                                                 \
        const int bx_ = (tile) % L.ntx, by_ = (tile) / L.ntx;                                         \
        PI = 0;                                                                                       \
        _Pragma("unroll") for (int q = 1; q < 4; ++q) PI += (q < L.np && bx_ >= L.p[q].tile_begin) ? 1 : 0;  \
        const BigProblem& Q = L.p[PI];                                                                \
        N0 = (bx_ - Q.tile_begin) * BN; M0 = by_ * BIG_BM; NK = Q.K / BIG_BK;                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) XA[i] = Q.X + (int64_t)min(M0 + srow + 32 * i, L.M - 1) * Q.ldx + skq;  \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) WB[i] = Q.W + (int64_t)min(N0 + srow + 32 * i, Q.N - 1) * Q.ldw + skq; \
        LN = LNT && Q.ln_g != nullptr;                                                                \
        GP = LN ? Q.ln_g + skq : XA[0]; BP = LN ? Q.ln_b + skq : XA[0];                               \
        /* row statistics: the 8 threads of a staging row fetch two parts each (no branch around the loads) */ \
        const float* st_ = LN ? Q.stats_in : Q.X;                                                     \
        const int np_ = LN ? Q.parts_in : 1, pw_ = LN ? Q.pw_in : 1;                                  \
        if (LNT) _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                      \
            const int64_t row_ = LN ? (int64_t)min(M0 + srow + 32 * i, L.M - 1) : 0;                  \
            const int p0_ = (tid & 7) * 2, p1_ = p0_ + 1;                                             \
            const float2 a_ = *reinterpret_cast<const float2*>(st_ + (row_ * np_ + min(p0_, np_ - 1)) * 2);  \
            const float2 b_ = *reinterpret_cast<const float2*>(st_ + (row_ * np_ + min(p1_, np_ - 1)) * 2);  \
            const float sm_ = group_sum<8>((p0_ < np_ ? a_.x : 0.f) + (p1_ < np_ ? b_.x : 0.f));      \
            const float mean_ = sm_ / (float)Q.K;                                                     \
            /* slices merged as (count, mean, M2) triples: M2 = sum M2_p + n_p (mean_p - mean)^2 */   \
            const float n0_ = (float)min(pw_, Q.K - p0_ * pw_), n1_ = (float)min(pw_, Q.K - p1_ * pw_);  \
            const float d0_ = a_.x / n0_ - mean_, d1_ = b_.x / n1_ - mean_;                           \
            const float m2_ = group_sum<8>((p0_ < np_ ? a_.y + n0_ * d0_ * d0_ : 0.f) + (p1_ < np_ ? b_.y + n1_ * d1_ * d1_ : 0.f)); \
            MU[i] = mean_; RS[i] = 1.0f / sqrtf(fmaxf(m2_ / (float)Q.K, 0.f) + LN_EPS);               \
        }                                                                                             \
    }
    BIG_TILE(tbase + tj, pi, m0, n0, nk, xa, wb, ln, mu, rs, gp, bp)

    f32x4 ra0[2], ra1[2], rb0[NB], rb1[NB];            // register stages 0 / 1
    f32x4 rg0, rg1, rs0, rs1;                          // LayerNorm scale / shift at the stage's k offset
    // operands of K-step `step` of the current tile, or of step - nk of the next tile once step >= nk
#define BIG_LOAD(S, step)                                                                             \
    {                                                                                                 \
        const bool cur_ = (step) < nk;                                                                \
        const int ko = (cur_ ? (step) : min((step) - nk, nkn - 1)) * BIG_BK;                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) ra##S[i] = *reinterpret_cast<const f32x4*>((cur_ ? xa[i] : xan[i]) + ko);  \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) rb##S[i] = *reinterpret_cast<const f32x4*>((cur_ ? wb[i] : wbn[i]) + ko); \
        if (LNT) {                                                                                    \
            rg##S = *reinterpret_cast<const f32x4*>((cur_ ? gp : gpn) + ko);                          \
            rs##S = *reinterpret_cast<const f32x4*>((cur_ ? bp : bpn) + ko);                          \
        }                                                                                             \
    }
    // registers -> LDS; `step` is the K-step the stage holds (>= nk: it belongs to the next tile)
#define BIG_STORE(S, buf, step)                                                                       \
    {                                                                                                 \
        const bool nx_ = (step) >= nk;                                                                \
        if (LNT && (nx_ ? lnn : ln)) {                                                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                           \
                const float m_ = nx_ ? mun[i] : mu[i], r_ = nx_ ? rsn[i] : rs[i];                     \
                ra##S[i].x = big_silu((ra##S[i].x - m_) * r_ * rg##S.x + rs##S.x);                    \
                ra##S[i].y = big_silu((ra##S[i].y - m_) * r_ * rg##S.y + rs##S.y);                    \
                ra##S[i].z = big_silu((ra##S[i].z - m_) * r_ * rg##S.z + rs##S.z);                    \
                ra##S[i].w = big_silu((ra##S[i].w - m_) * r_ * rg##S.w + rs##S.w);                    \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&sA[buf][srow + 32 * i][skq]) = ra##S[i];  \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&sB[buf][srow + 32 * i][skq]) = rb##S[i]; \
    }
#define BIG_FRAG(buf, g, F)                                                                           \
    {                                                                                                 \
        F##_a = *reinterpret_cast<const f32x4*>(&sA[buf][wm * 32 + li][(g) * 8 + lh * 4]);            \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                \
            F##_b[j] = *reinterpret_cast<const f32x4*>(&sB[buf][wn * 32 * NT + 32 * j + li][(g) * 8 + lh * 4]); \
    }
#define BIG_MMA1(F, c)                                                                                \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_a.c, F##_b[j].c, acc[j], 0, 0, 0);
#define BIG_MMA(F) BIG_MMA1(F, x) BIG_MMA1(F, y) BIG_MMA1(F, z) BIG_MMA1(F, w)
#define BIG_FENCE __builtin_amdgcn_sched_barrier(0);   /* the instruction scheduler moves nothing across this point */
#define BIG_HALF(buf, S, step)                                                                        \
    {                                                                                                 \
        BIG_FRAG(buf, 2, fc)                                                                          \
        BIG_MMA(fa)                                                                                   \
        BIG_FENCE                                                                                     \
        BIG_FRAG(buf, 3, fd)                                                                          \
        BIG_MMA(fb)                                                                                   \
        BIG_FENCE                                                                                     \
        BIG_STORE(S, (buf) ^ 1, (step) + 1)                                                           \
        BIG_FENCE                                                                                     \
        BIG_MMA(fc)                                                                                   \
        BIG_FENCE                                                                                     \
        __syncthreads();                                                                              \
        BIG_FENCE                                                                                     \
        BIG_FRAG((buf) ^ 1, 0, fa)                                                                    \
        BIG_FRAG((buf) ^ 1, 1, fb)                                                                    \
        BIG_FENCE                                                                                     \
        BIG_LOAD(S, (step) + 3)                                                                       \
        BIG_FENCE                                                                                     \
        BIG_MMA(fd)                                                                                   \
        BIG_FENCE                                                                                     \
    }

    f32x16 acc[NT];
    f32x4 fa_a, fb_a, fc_a, fd_a, fa_b[NT], fb_b[NT], fc_b[NT], fd_b[NT];
    STAMP_DECL_AT(blockIdx.x < 48)
    STAMP();
    // the first tile's prologue (the "next tile" of these loads is never reached: nk >= 1)
    pin = pi; m0n = m0; n0n = n0; nkn = nk; lnn = ln; gpn = gp; bpn = bp;
#pragma unroll
    for (int i = 0; i < 2; ++i) { xan[i] = xa[i]; mun[i] = mu[i]; rsn[i] = rs[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) wbn[i] = wb[i];
    BIG_LOAD(0, 0)
    BIG_LOAD(1, 1)
    BIG_STORE(0, 0, 0)
    __syncthreads();
    STAMP();
    BIG_LOAD(0, 2)
    BIG_FRAG(0, 0, fa)
    BIG_FRAG(0, 1, fb)

    for (;;) {
        const BigProblem& P = L.p[pi];
        const bool more = tj + tstride < tlimit;
        // (no branches in the tile set-up: a branch makes the compiler wait for every load in flight)
        BIG_TILE(tbase + (more ? tj + tstride : tj), pin, m0n, n0n, nkn, xan, wbn, lnn, mun, rsn, gpn, bpn)
        // bias of the lane's output columns, fetched before the K loop
        float bias_lane[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // no branch around the load (a branch here makes the compiler wait for every load in flight)
            const uintptr_t none = P.bias != nullptr ? (uintptr_t)0 : ~(uintptr_t)0;
            const float* bp = reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(P.bias) | (reinterpret_cast<uintptr_t>(P.W) & none));
            const float b = bp[min(n0 + wn * 32 * NT + 32 * j + li, P.N - 1)];
            bias_lane[j] = none ? 0.f : b;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

        int ks = 0;
        for (; ks + 1 < nk; ks += 2) {
            BIG_HALF(0, 1, ks)
            BIG_HALF(1, 0, ks + 1)
        }
        if (ks < nk) BIG_HALF(0, 1, ks)     // odd step count: only ever the last tile of a workgroup (host rule)

        // ---- epilogue.  C layout of 32x32x2: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).
        STAMP();
        if (P.tab == nullptr) {
            // straight from the accumulators: a wave instruction stores two rows x 32 consecutive columns (two whole
            // 128-byte lines when the output rows are line-aligned)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * 32 * NT + 32 * j + li;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] += bias_lane[j];
                if (n < P.N) {
                    float* out = P.Y + (int64_t)(m0 + wm * 32 + 4 * lh) * P.ldy + n;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int r = (i & 3) + 8 * (i >> 2);
                        if (m0 + wm * 32 + 4 * lh + r < L.M) out[(int64_t)r * P.ldy] = acc[j][i];
                    }
                }
            }
            if (LNT && P.stats_out != nullptr) {
                // per row, over the wave's 32 NT columns: the sum, then the sum of squared deviations from that slice
                // mean.  16 rows per lane half; per quantity one 32-value halving butterfly over the wave (each half
                // contributes zeros in the other half's slots), after which lane l holds the total of value number
                // bit-reverse5(l & 31); the slice means travel back to the lanes that hold the rows' entries by ds_bpermute
                int idx = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) idx |= ((lane >> j) & 1) << (4 - j);
                const int half = idx >> 4, irow = idx & 15;
                const int m = m0 + wm * 32 + (irow & 3) + 8 * (irow >> 2) + 4 * half;
                const int part = (n0 + wn * 32 * NT) / (32 * NT);
                const int cnt = min(32 * NT, P.N - (n0 + wn * 32 * NT));       // columns of this slice inside the matrix
                bool colv[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) colv[j] = (n0 + wn * 32 * NT + 32 * j + li) < P.N;
                float v[32];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float t = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) t += colv[j] ? acc[j][i] : 0.f;
                    v[i] = lh == 0 ? t : 0.f;
                    v[16 + i] = lh == 1 ? t : 0.f;
                }
                const float tot_sum = halving_sum32_big(v, lane);
                const float mean_l = tot_sum / (float)max(cnt, 1);            // of value number bit-reverse5(lane & 31)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // value number lh * 16 + i lives in lane bit-reverse5(lh * 16 + i) = (bit-reverse4(i) << 1) | lh
                    const int br4 = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
                    const float mean_i = __shfl(mean_l, (br4 << 1) | lh, 64);
                    float t = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float d = colv[j] ? acc[j][i] - mean_i : 0.f;
                        t += d * d;
                    }
                    v[i] = lh == 0 ? t : 0.f;
                    v[16 + i] = lh == 1 ? t : 0.f;
                }
                const float tot_m2 = halving_sum32_big(v, lane);
                if (lane < 32 && m < L.M && part < P.parts_out)
                    *reinterpret_cast<float2*>(P.stats_out + ((int64_t)m * P.parts_out + part) * 2) = make_float2(tot_sum, tot_m2);
            }
        } else {
            // through LDS (16-byte table-row reads and stores); the operand stages are overwritten, so such a tile is
            // the last of its workgroup
            __syncthreads();
            float (*sC)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(smem);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) sC[wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh][wn * 32 * NT + 32 * j + li] = acc[j][i];
            __syncthreads();
            big_epilogue<BN>(L, P, sC, m0, n0, tid);
            break;
        }
        STAMP();
        if (!more) break;
        tj += tstride;
        pi = pin; m0 = m0n; n0 = n0n; nk = nkn; ln = lnn; gp = gpn; bp = bpn;
#pragma unroll
        for (int i = 0; i < 2; ++i) { xa[i] = xan[i]; mu[i] = mun[i]; rs[i] = rsn[i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) wb[i] = wbn[i];
    }
    STAMP_FLUSH((unsigned long long)(7000 + NT));
#undef BIG_TILE
#undef BIG_LOAD
#undef BIG_STORE
#undef BIG_FRAG
#undef BIG_MMA1
#undef BIG_MMA
#undef BIG_HALF
#undef BIG_FENCE
}

// ---- bf16 matrix inputs, fp32 accumulation (v_mfma_f32_32x32x16_bf16) ------------------------------
// Same tiling and epilogue as the fp32 form.  The activation operand stays fp32 in memory and is rounded to bf16
// (round to nearest even) on its way into LDS; the weight operand is either treated the same way (WH = false) or
// read from a bf16 image [N][K] made once per call by k_split_planes (WH = true: half the weight bytes per step,
// which is what bounds wide models at few rows).  A lane's fragment is 8 consecutive k = one 16-byte LDS read, and
// a K-step of 32 is two MFMAs per 32x32 accumulator instead of sixteen.  LDS image: bf16 [row][k], row stride 40
// elements (80 bytes): the 16 lanes a ds_read_b128 serves together start 20 banks apart and cover all 64 banks once.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int BIG_LDH = 40;

// MT = 2: 128 x 128 tile, every wave two row tiles against the same weight fragments (8 LDS fragment reads per 8 MFMAs
// instead of 6 per 4), for launches with enough tiles to fill the CUs that way.
template <bool WH, int MT = 1>
__global__ __launch_bounds__(256) void k_linear_big_bf16(const BigLaunch L) {
    constexpr int BM = 64 * MT;
    constexpr int STAGE_WORDS = 2 * (BM + BIG_BN) * BIG_LDH / 2, TILE_WORDS = BIG_BM * (BIG_BN + 4);   // bf16 stages | 64-row epilogue tile
    __shared__ __attribute__((aligned(16))) float smem[STAGE_WORDS > TILE_WORDS ? STAGE_WORDS : TILE_WORDS];
    __bf16 (*sA)[BM][BIG_LDH] = reinterpret_cast<__bf16 (*)[BM][BIG_LDH]>(smem);
    __bf16 (*sB)[BIG_BN][BIG_LDH] = reinterpret_cast<__bf16 (*)[BIG_BN][BIG_LDH]>(reinterpret_cast<__bf16*>(smem) + 2 * BM * BIG_LDH);

    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    int lin = blockIdx.y * nbx + blockIdx.x;
    if ((nblk & 7) == 0) lin = (lin & 7) * (nblk >> 3) + (lin >> 3);
    const int bx = lin % nbx, by = lin / nbx;

    int pi = L.np - 1;
    while (pi > 0 && bx < L.p[pi].tile_begin) --pi;
    const BigProblem& P = L.p[pi];
    const int n0 = (bx - P.tile_begin) * BIG_BN;
    const int m0 = by * BM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int K = P.K;

    const int srow = tid >> 3, skq = (tid & 7) * 4;
    const float* xa[2 * MT];
    const float* wb[4];
    const uint16_t* wh[4];
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) xa[i] = P.X + (int64_t)min(m0 + srow + 32 * i, L.M - 1) * P.ldx + skq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = min(n0 + srow + 32 * i, P.N - 1);
        wb[i] = P.W + row * P.ldw + skq;
        wh[i] = WH ? P.Ws + row * K + skq : nullptr;
    }

    const int nk = K / BIG_BK;
    f32x4 ra0[2 * MT], ra1[2 * MT], rb0[4], rb1[4];
    u32x2 rh0[4], rh1[4];
#define BIGH_LOAD(S, ko)                                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i) ra##S[i] = *reinterpret_cast<const f32x4*>(xa[i] + (ko));  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
            if (WH) rh##S[i] = *reinterpret_cast<const u32x2*>(wh[i] + (ko));                         \
            else rb##S[i] = *reinterpret_cast<const f32x4*>(wb[i] + (ko));                            \
        }                                                                                             \
    }
#define BIGH_PUT(dst, v)                                                    \
    {                                                                       \
        bf16x4 h;                                                           \
        h[0] = (__bf16)(v).x; h[1] = (__bf16)(v).y; h[2] = (__bf16)(v).z; h[3] = (__bf16)(v).w; \
        *reinterpret_cast<bf16x4*>(dst) = h;                                \
    }
#define BIGH_STORE(S, buf)                                                                            \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i) BIGH_PUT(&sA[buf][srow + 32 * i][skq], ra##S[i])  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
            if (WH) *reinterpret_cast<u32x2*>(&sB[buf][srow + 32 * i][skq]) = rh##S[i];               \
            else BIGH_PUT(&sB[buf][srow + 32 * i][skq], rb##S[i])                                     \
        }                                                                                             \
    }
#define BIGH_STEP(buf)                                                                                  \
    {                                                                                                   \
        const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(&sA[buf][wm * 32 + li][lh * 8]);            \
        const bf16x8 fb00 = *reinterpret_cast<const bf16x8*>(&sB[buf][wn * 64 + li][lh * 8]);           \
        const bf16x8 fb10 = *reinterpret_cast<const bf16x8*>(&sB[buf][wn * 64 + 32 + li][lh * 8]);      \
        const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(&sA[buf][wm * 32 + li][16 + lh * 8]);       \
        const bf16x8 fb01 = *reinterpret_cast<const bf16x8*>(&sB[buf][wn * 64 + li][16 + lh * 8]);      \
        const bf16x8 fb11 = *reinterpret_cast<const bf16x8*>(&sB[buf][wn * 64 + 32 + li][16 + lh * 8]); \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb00, acc0, 0, 0, 0);                       \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb10, acc1, 0, 0, 0);                       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb01, acc0, 0, 0, 0);                       \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb11, acc1, 0, 0, 0);                       \
        if (MT == 2) {                      /* the wave's second row tile: rows 64 + 32 wm ... */        \
            const bf16x8 fc0 = *reinterpret_cast<const bf16x8*>(&sA[buf][64 + wm * 32 + li][lh * 8]);       \
            const bf16x8 fc1 = *reinterpret_cast<const bf16x8*>(&sA[buf][64 + wm * 32 + li][16 + lh * 8]);  \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc0, fb00, acc2, 0, 0, 0);                   \
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc0, fb10, acc3, 0, 0, 0);                   \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc1, fb01, acc2, 0, 0, 0);                   \
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc1, fb11, acc3, 0, 0, 0);                   \
        }                                                                                               \
    }
    // loads past the last step re-read it (no branch around the loads)
    BIGH_LOAD(0, 0)
    BIGH_LOAD(1, (nk > 1 ? 1 : 0) * BIG_BK)

    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }

    for (int ks = 0; ks < nk; ks += 2) {
        BIGH_STORE(0, 0)
        __syncthreads();
        BIGH_LOAD(0, min(ks + 2, nk - 1) * BIG_BK)
        BIGH_STEP(0)
        if (ks + 1 < nk) {
            BIGH_STORE(1, 1)
            __syncthreads();
            BIGH_LOAD(1, min(ks + 3, nk - 1) * BIG_BK)
            BIGH_STEP(1)
        }
    }
#undef BIGH_LOAD
#undef BIGH_PUT
#undef BIGH_STORE
#undef BIGH_STEP

    // ---- epilogue (as the fp32 form's path through LDS): + bias, + one-hot table rows, 16-byte stores
    float (*sC)[BIG_BN + 4] = reinterpret_cast<float (*)[BIG_BN + 4]>(smem);
#pragma unroll
    for (int pass = 0; pass < MT; ++pass) {      // 64 rows at a time: rows 64 pass + 32 wm ... of every wave
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sC[wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh][wn * 64 + li] = pass == 0 ? acc0[i] : acc2[i];
            sC[wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh][wn * 64 + 32 + li] = pass == 0 ? acc1[i] : acc3[i];
        }
        __syncthreads();
        big_epilogue<BIG_BN>(L, P, sC, m0 + 64 * pass, n0, tid);
    }
}

// ---- fp32 products on the bf16 matrix pipe: exact three-way split -----------------------------------
// Every fp32 operand is split exactly into three bf16 parts, a = h + m + l (h = bf16(a), m = bf16(a - h),
// l = bf16(a - h - m): 8 + 8 + 8 mantissa bits, both residuals exact), on its way into LDS.  The product
// a*b is accumulated from the six partial products hh' + hm' + mh' + hl' + lh' + mm' (each exact in fp32);
// the three dropped ones are below 2^-24 |a b|, i.e. below the rounding of an fp32 product.  Accumulation
// is fp32 as in k_linear_big, so results agree with the fp32 form to fp32 rounding, while the matrix pipe
// runs v_mfma_f32_32x32x16_bf16: 6 of them cover the k range of 48 v_mfma_f32_32x32x2_f32 at an eighth of
// the passes each, a third of the matrix-pipe time of the fp32 form.
// Same 64 x 128 tile, wave layout and epilogue as k_linear_big_bf16; one LDS stage (three bf16 images per
// operand, 46 KB), the next K-step's global loads in flight in registers during the MFMAs.
// PRESPLIT: the weight operand comes from three bf16 images made once per call by k_split_planes (weights are
// shared by every row tile and every step of a rollout), so only the activation operand is split here.
// NT = 32-column accumulators per wave: 2 -> 64 x 128 tile, 1 -> 64 x 64 tile (launches whose 128-column grid would
// put a single workgroup on a CU).
// MT = 2: 128 x 128 tile, every wave two row tiles (rows 32 wm ... and 64 + 32 wm ...) against the same weight fragments:
// 12 LDS fragment reads per 24 MFMAs instead of 9 per 12 -- for launches with enough tiles to fill the CUs that way.
// LNT: the consumer normalises, as in k_linear_big<., true> -- LayerNorm + SiLU of the X rows between the global load and
// the split, from the row statistics the producer left behind (a problem without ln_g in the same launch loads as before).
// four values -> their three bf16 parts (h = bf16(a), m = bf16(a - h), l = bf16(a - h - m)), pairwise: one
// v_cvt_pk_bf16_f32 per pair and part, the parts widened back by a shift / a mask, residuals as packed fp32 subtractions
typedef float big_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 big_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int big_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned big_pack2(big_f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, big_bf16x2)); }
__device__ __forceinline__ big_f32x2 big_widen2(unsigned p) {
    return big_f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}
__device__ __forceinline__ void big_split_put(__bf16* dh, __bf16* dm, __bf16* dl, float f0, float f1, float f2, float f3) {
    const big_f32x2 x0{f0, f1}, x1{f2, f3};
    const unsigned h0 = big_pack2(x0), h1 = big_pack2(x1);
    const big_f32x2 r0 = x0 - big_widen2(h0), r1 = x1 - big_widen2(h1);
    const unsigned m0 = big_pack2(r0), m1 = big_pack2(r1);
    const big_f32x2 q0 = r0 - big_widen2(m0), q1 = r1 - big_widen2(m1);
    *reinterpret_cast<big_u32x2*>(dh) = big_u32x2{h0, h1};
    *reinterpret_cast<big_u32x2*>(dm) = big_u32x2{m0, m1};
    *reinterpret_cast<big_u32x2*>(dl) = big_u32x2{big_pack2(q0), big_pack2(q1)};
}

template <bool PRESPLIT, int NT, bool LNT, int MT = 1>
__global__ __launch_bounds__(256) void k_linear_big_split(const BigLaunch L) {
    constexpr int BN = 64 * NT, BM = 64 * MT;
    constexpr int A_PART = BM * BIG_LDH, B_PART = BN * BIG_LDH;
    __shared__ __attribute__((aligned(16))) __bf16 smem_h[3 * A_PART + 3 * B_PART];      // NT = 2: 46,080 B
    static_assert(sizeof(__bf16) * (3 * A_PART + 3 * B_PART) >= sizeof(float) * BIG_BM * (BN + 4), "epilogue tile");
    __bf16 (*sA)[BM][BIG_LDH] = reinterpret_cast<__bf16 (*)[BM][BIG_LDH]>(smem_h);
    __bf16 (*sB)[BN][BIG_LDH] = reinterpret_cast<__bf16 (*)[BN][BIG_LDH]>(smem_h + 3 * A_PART);

    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    int lin = blockIdx.y * nbx + blockIdx.x;
    if ((nblk & 7) == 0) lin = (lin & 7) * (nblk >> 3) + (lin >> 3);
    const int bx = lin % nbx, by = lin / nbx;

    int pi = L.np - 1;
    while (pi > 0 && bx < L.p[pi].tile_begin) --pi;
    const BigProblem& P = L.p[pi];
    const int n0 = (bx - P.tile_begin) * BN;
    const int m0 = by * BM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int K = P.K;

    const int srow = tid >> 3, skq = (tid & 7) * 4;
    const float* xa[2 * MT];
    const float* wb[4];
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) xa[i] = P.X + (int64_t)min(m0 + srow + 32 * i, L.M - 1) * P.ldx + skq;
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) wb[i] = P.W + (int64_t)min(n0 + srow + 32 * i, P.N - 1) * P.ldw + skq;

    // pre-split weights: BN rows x 32 k x 3 images = 12 BN 16-byte pieces per K-step, 3 NT per thread:
    // thread t takes row t / 4 (and row t / 4 + 64 for NT = 2), k offset 8 * (t % 4), of each image
    const int prow = tid >> 2, pkq = (tid & 3) * 8;
    const uint16_t* ws0 = nullptr;
    const uint16_t* ws1 = nullptr;
    int64_t wplane = 0;
    if (PRESPLIT) {
        ws0 = P.Ws + (int64_t)min(n0 + prow, P.N - 1) * K + pkq;
        ws1 = P.Ws + (int64_t)min(n0 + prow + 64, P.N - 1) * K + pkq;
        wplane = (int64_t)P.N * K;
    }
    __bf16* const wdst = &sB[0][prow][pkq];

    const int nk = K / BIG_BK;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // two register stages of global loads (stage = K-step parity): a K-step's operands are requested two steps ahead
    float4 raA[2 * MT], raB[2 * MT], rbA[4], rbB[4];
    u32x4 rsA[6], rsB[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { rsA[i] = (u32x4)(0u); rsB[i] = (u32x4)(0u); }
    // row statistics of this thread's two staging rows: the 8 threads of a row fetch two parts each
    const bool ln = LNT && P.ln_g != nullptr;
    float mu[2 * MT], rsd[2 * MT];
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) { mu[i] = 0.f; rsd[i] = 1.f; }
    float4 rg = make_float4(0.f, 0.f, 0.f, 0.f), re = rg;
    const float* gp = ln ? P.ln_g + skq : xa[0];
    const float* ep = ln ? P.ln_b + skq : xa[0];
    if (LNT) {
        const float* st = ln ? P.stats_in : P.X;
        const int np = ln ? P.parts_in : 1, pw = ln ? P.pw_in : 1;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int64_t row = ln ? (int64_t)min(m0 + srow + 32 * i, L.M - 1) : 0;
            const int p0 = (tid & 7) * 2, p1 = p0 + 1;
            const float2 a = *reinterpret_cast<const float2*>(st + (row * np + min(p0, np - 1)) * 2);
            const float2 b = *reinterpret_cast<const float2*>(st + (row * np + min(p1, np - 1)) * 2);
            const float sm = group_sum<8>((p0 < np ? a.x : 0.f) + (p1 < np ? b.x : 0.f));
            const float mean = sm / (float)K;
            // slices merged as (count, mean, M2) triples: M2 = sum M2_p + n_p (mean_p - mean)^2
            const float c0 = (float)min(pw, K - p0 * pw), c1 = (float)min(pw, K - p1 * pw);
            const float d0 = a.x / c0 - mean, d1 = b.x / c1 - mean;
            const float m2 = group_sum<8>((p0 < np ? a.y + c0 * d0 * d0 : 0.f) + (p1 < np ? b.y + c1 * d1 * d1 : 0.f));
            mu[i] = mean; rsd[i] = 1.0f / sqrtf(fmaxf(m2 / (float)K, 0.f) + LN_EPS);
        }
    }
    // loads are unconditional (past the end they re-read the last step)
#define BIGS_LOADG(ks_)                                                     \
    if (LNT) {                                                              \
        const int ko_ = min((ks_), nk - 1) * BIG_BK;                        \
        rg = *reinterpret_cast<const float4*>(gp + ko_);                    \
        re = *reinterpret_cast<const float4*>(ep + ko_);                    \
    }
#define BIGS_LOAD(S, ks_)                                                   \
    {                                                                       \
        const int ko = min((ks_), nk - 1) * BIG_BK;                         \
        _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i) ra##S[i] = *reinterpret_cast<const float4*>(xa[i] + ko); \
        if (PRESPLIT) {                                                     \
            rs##S[0] = *reinterpret_cast<const u32x4*>(ws0 + ko);           \
            rs##S[2] = *reinterpret_cast<const u32x4*>(ws0 + wplane + ko);  \
            rs##S[4] = *reinterpret_cast<const u32x4*>(ws0 + 2 * wplane + ko); \
            if (NT == 2) {                                                  \
                rs##S[1] = *reinterpret_cast<const u32x4*>(ws1 + ko);       \
                rs##S[3] = *reinterpret_cast<const u32x4*>(ws1 + wplane + ko); \
                rs##S[5] = *reinterpret_cast<const u32x4*>(ws1 + 2 * wplane + ko); \
            }                                                               \
        } else {                                                            \
            rb##S[0] = *reinterpret_cast<const float4*>(wb[0] + ko);        \
            rb##S[1] = *reinterpret_cast<const float4*>(wb[1] + ko);        \
            if (NT == 2) {                                                  \
                rb##S[2] = *reinterpret_cast<const float4*>(wb[2] + ko);    \
                rb##S[3] = *reinterpret_cast<const float4*>(wb[3] + ko);    \
            }                                                               \
        }                                                                   \
    }
#define BIGS_PUT(S, row, v) big_split_put(&S[0][row][skq], &S[1][row][skq], &S[2][row][skq], (v).x, (v).y, (v).z, (v).w);
#define BIGS_FRAG(S, part, row, half) (*reinterpret_cast<const bf16x8*>(&S[part][row][(half) * 16 + lh * 8]))
#define BIGS_MMA(a, b, acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    // smallest partial products first; the two accumulators alternate so that consecutive MFMAs are independent
#define BIGS_TWELVE(ah, am, al, b0h, b0m, b0l, b1h, b1m, b1l, A0, A1)       \
    BIGS_MMA(am, b0m, A0) BIGS_MMA(am, b1m, A1) BIGS_MMA(al, b0h, A0) BIGS_MMA(al, b1h, A1) \
    BIGS_MMA(ah, b0l, A0) BIGS_MMA(ah, b1l, A1) BIGS_MMA(am, b0h, A0) BIGS_MMA(am, b1h, A1) \
    BIGS_MMA(ah, b0m, A0) BIGS_MMA(ah, b1m, A1) BIGS_MMA(ah, b0h, A0) BIGS_MMA(ah, b1h, A1)
#define BIGS_SIX(ah, am, al, b0h, b0m, b0l, A0)                             \
    BIGS_MMA(am, b0m, A0) BIGS_MMA(al, b0h, A0) BIGS_MMA(ah, b0l, A0)       \
    BIGS_MMA(am, b0h, A0) BIGS_MMA(ah, b0m, A0) BIGS_MMA(ah, b0h, A0)
    // one K-step from register stage S: LDS <- registers (normalised, split), then the requests for the step after next
    // (the LayerNorm parameters of the next step first: they are needed one step earlier than the stage's operands)
#define BIGS_STEP(S, ks_)                                                   \
    {                                                                       \
        if ((ks_) > 0) __syncthreads();          /* every wave has read the previous stage */ \
        if (LNT && ln) {                                                    \
            _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i) {            \
                ra##S[i].x = big_silu((ra##S[i].x - mu[i]) * rsd[i] * rg.x + re.x); \
                ra##S[i].y = big_silu((ra##S[i].y - mu[i]) * rsd[i] * rg.y + re.y); \
                ra##S[i].z = big_silu((ra##S[i].z - mu[i]) * rsd[i] * rg.z + re.z); \
                ra##S[i].w = big_silu((ra##S[i].w - mu[i]) * rsd[i] * rg.w + re.w); \
            }                                                               \
        }                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i) BIGS_PUT(sA, srow + 32 * i, ra##S[i]) \
        if (PRESPLIT) {                                                     \
            *reinterpret_cast<u32x4*>(wdst) = rs##S[0];                     \
            *reinterpret_cast<u32x4*>(wdst + B_PART) = rs##S[2];            \
            *reinterpret_cast<u32x4*>(wdst + 2 * B_PART) = rs##S[4];        \
            if (NT == 2) {                                                  \
                *reinterpret_cast<u32x4*>(wdst + 64 * BIG_LDH) = rs##S[1];  \
                *reinterpret_cast<u32x4*>(wdst + B_PART + 64 * BIG_LDH) = rs##S[3]; \
                *reinterpret_cast<u32x4*>(wdst + 2 * B_PART + 64 * BIG_LDH) = rs##S[5]; \
            }                                                               \
        } else {                                                            \
            BIGS_PUT(sB, srow, rb##S[0])                                    \
            BIGS_PUT(sB, srow + 32, rb##S[1])                               \
            if (NT == 2) {                                                  \
                BIGS_PUT(sB, srow + 64, rb##S[2])                           \
                BIGS_PUT(sB, srow + 96, rb##S[3])                           \
            }                                                               \
        }                                                                   \
        __syncthreads();                                                    \
        BIGS_LOADG((ks_) + 1)                                               \
        BIGS_LOAD(S, (ks_) + 2)                                             \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {            \
            const bf16x8 ah = BIGS_FRAG(sA, 0, wm * 32 + li, half), am = BIGS_FRAG(sA, 1, wm * 32 + li, half), \
                         al = BIGS_FRAG(sA, 2, wm * 32 + li, half);         \
            const bf16x8 b0h = BIGS_FRAG(sB, 0, wn * 32 * NT + li, half), b0m = BIGS_FRAG(sB, 1, wn * 32 * NT + li, half), \
                         b0l = BIGS_FRAG(sB, 2, wn * 32 * NT + li, half);   \
            if (NT == 2) {                                                  \
                const bf16x8 b1h = BIGS_FRAG(sB, 0, wn * 64 + 32 + li, half), b1m = BIGS_FRAG(sB, 1, wn * 64 + 32 + li, half), \
                             b1l = BIGS_FRAG(sB, 2, wn * 64 + 32 + li, half); \
                BIGS_TWELVE(ah, am, al, b0h, b0m, b0l, b1h, b1m, b1l, acc0, acc1) \
                if (MT == 2) {          /* the wave's second row tile: rows 64 + 32 wm ..., the same weight fragments */ \
                    const bf16x8 ch = BIGS_FRAG(sA, 0, 64 + wm * 32 + li, half), cm = BIGS_FRAG(sA, 1, 64 + wm * 32 + li, half), \
                                 cl = BIGS_FRAG(sA, 2, 64 + wm * 32 + li, half); \
                    BIGS_TWELVE(ch, cm, cl, b0h, b0m, b0l, b1h, b1m, b1l, acc2, acc3) \
                }                                                           \
            } else {                                                        \
                BIGS_SIX(ah, am, al, b0h, b0m, b0l, acc0)                   \
            }                                                               \
        }                                                                   \
    }

    BIGS_LOADG(0)
    BIGS_LOAD(A, 0)
    BIGS_LOAD(B, 1)
    static_assert(MT == 1 || NT == 2, "128-row tiles come with 128 columns");
    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }

    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {               // no branch inside: a branch around a step's loads makes every wait a full one
        BIGS_STEP(A, ks)
        BIGS_STEP(B, ks + 1)
    }
    if (nk & 1) BIGS_STEP(A, ks)
#undef BIGS_LOADG
#undef BIGS_LOAD
#undef BIGS_PUT
#undef BIGS_FRAG
#undef BIGS_MMA
#undef BIGS_TWELVE
#undef BIGS_SIX
#undef BIGS_STEP

    // ---- epilogue (as k_linear_big): tile through LDS, + bias, + one-hot table rows, 16-byte stores
    float (*sC)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(smem_h);
#pragma unroll
    for (int pass = 0; pass < MT; ++pass) {      // 64 rows at a time: rows 64 pass + 32 wm ... of every wave
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sC[wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh][wn * 32 * NT + li] = pass == 0 ? acc0[i] : acc2[i];
            if (NT == 2) sC[wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh][wn * 64 + 32 + li] = pass == 0 ? acc1[i] : acc3[i];
        }
        __syncthreads();
        big_epilogue<BN>(L, P, sC, m0 + 64 * pass, n0, tid);
    }
}

// W (N, K) with row stride ldw -> bf16 images [planes][N][K]: h = bf16(w) and, with three planes,
// m = bf16(w - h), l = bf16(w - h - m)
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ W, int64_t ldw, int N, int K,
                                                       uint16_t* __restrict__ planes, int nplanes) {
    const int64_t total4 = (int64_t)N * K / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
        const float4 v = *reinterpret_cast<const float4*>(W + (int64_t)n * ldw + k);
        const float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, m, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            h[c] = (__bf16)f[c];
            const float r1 = f[c] - (float)h[c];
            m[c] = (__bf16)r1;
            l[c] = (__bf16)(r1 - (float)m[c]);
        }
        __bf16* out = reinterpret_cast<__bf16*>(planes);
        *reinterpret_cast<bf16x4*>(out + e) = h;
        if (nplanes == 3) {
            *reinterpret_cast<bf16x4*>(out + (int64_t)N * K + e) = m;
            *reinterpret_cast<bf16x4*>(out + 2 * (int64_t)N * K + e) = l;
        }
    }
}

// ---- elementwise companions of the large-row path ---------------------------------------------------
// Y = SiLU(LayerNorm(X row)); one wave per row, the row (K <= 256 * NV) kept in registers (two-pass statistics).
template <int NV>
__global__ __launch_bounds__(256) void k_ln_silu_rows(const float* __restrict__ X, int64_t ldx, const float* __restrict__ g,
                                                       const float* __restrict__ e, float* __restrict__ Y, int64_t ldy,
                                                       int M, int K) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const float* x = X + (int64_t)row * ldx;
    float4 v[NV], gg[NV], ee[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        v[i] = k < K ? *reinterpret_cast<const float4*>(x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (NV <= 4) {   // scale and shift fetched beside the row, not after the two reductions
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = min((i * 64 + lane) * 4, K - 4);
            gg[i] = *reinterpret_cast<const float4*>(g + k);
            ee[i] = *reinterpret_cast<const float4*>(e + k);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = group_sum<64>(s) / (float)K;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        if (k < K) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(group_sum<64>(q) / (float)K + LN_EPS);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        if (k < K) {
            if (NV > 4) {
                gg[i] = *reinterpret_cast<const float4*>(g + k);
                ee[i] = *reinterpret_cast<const float4*>(e + k);
            }
            float4 o;
            o.x = silu_f((v[i].x - mean) * rstd * gg[i].x + ee[i].x);
            o.y = silu_f((v[i].y - mean) * rstd * gg[i].y + ee[i].y);
            o.z = silu_f((v[i].z - mean) * rstd * gg[i].z + ee[i].z);
            o.w = silu_f((v[i].w - mean) * rstd * gg[i].w + ee[i].w);
            *reinterpret_cast<float4*>(Y + (int64_t)row * ldy + k) = o;
        }
    }
}

// h_new = GRU gates(gi (bias excluded), gh (bias included), h_prev), elementwise over (M, D)
__global__ __launch_bounds__(256) void k_gru_gates(const float* __restrict__ gi, int64_t ldgi, const float* __restrict__ b_ih,
                                                    const float* __restrict__ gh, int64_t ldgh,
                                                    const float* __restrict__ h_prev, int64_t ldh, float* __restrict__ h_new,
                                                    int64_t ldhn, int M, int D) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * D) return;
    const int m = (int)(idx / D), j = (int)(idx - (int64_t)m * D);
    const float* a = gi + (int64_t)m * ldgi;
    const float* b = gh + (int64_t)m * ldgh;
    const float rr = sigmoid_f(a[j] + b_ih[j] + b[j]);
    const float uu = sigmoid_f(a[D + j] + b_ih[D + j] + b[D + j]);
    const float nn = tanhf((a[2 * D + j] + b_ih[2 * D + j]) + rr * b[2 * D + j]);
    h_new[(int64_t)m * ldhn + j] = (1.0f - uu) * nn + uu * h_prev[(int64_t)m * ldh + j];
}

}  // namespace drm
