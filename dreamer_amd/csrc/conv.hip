// Stride-2 4x4 convolutions of the encoder / decoder towers (nn.Conv2d(k4, s2, p1), lib/world_model.py:94-108, and
// nn.ConvTranspose2d(k4, s2, p1), lib/world_model.py:290-302) as implicit matrix products on the matrix cores,
// activations in channels-last memory (N, H, W, C), no im2col buffer.  Two forms with the same results to fp32 rounding:
// k_conv_gemm on v_mfma_f32_32x32x2_f32, and (the default) k_conv_split / k_conv_up4_split on v_mfma_f32_32x32x16_bf16
// with both operands split exactly into three bf16 parts.
//
// One layer relates a big-resolution tensor T (NI, 2Hs, 2Ws, cb) and a small-resolution tensor S (NI, Hs, Ws, cs)
// through a kernel Q stored [cs][kh][kw][cb]: nn.Conv2d maps T -> S (Q = weight (cs, cb, 4, 4) in channels-last
// memory), nn.ConvTranspose2d maps S -> T (Q = weight (cs, cb, 4, 4) in channels-last memory as well: its first
// dimension is the input = small-resolution side).  Pixel relation: big (y, x) meets small (a, b) through tap
// (kh, kw) when y = 2a - 1 + kh, x = 2b - 1 + kw.  Three products cover forward and both gradients of both layers:
//
//   DOWN  S[m, j]        = sum_{kh,kw,c} T[pix(m; kh,kw), c] Q[j, kh, kw, c]        conv forward, convT input gradient
//   UP    T[p, c]        = sum_{taps,s}  S[pix(p; tap), s]   Q[s, kh, kw, c]        convT forward, conv input gradient
//                          (per parity class of p only 2 x 2 of the 16 taps meet a small pixel)
//   WGRAD dQ[j, kh,kw,c] = sum_m         S[m, j] T[pix(m; kh,kw), c]                weight gradient of both
//
// k_conv_gemm: workgroup = 4 waves = 64 x 64 output tile (2 x 2 waves, one 32 x 32 accumulator each), reduction in steps of 32
// through two LDS buffers with two register stages of global loads in flight.  Operands whose reduction index is
// contiguous in memory are staged [row][k] (stride 36 words, 16-byte fragment reads), the others [k][row] (stride 68,
// one word per lane and MFMA).  WGRAD splits its long reduction (all output pixels) over blockIdx.z into partial
// tiles that a second kernel sums in a fixed order.
#include "common.hpp"
#include <cstdlib>
#include "../../include/dreamer_amd.h"

namespace drm {

constexpr int CV_BK = 32, CV_LDK = 36;
constexpr int cv_stage_words(int rb) { return 64 * rb * CV_LDK > CV_BK * (64 * rb + 4) ? 64 * rb * CV_LDK : CV_BK * (64 * rb + 4); }

struct ConvArgs {
    const float* T; const float* S; const float* Q;   // DOWN reads T, Q; UP reads S, Q; WGRAD reads S, T
    float* out;                                        // DOWN: S-shaped; UP: T-shaped; WGRAD: Q-shaped (or partials)
    int NI, Hs, Ws, cs, cb;
    int ksteps_per_split;                              // WGRAD
    float* partial;                                    // WGRAD with splits > 1: (splits, cs, 16 cb)
    // MAT_TN: S = A (K = NI rows, lda), T = B (K rows, ldb), out (cs = M, cb = N; ldc);  MAT_NT: S = A (NI = M rows, lda),
    // Q = B (cs = K rows, ldb), out (M, cb = N; ldc)
    int64_t lda = 0, ldb = 0, ldc = 0;
};

enum { CONV_DOWN = 0, CONV_UP = 1, CONV_WGRAD = 2,
       // plain matrices on the split kernel's staging (no pixel geometry), for the large Linear gradients whose operands
       // are not reduction-contiguous: C (M, N) = A^T B with A (K, M), B (K, N) [weight gradient dW = dY^T X], and
       // C (M, N) = A B with A (M, K), B (K, N) [input gradient dX = dY W]
       MAT_TN = 3, MAT_NT = 4 };

// An operand tile of R = 64 * RB rows (RB = 1 or 2) by 32 reduction steps; every thread moves 2 * RB f32x4.
//   [row][k] image: rows (tid >> 3) + 32 i, k offset (tid & 7) * 4, row stride 36 words
//   [k][row] image: reduction rows (tid >> 4) + 16 (i % 2), columns (tid & 15) * 4 + 64 (i / 2), row stride 64 RB + 4 words
template <int RB>
__device__ __forceinline__ void st_rowk(float* buf, int tid, const f32x4 (&v)[2 * RB]) {
    const int kq = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2 * RB; ++i) *reinterpret_cast<f32x4*>(buf + ((tid >> 3) + 32 * i) * CV_LDK + kq) = v[i];
}
template <int RB>
__device__ __forceinline__ void st_krow(float* buf, int tid, const f32x4 (&v)[2 * RB]) {
    const int r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2 * RB; ++i)
        *reinterpret_cast<f32x4*>(buf + ((tid >> 4) + 16 * (i & 1)) * (64 * RB + 4) + r4 + 64 * (i >> 1)) = v[i];
}
__device__ __forceinline__ f32x4 fr_rowk(const float* buf, int row, int g, int lh) {
    return *reinterpret_cast<const f32x4*>(buf + row * CV_LDK + g * 8 + lh * 4);
}
template <int RB>
__device__ __forceinline__ f32x4 fr_krow(const float* buf, int row, int g, int lh) {
    constexpr int LD = 64 * RB + 4;
    const float* p = buf + (g * 8 + lh * 4) * LD + row;
    return f32x4{p[0], p[LD], p[2 * LD], p[3 * LD]};
}

// MT x NT: 32 x 32 accumulators per wave (rows x columns); WN: waves side by side along the columns (2: waves 2 x 2,
// workgroup tile 64 MT x 64 NT; 1: waves 4 x 1, tile 128 MT x 32 NT, for outputs of 32 columns).  RA / RB: 64-row
// blocks of the staged operand tiles (a 32-column B tile is staged as half of a 64-column image).
template <int MODE, int MT, int NT, int WN = 2>
__global__ __launch_bounds__(256) void k_conv_gemm(const ConvArgs G) {
    constexpr int BM = 32 * MT * (4 / WN), BN = 32 * NT * WN;
    constexpr int RA = BM / 64, RB = BN >= 64 ? BN / 64 : 1;
    constexpr int A_WORDS = cv_stage_words(RA), B_WORDS = cv_stage_words(RB);
    constexpr int SMEM_WORDS = 2 * (A_WORDS + B_WORDS) > BM * (BN + 4) ? 2 * (A_WORDS + B_WORDS) : BM * (BN + 4);
    __shared__ __attribute__((aligned(16))) float smem[SMEM_WORDS];
    float* sA[2] = {smem, smem + A_WORDS};
    float* sB[2] = {smem + 2 * A_WORDS, smem + 2 * A_WORDS + B_WORDS};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = WN == 2 ? (wave & 1) : wave, wn = WN == 2 ? (wave >> 1) : 0, li = lane & 31, lh = lane >> 5;
    const int Hs = G.Hs, Ws = G.Ws, cs = G.cs, cb = G.cb, Hb = 2 * Hs, Wb = 2 * Ws;
    const int pixels = G.NI * Hs * Ws;                 // small-resolution pixels
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // ---- per-thread row bookkeeping
    // [row][k] staging: rows (tid >> 3) + 32 i; [k][row] staging: reduction rows (tid >> 4) + 16 i, columns (tid & 15) * 4
    int pn[2 * RA], pa[2 * RA], pb[2 * RA];   // image, small row, small column of the thread's A rows
    bool pv[2 * RA];
#pragma unroll
    for (int i = 0; i < 2 * RA; ++i) { pn[i] = pa[i] = pb[i] = 0; pv[i] = false; }
    int py = 0, px = 0;
    if (MODE == CONV_DOWN || MODE == CONV_UP) {
        if (MODE == CONV_UP) { py = blockIdx.z >> 1; px = blockIdx.z & 1; }
#pragma unroll
        for (int i = 0; i < 2 * RA; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            pv[i] = m < pixels;
            const int mm = pv[i] ? m : 0;
            pn[i] = mm / (Hs * Ws);
            const int r = mm - pn[i] * (Hs * Ws);
            pa[i] = r / Ws;
            pb[i] = r - pa[i] * Ws;
        }
    }
    // WGRAD: the thread's 4 output columns (kh, kw, c) are fixed for the whole reduction
    int w_kh[RB], w_kw[RB], w_c[RB];
    bool w_nv[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int col = n0 + (tid & 15) * 4 + 64 * q;
        w_nv[q] = MODE == CONV_WGRAD && col < 16 * cb;
        const int cell = (w_nv[q] ? col : 0) / cb;
        w_c[q] = (w_nv[q] ? col : 0) - cell * cb;
        w_kh[q] = cell >> 2; w_kw[q] = cell & 3;
    }

    int nk, ks0 = 0;
    if (MODE == CONV_DOWN) nk = 16 * cb / CV_BK;
    else if (MODE == CONV_UP) nk = 4 * cs / CV_BK;
    else {
        const int nk_all = (pixels + CV_BK - 1) / CV_BK;
        ks0 = blockIdx.z * G.ksteps_per_split;
        nk = min(ks0 + G.ksteps_per_split, nk_all) - ks0;
    }

    // one f32x4 of an operand tile per call; the callers' loops are fully unrolled so that the staging arrays stay in
    // registers (a lambda capturing them by reference makes them addressable, and they end up in scratch memory)
#define CV_LOAD_A(v, ks_)                                                                                                \
    {                                                                                                                    \
        const int kst = (ks_);                                                                                            \
        if (MODE == CONV_DOWN) {                                                                                         \
            /* im2col row of T: the 32 k of this step lie inside one (kh, kw) cell (cb % 32 == 0) */                    \
            const int k0 = kst * CV_BK, cell = k0 / cb, c0 = k0 - cell * cb + (tid & 7) * 4;                              \
            const int kh = cell >> 2, kw = cell & 3;                                                                     \
            _Pragma("unroll") for (int i = 0; i < 2 * RA; ++i) {                                                         \
                const int y = 2 * pa[i] - 1 + kh, x = 2 * pb[i] - 1 + kw;                                                \
                const bool ok = pv[i] && (unsigned)y < (unsigned)Hb && (unsigned)x < (unsigned)Wb;                       \
                v[i] = ok ? *reinterpret_cast<const f32x4*>(G.T + ((int64_t)(pn[i] * Hb + y) * Wb + x) * cb + c0) : zero; \
            }                                                                                                            \
        } else if (MODE == CONV_UP) {                                                                                    \
            /* small-resolution pixel met through tap (th, tw) of this parity class (cs % 32 == 0: one tap per step) */ \
            const int k0 = kst * CV_BK, tap = k0 / cs, s0 = k0 - tap * cs + (tid & 7) * 4;                                \
            const int th = tap >> 1, tw = tap & 1;                                                                       \
            const int da = py ? (th ? 0 : 1) : (th ? -1 : 0), db = px ? (tw ? 0 : 1) : (tw ? -1 : 0);                    \
            _Pragma("unroll") for (int i = 0; i < 2 * RA; ++i) {                                                         \
                const int a = pa[i] + da, b = pb[i] + db;                                                                \
                const bool ok = pv[i] && (unsigned)a < (unsigned)Hs && (unsigned)b < (unsigned)Ws;                       \
                v[i] = ok ? *reinterpret_cast<const f32x4*>(G.S + ((int64_t)(pn[i] * Hs + a) * Ws + b) * cs + s0) : zero; \
            }                                                                                                            \
        } else {                                                                                                         \
            /* S as (pixels, cs): reduction row m, 4 consecutive output rows j */                                        \
            _Pragma("unroll") for (int i = 0; i < 2 * RA; ++i) {                                                         \
                const int j = m0 + (tid & 15) * 4 + 64 * (i >> 1);                                                       \
                const int m = (ks0 + kst) * CV_BK + (tid >> 4) + 16 * (i & 1);                                            \
                v[i] = (m < pixels && j < cs) ? *reinterpret_cast<const f32x4*>(G.S + (int64_t)m * cs + j) : zero;      \
            }                                                                                                            \
        }                                                                                                                \
    }
#define CV_LOAD_B(v, ks_)                                                                                                \
    {                                                                                                                    \
        const int kst = (ks_);                                                                                            \
        if (MODE == CONV_DOWN) {                                                                                         \
            const int k = kst * CV_BK + (tid & 7) * 4;                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2 * RB; ++i) {                                                         \
                const int j = n0 + (tid >> 3) + 32 * i;                                                                  \
                v[i] = j < cs ? *reinterpret_cast<const f32x4*>(G.Q + (int64_t)j * 16 * cb + k) : zero;                 \
            }                                                                                                            \
        } else if (MODE == CONV_UP) {                                                                                    \
            const int k0 = kst * CV_BK, tap = k0 / cs, s0 = k0 - tap * cs;                                                \
            const int th = tap >> 1, tw = tap & 1;                                                                       \
            const int kh = py ? (th ? 2 : 0) : (th ? 3 : 1), kw = px ? (tw ? 2 : 0) : (tw ? 3 : 1);                      \
            _Pragma("unroll") for (int i = 0; i < 2 * RB; ++i) {                                                         \
                const int c = n0 + (tid & 15) * 4 + 64 * (i >> 1);                                                       \
                const int s = s0 + (tid >> 4) + 16 * (i & 1);                                                            \
                v[i] = c < cb ? *reinterpret_cast<const f32x4*>(G.Q + ((int64_t)s * 16 + kh * 4 + kw) * cb + c) : zero; \
            }                                                                                                            \
        } else {                                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                              \
                const int m = (ks0 + kst) * CV_BK + (tid >> 4) + 16 * i;                                                  \
                const bool in = m < pixels;                                                                              \
                const int mm = in ? m : 0;                                                                               \
                const int n = mm / (Hs * Ws), r = mm - n * (Hs * Ws), a = r / Ws, b = r - a * Ws;                        \
                _Pragma("unroll") for (int q = 0; q < RB; ++q) {                                                         \
                    const int y = 2 * a - 1 + w_kh[q], x = 2 * b - 1 + w_kw[q];                                          \
                    const bool ok = in && w_nv[q] && (unsigned)y < (unsigned)Hb && (unsigned)x < (unsigned)Wb;           \
                    v[i + 2 * q] = ok ? *reinterpret_cast<const f32x4*>(G.T + ((int64_t)(n * Hb + y) * Wb + x) * cb + w_c[q]) : zero; \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
    }
    constexpr bool A_ROWK = MODE != CONV_WGRAD;     // DOWN / UP: A is [row][k]; WGRAD: A is [k][row]
    constexpr bool B_ROWK = MODE == CONV_DOWN;      // DOWN: B = Q rows [j][k]; UP / WGRAD: [k][col]

    f32x4 a0[2 * RA], a1[2 * RA], b0[2 * RB], b1[2 * RB];
#pragma unroll
    for (int i = 0; i < 2 * RA; ++i) a0[i] = a1[i] = zero;
#pragma unroll
    for (int i = 0; i < 2 * RB; ++i) b0[i] = b1[i] = zero;
    if (nk > 0) { CV_LOAD_A(a0, 0) CV_LOAD_B(b0, 0) }
    if (nk > 1) { CV_LOAD_A(a1, 1) CV_LOAD_B(b1, 1) }

    f32x16 acc[MT * NT];
#pragma unroll
    for (int i = 0; i < MT * NT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // one K-step: four groups of 8 k; the fragments of group g + 1 are read before the MFMAs of group g
    // (macros, not lambdas: accumulators captured by reference end up in scratch memory)
#define CV_FRAGS(buf, g, fa, fb)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                       \
        fa[i] = A_ROWK ? fr_rowk(sA[buf], wm * 32 * MT + 32 * i + li, g, lh) : fr_krow<RA>(sA[buf], wm * 32 * MT + 32 * i + li, g, lh); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                       \
        fb[j] = B_ROWK ? fr_rowk(sB[buf], wn * 32 * NT + 32 * j + li, g, lh) : fr_krow<RB>(sB[buf], wn * 32 * NT + 32 * j + li, g, lh);
#define CV_MMA1(fa, fb, c)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                       \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                   \
            acc[i * NT + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].c, fb[j].c, acc[i * NT + j], 0, 0, 0);
#define CV_MMA(fa, fb) CV_MMA1(fa, fb, x) CV_MMA1(fa, fb, y) CV_MMA1(fa, fb, z) CV_MMA1(fa, fb, w)
#define CV_STEP(buf)                                   \
    {                                                  \
        f32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];     \
        CV_FRAGS(buf, 0, fa0, fb0)                     \
        CV_FRAGS(buf, 1, fa1, fb1)                     \
        CV_MMA(fa0, fb0)                               \
        CV_FRAGS(buf, 2, fa0, fb0)                     \
        CV_MMA(fa1, fb1)                               \
        CV_FRAGS(buf, 3, fa1, fb1)                     \
        CV_MMA(fa0, fb0)                               \
        CV_MMA(fa1, fb1)                               \
    }
#define CV_STAGE(av, bv, buf)                                                             \
    if (A_ROWK) st_rowk<RA>(sA[buf], tid, av); else st_krow<RA>(sA[buf], tid, av);        \
    if (B_ROWK) st_rowk<RB>(sB[buf], tid, bv); else st_krow<RB>(sB[buf], tid, bv);

    for (int ks = 0; ks < nk; ks += 2) {
        CV_STAGE(a0, b0, 0)
        __syncthreads();
        if (ks + 2 < nk) { CV_LOAD_A(a0, ks + 2) CV_LOAD_B(b0, ks + 2) }
        CV_STEP(0)
        if (ks + 1 < nk) {
            CV_STAGE(a1, b1, 1)
            __syncthreads();
            if (ks + 3 < nk) { CV_LOAD_A(a1, ks + 3) CV_LOAD_B(b1, ks + 3) }
            CV_STEP(1)
        }
    }
#undef CV_LOAD_A
#undef CV_LOAD_B
#undef CV_FRAGS
#undef CV_MMA1
#undef CV_MMA
#undef CV_STEP
#undef CV_STAGE

    // ---- epilogue through LDS: every thread owns 4 consecutive columns of a row
    __syncthreads();
    float (*sC)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(smem);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                sC[wm * 32 * MT + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh][wn * 32 * NT + 32 * j + li] = acc[i * NT + j][e];
    __syncthreads();
    constexpr int TPR = BN / 4;                 // threads per output row
    const int c4 = (tid % TPR) * 4, col = n0 + c4;
    const int ncols = MODE == CONV_DOWN ? cs : (MODE == CONV_UP ? cb : 16 * cb);
    const int nrows = MODE == CONV_WGRAD ? cs : pixels;
    if (col >= ncols) return;
    for (int r = tid / TPR; r < BM; r += 256 / TPR) {
        const int m = m0 + r;
        if (m >= nrows) break;
        const f32x4 y = *reinterpret_cast<const f32x4*>(&sC[r][c4]);
        float* o;
        if (MODE == CONV_DOWN) o = G.out + (int64_t)m * cs + col;
        else if (MODE == CONV_UP) {
            const int n = m / (Hs * Ws), rr = m - n * (Hs * Ws), a = rr / Ws, b = rr - a * Ws;
            o = G.out + ((int64_t)(n * Hb + 2 * a + py) * Wb + 2 * b + px) * cb + col;
        } else {
            float* base = G.partial ? G.partial + (int64_t)blockIdx.z * cs * 16 * cb : G.out;
            o = base + (int64_t)m * 16 * cb + col;
        }
        *reinterpret_cast<f32x4*>(o) = y;      // ncols % 4 == 0
    }
}

// out = sum over the slices of partial, in a fixed order.  A workgroup owns 32 consecutive 16-byte outputs; its 8 groups of
// 32 threads add every eighth slice each (four loads in flight per thread), and group 0 adds the eight sums in group order.
// Many threads per output on purpose: the partials are 17-34 MB for a 2 MB result, and one thread walking 128 slices of an
// output is a chain of load latencies (19 us); 8 x more loads in flight stream them.
__global__ __launch_bounds__(256) void k_conv_reduce(const float* __restrict__ partial, int splits, int64_t total4,
                                                      float* __restrict__ out) {
    __shared__ f32x4 red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + o;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < total4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(partial) + i;
        int z = g;
        for (; z + 24 < splits; z += 32) {
            const f32x4 v0 = p[(int64_t)z * total4], v1 = p[(int64_t)(z + 8) * total4], v2 = p[(int64_t)(z + 16) * total4],
                        v3 = p[(int64_t)(z + 24) * total4];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; z < splits; z += 8) acc += p[(int64_t)z * total4];
    }
    red[g][o] = acc;
    __syncthreads();
    if (g == 0 && i < total4) {
        f32x4 t = red[0][o];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += red[q][o];
        reinterpret_cast<f32x4*>(out)[i] = t;
    }
}

// ---- the same three products on the bf16 matrix pipe: exact three-way split of both operands ------------------------
// Every fp32 operand value is split exactly into three bf16 parts (a = h + m + l: 8 + 8 + 8 mantissa bits) on its way
// into LDS and a*b is accumulated in fp32 from the six partial products above 2^-24 |a b|, as in the large Linear
// products: results agree with the fp32 form to fp32 rounding while the matrix pipe runs v_mfma_f32_32x32x16_bf16,
// a third of the pipe time of v_mfma_f32_32x32x2_f32 for the same reduction range.
//
// LDS image of an operand tile of R rows x 32 k, one per part: [k / 8][slot][8 k] (16 bytes per slot and k group; a
// lane's MFMA fragment is one 16-byte read); the k groups start 0, 128, 64 and 192 bytes into the 256 bytes of the
// banks, so that the four k groups that 32 lanes of a staging wave write (4 rows x 16 bytes each) fall on the four
// quarters of the banks and the two k groups of one MFMA's fragments (lanes 0-31 / 32-63) on different halves.  Operands whose reduction index is contiguous in memory are staged as in
// the fp32 form (slot = row).  The others (reduction index = pixel or input channel, rows contiguous) are loaded as
// 4 k x 4 row blocks per thread -- rows 4 g ... 4 g + 3 of k quad q, 16 bytes per k -- and transposed in registers; row
// 4 g + e goes to slot e R/4 + g and neighbouring lanes hold the two k quads of a k group, which makes the 8-byte LDS
// writes of 32 lanes one contiguous 256 bytes.  Which tile row an MFMA row
// stands for is undone on the way out, in the epilogue's tile through LDS.
__device__ __attribute__((aligned(16))) const float cv_zeros[4] = {0.f, 0.f, 0.f, 0.f};   // what a load of an invalid element reads
typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cv_bf16x4 __attribute__((ext_vector_type(4)));
// element offset of k group kg (0 ... 3) in a part of R rows, and the elements of a part
__device__ __host__ constexpr int cvs_kgo(int rows, int kg) { return kg * rows * 8 + 64 * (kg & 1) + 160 * (kg >> 1); }
constexpr int cvs_part(int rows) { return 4 * rows * 8 + 256; }

// four values -> their three bf16 parts, pairwise: one v_cvt_pk_bf16_f32 per pair and part, the parts widened back by a
// shift / a mask, residuals as packed fp32 subtractions
typedef float cv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cv_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int cv_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvs_pack(cv_f32x2 v) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cv_bf16x2));
}
__device__ __forceinline__ cv_f32x2 cvs_widen(unsigned p) {
    return cv_f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}
__device__ __forceinline__ void cvs_put(__bf16* dst, int part_stride, float f0, float f1, float f2, float f3) {
    const cv_f32x2 x0{f0, f1}, x1{f2, f3};
    const unsigned h0 = cvs_pack(x0), h1 = cvs_pack(x1);
    const cv_f32x2 r0 = x0 - cvs_widen(h0), r1 = x1 - cvs_widen(h1);
    const unsigned m0 = cvs_pack(r0), m1 = cvs_pack(r1);
    const cv_f32x2 q0 = r0 - cvs_widen(m0), q1 = r1 - cvs_widen(m1);
    *reinterpret_cast<cv_u32x2*>(dst) = cv_u32x2{h0, h1};
    *reinterpret_cast<cv_u32x2*>(dst + part_stride) = cv_u32x2{m0, m1};
    *reinterpret_cast<cv_u32x2*>(dst + 2 * part_stride) = cv_u32x2{cvs_pack(q0), cvs_pack(q1)};
}

// PARTS = 1: the operands rounded to bf16 (their first part only) -- a bf16-input product with fp32 accumulation, one
// MFMA where the exact form has six, a third of the LDS image
template <int PARTS>
__device__ __forceinline__ void cvs_putp(__bf16* dst, int part_stride, float f0, float f1, float f2, float f3) {
    if (PARTS == 3) cvs_put(dst, part_stride, f0, f1, f2, f3);
    else *reinterpret_cast<cv_u32x2*>(dst) = cv_u32x2{cvs_pack(cv_f32x2{f0, f1}), cvs_pack(cv_f32x2{f2, f3})};
}

template <int MODE, int MT, int NT, int WN = 2, int PARTS = 3>
__global__ __launch_bounds__(256) void k_conv_split(const ConvArgs G) {
    constexpr int BM = 32 * MT * (4 / WN), BN = 32 * NT * WN;
    constexpr bool A_ROWK = MODE != CONV_WGRAD && MODE != MAT_TN;   // DOWN / UP / NT: A is [row][k]; WGRAD / TN: A is [k][row]
    constexpr bool B_ROWK = MODE == CONV_DOWN;                      // DOWN: B = Q rows [j][k]; the others: [k][col]
    constexpr int A_KGS = BM, B_KGS = BN, A_PART = cvs_part(BM), B_PART = cvs_part(BN);      // (the macros' KGS = rows of the part)
    constexpr int STAGE_BYTES = 2 * PARTS * (A_PART + B_PART), TILE_BYTES = 4 * BM * (BN + 4);
    constexpr int NVB = B_ROWK ? BN / 32 : 4;       // f32x4 per thread and K-step of the B tile (A: 4 either way)
    static_assert(A_ROWK ? BM == 128 : true, "pixel-row tiles are 128 rows");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES];
    __bf16* const sA = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* const sB = sA + PARTS * A_PART;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = WN == 2 ? (wave & 1) : wave, wn = WN == 2 ? (wave >> 1) : 0, li = lane & 31, lh = lane >> 5;
    const int Hs = G.Hs, Ws = G.Ws, cs = G.cs, cb = G.cb, Hb = 2 * Hs, Wb = 2 * Ws;
    const int pixels = G.NI * Hs * Ws;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // [row][k] staging: rows (tid >> 3) + 32 i, k offset (tid & 7) * 4
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    // [k][row] staging: row group and k quad of this thread's 4 x 4 block (threads past the tile's 8 k quads stage nothing)
    const int a_rg = (tid >> 1) % (BM / 4), a_kq = (tid & 1) + 2 * ((tid >> 1) / (BM / 4));
    const int b_rg = (tid >> 1) % (BN / 4), b_kq = (tid & 1) + 2 * ((tid >> 1) / (BN / 4));
    const bool a_on = A_ROWK || a_kq < 8, b_on = B_ROWK || b_kq < 8;

    int pn[4], pa[4], pb[4];                        // image, small row, small column of the thread's A rows (DOWN / UP)
    bool pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { pn[i] = pa[i] = pb[i] = 0; pv[i] = false; }
    int py = 0, px = 0;
    if (MODE == MAT_NT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = m0 + srow + 32 * i < pixels;
    }
    if (MODE == CONV_DOWN || MODE == CONV_UP) {
        if (MODE == CONV_UP) { py = blockIdx.z >> 1; px = blockIdx.z & 1; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + srow + 32 * i;
            pv[i] = m < pixels;
            const int mm = pv[i] ? m : 0;
            pn[i] = mm / (Hs * Ws);
            const int r = mm - pn[i] * (Hs * Ws);
            pa[i] = r / Ws;
            pb[i] = r - pa[i] * Ws;
        }
    }
    // WGRAD: the thread's 4 output columns (kh, kw, c) are fixed for the whole reduction
    int w_kh = 0, w_kw = 0, w_c = 0;
    bool w_nv = false;
    if (MODE == CONV_WGRAD) {
        const int col = n0 + 4 * b_rg;
        w_nv = col < 16 * cb;
        const int cell = (w_nv ? col : 0) / cb;
        w_c = (w_nv ? col : 0) - cell * cb;
        w_kh = cell >> 2; w_kw = cell & 3;
    }

    // order of the reduction: DOWN walks the 16 taps parity class by parity class ((kh, kw) = (0,0) (0,2) (2,0) (2,2),
    // then (0,1) ...): the four taps of a class read the same input pixels shifted by one output pixel, so three of
    // four staging loads of a tile hit lines the CU's vector cache just held, where the natural order comes back to a
    // line 2 or 8 steps later, after the tile's other 16 or 128 KB.  UP walks the 4 taps (the 9 neighbours) inside a
    // 32-channel chunk for the same reason.
    const int cpc = cb / CV_BK;                     // K-steps per tap (DOWN)
    int nk, ks0 = 0;
    if (MODE == CONV_DOWN) nk = 16 * cb / CV_BK;
    else if (MODE == CONV_UP) nk = 4 * cs / CV_BK;
    else if (MODE == MAT_NT) nk = cs / CV_BK;
    else {
        const int nk_all = (pixels + CV_BK - 1) / CV_BK;
        ks0 = blockIdx.z * G.ksteps_per_split;
        nk = min(ks0 + G.ksteps_per_split, nk_all) - ks0;
    }
    const int nk_last = max(nk, 1) - 1;

    // loads are unconditional -- an invalid element reads 16 bytes of zeros instead, steps past the end read the last
    // step again: a branch around a load makes the compiler wait for every outstanding load at the join, i.e. for the
    // loads just issued
#define CS_LOAD_A(v, ks_)                                                                                                \
    {                                                                                                                    \
        const int kst = min((ks_), nk_last);                                                                             \
        if (MODE == CONV_DOWN) {                                                                                         \
            const int t = kst / cpc, kh = ((t >> 3) & 1) + ((t >> 1) & 1) * 2, kw = ((t >> 2) & 1) + (t & 1) * 2;         \
            const int c0 = (kst - t * cpc) * CV_BK + skq;                                                                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
                const int y = 2 * pa[i] - 1 + kh, x = 2 * pb[i] - 1 + kw;                                                \
                const bool ok = pv[i] && (unsigned)y < (unsigned)Hb && (unsigned)x < (unsigned)Wb;                       \
                v[i] = *reinterpret_cast<const f32x4*>(ok ? G.T + ((int64_t)(pn[i] * Hb + y) * Wb + x) * cb + c0 : cv_zeros); \
            }                                                                                                            \
        } else if (MODE == CONV_UP) {                                                                                    \
            const int tap = kst & 3, s0 = (kst >> 2) * CV_BK + skq;                                                      \
            const int th = tap >> 1, tw = tap & 1;                                                                       \
            const int da = py ? (th ? 0 : 1) : (th ? -1 : 0), db = px ? (tw ? 0 : 1) : (tw ? -1 : 0);                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
                const int a = pa[i] + da, b = pb[i] + db;                                                                \
                const bool ok = pv[i] && (unsigned)a < (unsigned)Hs && (unsigned)b < (unsigned)Ws;                       \
                v[i] = *reinterpret_cast<const f32x4*>(ok ? G.S + ((int64_t)(pn[i] * Hs + a) * Ws + b) * cs + s0 : cv_zeros); \
            }                                                                                                            \
        } else if (MODE == MAT_NT) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
                v[i] = *reinterpret_cast<const f32x4*>(pv[i] ? G.S + (int64_t)(m0 + srow + 32 * i) * G.lda + kst * CV_BK + skq : cv_zeros); \
        } else {                                                                                                         \
            /* S as (pixels, cs): 4 consecutive reduction rows m, the thread's 4 output rows j */                        \
            const int j = m0 + 4 * a_rg;                                                                                 \
            const int64_t lds_ = MODE == MAT_TN ? G.lda : (int64_t)cs;                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
                const int m = (ks0 + kst) * CV_BK + a_kq * 4 + q;                                                        \
                const bool ok = a_on && m < pixels && j < cs;                                                            \
                v[q] = *reinterpret_cast<const f32x4*>(ok ? G.S + (int64_t)m * lds_ + j : cv_zeros);                     \
            }                                                                                                            \
        }                                                                                                                \
    }
#define CS_LOAD_B(v, ks_)                                                                                                \
    {                                                                                                                    \
        const int kst = min((ks_), nk_last);                                                                             \
        if (MODE == CONV_DOWN) {                                                                                         \
            const int t = kst / cpc, kh = ((t >> 3) & 1) + ((t >> 1) & 1) * 2, kw = ((t >> 2) & 1) + (t & 1) * 2;         \
            const int k = (kh * 4 + kw) * cb + (kst - t * cpc) * CV_BK + skq;                                            \
            _Pragma("unroll") for (int i = 0; i < NVB; ++i) {                                                            \
                const int j = n0 + srow + 32 * i;                                                                        \
                v[i] = *reinterpret_cast<const f32x4*>(j < cs ? G.Q + (int64_t)j * 16 * cb + k : cv_zeros);                  \
            }                                                                                                            \
        } else if (MODE == CONV_UP) {                                                                                    \
            const int tap = kst & 3, s0 = (kst >> 2) * CV_BK;                                                            \
            const int th = tap >> 1, tw = tap & 1;                                                                       \
            const int kh = py ? (th ? 2 : 0) : (th ? 3 : 1), kw = px ? (tw ? 2 : 0) : (tw ? 3 : 1);                      \
            const int c = n0 + 4 * b_rg;                                                                                 \
            const bool ok = b_on && c < cb;                                                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
                const int s = s0 + b_kq * 4 + q;                                                                         \
                v[q] = *reinterpret_cast<const f32x4*>(ok ? G.Q + ((int64_t)s * 16 + kh * 4 + kw) * cb + c : cv_zeros);      \
            }                                                                                                            \
        } else if (MODE == MAT_TN || MODE == MAT_NT) {                                                                   \
            /* B as (K, N): 4 consecutive reduction rows, the thread's 4 output columns */                               \
            const int c = n0 + 4 * b_rg;                                                                                 \
            const int kmax = MODE == MAT_TN ? pixels : cs;                                                               \
            const float* const bp = MODE == MAT_TN ? G.T : G.Q;                                                          \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
                const int kr = (ks0 + kst) * CV_BK + b_kq * 4 + q;                                                       \
                const bool ok = b_on && kr < kmax && c < cb;                                                             \
                v[q] = *reinterpret_cast<const f32x4*>(ok ? bp + (int64_t)kr * G.ldb + c : cv_zeros);                    \
            }                                                                                                            \
        } else {                                                                                                         \
            const int mb = (ks0 + kst) * CV_BK + b_kq * 4;                                                               \
            const int mm = mb < pixels ? mb : 0;                                                                         \
            int n = mm / (Hs * Ws);                                                                                      \
            const int r = mm - n * (Hs * Ws);                                                                            \
            int a = r / Ws, b = r - a * Ws;                                                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
                const int y = 2 * a - 1 + w_kh, x = 2 * b - 1 + w_kw;                                                    \
                const bool ok = b_on && w_nv && mb + q < pixels && (unsigned)y < (unsigned)Hb && (unsigned)x < (unsigned)Wb; \
                v[q] = *reinterpret_cast<const f32x4*>(ok ? G.T + ((int64_t)(n * Hb + y) * Wb + x) * cb + w_c : cv_zeros);   \
                if (++b == Ws) { b = 0; if (++a == Hs) { a = 0; ++n; } }                                                 \
            }                                                                                                            \
        }                                                                                                                \
    }
    // registers -> LDS, split on the way
#define CS_PUT_ROWK(S, KGS, PART, v, NV)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NV; ++i)                                                                       \
        cvs_putp<PARTS>(S + cvs_kgo(KGS, skq >> 3) + (srow + 32 * i) * 8 + (skq & 7), PART, v[i].x, v[i].y, v[i].z, v[i].w);
#define CS_PUT_KROW(S, KGS, PART, R, rg, kq, on, v)                                                                      \
    if (on) {                                                                                                            \
        __bf16* const d_ = S + cvs_kgo(KGS, kq >> 1) + rg * 8 + (kq & 1) * 4;                                            \
        cvs_putp<PARTS>(d_, PART, v[0].x, v[1].x, v[2].x, v[3].x);                                                               \
        cvs_putp<PARTS>(d_ + (R / 4) * 8, PART, v[0].y, v[1].y, v[2].y, v[3].y);                                                 \
        cvs_putp<PARTS>(d_ + 2 * (R / 4) * 8, PART, v[0].z, v[1].z, v[2].z, v[3].z);                                             \
        cvs_putp<PARTS>(d_ + 3 * (R / 4) * 8, PART, v[0].w, v[1].w, v[2].w, v[3].w);                                             \
    }
#define CS_FRAG(S, KGS, PART, part, slot, half) \
    (*reinterpret_cast<const cv_bf16x8*>(S + (part) * PART + cvs_kgo(KGS, 2 * (half) + lh) + (slot) * 8))

    f32x4 a0[4], a1[4], b0[NVB], b1[NVB];
    CS_LOAD_A(a0, 0) CS_LOAD_B(b0, 0)
    CS_LOAD_A(a1, 1) CS_LOAD_B(b1, 1)

    f32x16 acc[MT * NT];
#pragma unroll
    for (int i = 0; i < MT * NT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // one K-step from a register stage: LDS <- registers (split), then the requests for the step after next, then per
    // 16 k the six partial products of every accumulator, smallest first, consecutive MFMAs on different accumulators
#define CS_STEP(av, bv, ks_)                                                                                                       \
    {                                                                                                                    \
        if ((ks_) > 0) __syncthreads();          /* every wave has read the previous step's images */                    \
        if (A_ROWK) { CS_PUT_ROWK(sA, A_KGS, A_PART, av, 4) } else { CS_PUT_KROW(sA, A_KGS, A_PART, BM, a_rg, a_kq, a_on, av) } \
        if (B_ROWK) { CS_PUT_ROWK(sB, B_KGS, B_PART, bv, NVB) } else { CS_PUT_KROW(sB, B_KGS, B_PART, BN, b_rg, b_kq, b_on, bv) } \
        __syncthreads();                                                                                                 \
        CS_LOAD_A(av, (ks_) + 2) CS_LOAD_B(bv, (ks_) + 2)                                                                          \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                         \
            cv_bf16x8 fa[MT][PARTS], fb[NT][PARTS];                                                                      \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                               \
                _Pragma("unroll") for (int p = 0; p < PARTS; ++p) fa[i][p] = CS_FRAG(sA, A_KGS, A_PART, p, wm * 32 * MT + 32 * i + li, half); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                               \
                _Pragma("unroll") for (int p = 0; p < PARTS; ++p) fb[j][p] = CS_FRAG(sB, B_KGS, B_PART, p, wn * 32 * NT + 32 * j + li, half); \
            _Pragma("unroll") for (int t = (PARTS == 3 ? 0 : 5); t < 6; ++t) {                                           \
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};                                    \
                _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
                    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
                        acc[i * NT + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[t]], fb[j][PB[t]], acc[i * NT + j], 0, 0, 0); \
            }                                                                                                            \
        }                                                                                                                \
    }

    for (int ks = 0; ks < nk; ks += 2) {
        CS_STEP(a0, b0, ks)
        if (ks + 1 < nk) CS_STEP(a1, b1, ks + 1)
    }
#undef CS_LOAD_A
#undef CS_LOAD_B
#undef CS_PUT_ROWK
#undef CS_PUT_KROW
#undef CS_FRAG
#undef CS_STEP

    // ---- epilogue through LDS: MFMA rows back to tile rows, then every thread owns 4 consecutive columns of a row
    __syncthreads();
    float (*sC)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(smem_raw);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cslot = wn * 32 * NT + 32 * j + li;
            const int ccol = B_ROWK ? cslot : 4 * (cslot % (BN / 4)) + cslot / (BN / 4);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rslot = wm * 32 * MT + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int rrow = A_ROWK ? rslot : 4 * (rslot % (BM / 4)) + rslot / (BM / 4);
                sC[rrow][ccol] = acc[i * NT + j][e];
            }
        }
    __syncthreads();
    constexpr int TPR = BN / 4;                 // threads per output row
    const int c4 = (tid % TPR) * 4, col = n0 + c4;
    const int ncols = MODE == CONV_DOWN ? cs : (MODE == CONV_WGRAD ? 16 * cb : cb);
    const int nrows = (MODE == CONV_WGRAD || MODE == MAT_TN) ? cs : pixels;
    if (col >= ncols) return;
    for (int r = tid / TPR; r < BM; r += 256 / TPR) {
        const int m = m0 + r;
        if (m >= nrows) break;
        const f32x4 y = *reinterpret_cast<const f32x4*>(&sC[r][c4]);
        float* o;
        if (MODE == CONV_DOWN) o = G.out + (int64_t)m * cs + col;
        else if (MODE == CONV_UP) {
            const int n = m / (Hs * Ws), rr = m - n * (Hs * Ws), a = rr / Ws, b = rr - a * Ws;
            o = G.out + ((int64_t)(n * Hb + 2 * a + py) * Wb + 2 * b + px) * cb + col;
        } else if (MODE == MAT_NT) o = G.out + (int64_t)m * G.ldc + col;
        else if (MODE == MAT_TN) o = G.partial ? G.partial + ((int64_t)blockIdx.z * cs + m) * cb + col : G.out + (int64_t)m * G.ldc + col;
        else {
            float* base = G.partial ? G.partial + (int64_t)blockIdx.z * cs * 16 * cb : G.out;
            o = base + (int64_t)m * 16 * cb + col;
        }
        *reinterpret_cast<f32x4*>(o) = y;      // ncols % 4 == 0
    }
}

// UP for outputs of at most 32 channels, split form: the four parity classes of a 128-pixel tile in one workgroup.
// A class-by-class launch reads every small-resolution pixel 16 times (4 classes x 4 taps) for 32 output columns each,
// too few MFMAs per staged value; here the reduction runs over the 9 neighbours (da, db) of the small pixel, each
// staged (and split) once and multiplied into the 1, 2 or 4 classes that meet it: class (py, px) meets neighbour
// da in {0, py ? +1 : -1} through kh = 1 + py - 2 da (columns alike).  Wave w owns rows 32 w ... of the tile and one
// 32 x 32 accumulator per class, and stages the weights of class w (32 k x 32 c per step, transposed into
// [k / 8][slot][8 k] like every [k][col] operand of k_conv_split).  Outputs go from the accumulators straight to
// memory: the 32 lanes of an accumulator row hold the 32 channels of one output pixel, one 128-byte line.
__global__ __launch_bounds__(256) void k_conv_up4_split(const ConvArgs G) {
    constexpr int A_KGS = 128, A_PART = cvs_part(128);
    constexpr int B_KGS = 32, B_PART = cvs_part(32), B_CLASS = 3 * B_PART;
    __shared__ __attribute__((aligned(16))) __bf16 smem[3 * A_PART + 4 * B_CLASS];
    __shared__ int s_pix[128];
    __bf16* const sA = smem;
    __bf16* const sB = smem + 3 * A_PART;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int Hs = G.Hs, Ws = G.Ws, cs = G.cs, cb = G.cb, Hb = 2 * Hs, Wb = 2 * Ws;
    const int pixels = G.NI * Hs * Ws;
    const int m0 = blockIdx.x * 128;
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    const int b_rg = lane & 7, b_kq = lane >> 3;        // this thread's 4 x 4 block of its wave's class: channels 4 b_rg ..., k quad b_kq
    const int upy = wave >> 1, upx = wave & 1;
    const int cpk = cs / CV_BK, nk = 9 * cpk;

    int pn[4], pa[4], pb[4];
    bool pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        pv[i] = m < pixels;
        const int mm = pv[i] ? m : 0;
        pn[i] = mm / (Hs * Ws);
        const int r = mm - pn[i] * (Hs * Ws);
        pa[i] = r / Ws;
        pb[i] = r - pa[i] * Ws;
    }

#define U4_NEIGHBOUR(kst)                                                                                                \
    const int nb = (kst) % 9, s0 = ((kst) / 9) * CV_BK;                                                                  \
    const int da = nb / 3 - 1, db = nb - (nb / 3) * 3 - 1;
#define U4_MEETS(py_, px_) ((da == 0 || da == ((py_) ? 1 : -1)) && (db == 0 || db == ((px_) ? 1 : -1)))
    /* unconditional loads, as in k_conv_split */                                                                        \
#define U4_LOAD(av, bv, ks_)                                                                                             \
    {                                                                                                                    \
        U4_NEIGHBOUR(min((ks_), nk - 1))                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
            const int a = pa[i] + da, b = pb[i] + db;                                                                    \
            const bool ok = pv[i] && (unsigned)a < (unsigned)Hs && (unsigned)b < (unsigned)Ws;                           \
            av[i] = *reinterpret_cast<const f32x4*>(ok ? G.S + ((int64_t)(pn[i] * Hs + a) * Ws + b) * cs + s0 + skq : cv_zeros); \
        }                                                                                                                \
        const bool use = U4_MEETS(upy, upx) && 4 * b_rg < cb;                                                            \
        const int kh = 1 + upy - 2 * da, kw = 1 + upx - 2 * db;                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                    \
            bv[q] = *reinterpret_cast<const f32x4*>(use ? G.Q + ((int64_t)(s0 + 4 * b_kq + q) * 16 + kh * 4 + kw) * cb + 4 * b_rg : cv_zeros); \
    }
#define U4_FRAG(S, KGS, PART, part, slot, half) \
    (*reinterpret_cast<const cv_bf16x8*>(S + (part) * PART + cvs_kgo(KGS, 2 * (half) + lh) + (slot) * 8))
#define U4_CLASS(v, half)                                                                                                \
    if (U4_MEETS((v) >> 1, (v) & 1)) {                                                                                   \
        const cv_bf16x8 bh = U4_FRAG(sB + (v) * B_CLASS, B_KGS, B_PART, 0, li, half),                                    \
                        bm = U4_FRAG(sB + (v) * B_CLASS, B_KGS, B_PART, 1, li, half),                                    \
                        bl = U4_FRAG(sB + (v) * B_CLASS, B_KGS, B_PART, 2, li, half);                                    \
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[v], 0, 0, 0);                                       \
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[v], 0, 0, 0);                                       \
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[v], 0, 0, 0);                                       \
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[v], 0, 0, 0);                                       \
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[v], 0, 0, 0);                                       \
        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[v], 0, 0, 0);                                       \
    }
#define U4_STEP(av, bv, ks_)                                                                                                 \
    {                                                                                                                    \
        U4_NEIGHBOUR(ks_)                                                                                                \
        (void)s0;                                                                                                        \
        if ((ks_) > 0) __syncthreads();                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
            cvs_put(sA + cvs_kgo(A_KGS, skq >> 3) + (srow + 32 * i) * 8 + (skq & 7), A_PART, av[i].x, av[i].y, av[i].z, av[i].w); \
                                                                                                                         \
        if (U4_MEETS(upy, upx)) {                                                                                        \
            __bf16* const d_ = sB + wave * B_CLASS + cvs_kgo(B_KGS, b_kq >> 1) + b_rg * 8 + (b_kq & 1) * 4;               \
            cvs_put(d_, B_PART, bv[0].x, bv[1].x, bv[2].x, bv[3].x);                                                     \
            cvs_put(d_ + 64, B_PART, bv[0].y, bv[1].y, bv[2].y, bv[3].y);                                                \
            cvs_put(d_ + 128, B_PART, bv[0].z, bv[1].z, bv[2].z, bv[3].z);                                               \
            cvs_put(d_ + 192, B_PART, bv[0].w, bv[1].w, bv[2].w, bv[3].w);                                               \
        }                                                                                                                \
        __syncthreads();                                                                                                 \
        U4_LOAD(av, bv, (ks_) + 2)                                                                                           \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                         \
            const cv_bf16x8 ah = U4_FRAG(sA, A_KGS, A_PART, 0, wave * 32 + li, half),                                    \
                            am = U4_FRAG(sA, A_KGS, A_PART, 1, wave * 32 + li, half),                                    \
                            al = U4_FRAG(sA, A_KGS, A_PART, 2, wave * 32 + li, half);                                    \
            U4_CLASS(0, half) U4_CLASS(1, half) U4_CLASS(2, half) U4_CLASS(3, half)                                      \
        }                                                                                                                \
    }

    f32x4 a0[4], a1[4], b0[4], b1[4];
    U4_LOAD(a0, b0, 0)
    U4_LOAD(a1, b1, 1)
    f32x16 acc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;

    for (int ks = 0; ks < nk; ks += 2) {       // nk = 9 cs / 32 >= 9; an odd count ends on the first stage
        U4_STEP(a0, b0, ks)
        if (ks + 1 < nk) U4_STEP(a1, b1, ks + 1)
    }
#undef U4_NEIGHBOUR
#undef U4_MEETS
#undef U4_LOAD
#undef U4_FRAG
#undef U4_CLASS
#undef U4_STEP

    // ---- outputs: big-resolution pixel (2a + py, 2b + px) of every tile row, channel 4 (li % 8) + li / 8 in lane li
    if (tid < 128) {
        const int m = m0 + tid;
        int p = -1;
        if (m < pixels) {
            const int n = m / (Hs * Ws), r = m - n * (Hs * Ws), a = r / Ws, b = r - a * Ws;
            p = (n * Hb + 2 * a) * Wb + 2 * b;
        }
        s_pix[tid] = p;
    }
    __syncthreads();
    const int col = 4 * (li & 7) + (li >> 3);
    if (col >= cb) return;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int shift = (v >> 1) * Wb + (v & 1);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int p = s_pix[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh];
            if (p >= 0) G.out[(int64_t)(p + shift) * cb + col] = acc[v][e];
        }
    }
}

// 128 x 128 output tiles (waves of 64 x 64) where the channel counts fill them: a third fewer operand bytes through the
// vector caches per product than 128 x 64, and those bytes, not the matrix pipe, are what these kernels wait for
static bool conv_wide_tiles(int channels) {
    return channels % 128 == 0 && std::getenv("DRM_CONV_NO_WIDE_TILES") == nullptr;
}

// the split form is the default; DRM_NO_SPLIT_PRODUCTS=1 (the switch of the large Linear products) keeps the fp32 MFMA form
static bool conv_split_form() {
    const char* e = std::getenv("DRM_NO_SPLIT_PRODUCTS");
    return !(e && e[0] == '1');
}

static int wgrad_splits(int64_t pixels, int cs, int cb) {
    const int rows = conv_wide_tiles(cs) ? 128 : 64;
    const int64_t tiles = (int64_t)((cs + rows - 1) / rows) * ((16 * cb + 127) / 128);
    const int64_t nk = (pixels + CV_BK - 1) / CV_BK;
    // two workgroups per CU; the split form has one LDS stage, so a workgroup alone on a CU cannot hide its staging
    // behind another's MFMAs and the few-tile layers take up to 128 slices there (236 against 331 us for 64 x 512 outputs)
    const int cap = conv_split_form() ? 128 : 64;
    int s = 1;
    while (tiles * s < 512 && s * 2 <= cap && nk / (s * 2) >= 8) s *= 2;
    return s;
}

}  // namespace drm

using namespace drm;

static int check_conv(const void* a, const void* b, const void* out, int64_t NI, int32_t Hs, int32_t Ws, int32_t cs, int32_t cb) {
    DRM_REQUIRE(a && b && out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(NI >= 1 && Hs >= 1 && Ws >= 1 && NI * Hs * Ws < (1ll << 31) / 4, DRM_EINVAL, "images=%lld, %d x %d", (long long)NI, Hs, Ws);
    DRM_REQUIRE(cs >= 4 && cb >= 4 && cs % 4 == 0 && cb % 4 == 0, DRM_ENOTSUP, "channel counts %d, %d must be multiples of 4", cs, cb);
    DRM_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out), DRM_EINVAL, "tensors must be 16-byte aligned");
    return DRM_OK;
}

extern "C" int drm_conv4s2_down_prec(const float* big, const float* kernel, float* small_out, int64_t NI, int32_t Hs, int32_t Ws,
                                     int32_t cs, int32_t cb, int32_t precision, drm_stream_t stream);
extern "C" int drm_conv4s2_down(const float* big, const float* kernel, float* small_out, int64_t NI, int32_t Hs, int32_t Ws,
                                int32_t cs, int32_t cb, drm_stream_t stream) {
    return drm_conv4s2_down_prec(big, kernel, small_out, NI, Hs, Ws, cs, cb, 0, stream);
}

// precision 0: fp32 results (split form, or the fp32 MFMA form under DRM_NO_SPLIT_PRODUCTS); 1: operands rounded to bf16
extern "C" int drm_conv4s2_down_prec(const float* big, const float* kernel, float* small_out, int64_t NI, int32_t Hs, int32_t Ws,
                                     int32_t cs, int32_t cb, int32_t precision, drm_stream_t stream) {
    if (int rc = check_conv(big, kernel, small_out, NI, Hs, Ws, cs, cb)) return rc;
    DRM_REQUIRE(cb % 32 == 0, DRM_ENOTSUP, "big-resolution channel count %d must be a multiple of 32", cb);
    ConvArgs g{big, nullptr, kernel, small_out, (int)NI, Hs, Ws, cs, cb, 0, nullptr};
    const int64_t pixels = NI * Hs * Ws;
    ProfScope prof(PC_LINEAR_BIG, (hipStream_t)stream, 2.0 * pixels * cs * 16 * cb, 4.0 * (4.0 * pixels * cb + 16.0 * cs * cb + (double)pixels * cs));
    const dim3 wide(cs / 128, (unsigned)((pixels + 127) / 128), 1), narrow((cs + 63) / 64, (unsigned)((pixels + 127) / 128), 1);
    if (precision == 1) {
        if (conv_wide_tiles(cs)) k_conv_split<CONV_DOWN, 2, 2, 2, 1><<<wide, dim3(256), 0, (hipStream_t)stream>>>(g);
        else k_conv_split<CONV_DOWN, 2, 1, 2, 1><<<narrow, dim3(256), 0, (hipStream_t)stream>>>(g);
    } else if (conv_split_form()) {
        if (conv_wide_tiles(cs)) k_conv_split<CONV_DOWN, 2, 2><<<wide, dim3(256), 0, (hipStream_t)stream>>>(g);
        else k_conv_split<CONV_DOWN, 2, 1><<<narrow, dim3(256), 0, (hipStream_t)stream>>>(g);
    } else if (conv_wide_tiles(cs))
        k_conv_gemm<CONV_DOWN, 2, 2><<<wide, dim3(256), 0, (hipStream_t)stream>>>(g);
    else
        k_conv_gemm<CONV_DOWN, 2, 1><<<narrow, dim3(256), 0, (hipStream_t)stream>>>(g);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_conv4s2_up_prec(const float* small, const float* kernel, float* big_out, int64_t NI, int32_t Hs, int32_t Ws,
                                   int32_t cs, int32_t cb, int32_t precision, drm_stream_t stream);
extern "C" int drm_conv4s2_up(const float* small, const float* kernel, float* big_out, int64_t NI, int32_t Hs, int32_t Ws,
                              int32_t cs, int32_t cb, drm_stream_t stream) {
    return drm_conv4s2_up_prec(small, kernel, big_out, NI, Hs, Ws, cs, cb, 0, stream);
}

extern "C" int drm_conv4s2_up_prec(const float* small, const float* kernel, float* big_out, int64_t NI, int32_t Hs, int32_t Ws,
                                   int32_t cs, int32_t cb, int32_t precision, drm_stream_t stream) {
    if (int rc = check_conv(small, kernel, big_out, NI, Hs, Ws, cs, cb)) return rc;
    DRM_REQUIRE(cs % 32 == 0, DRM_ENOTSUP, "small-resolution channel count %d must be a multiple of 32", cs);
    ConvArgs g{nullptr, small, kernel, big_out, (int)NI, Hs, Ws, cs, cb, 0, nullptr};
    const int64_t pixels = NI * Hs * Ws;
    ProfScope prof(PC_LINEAR_BIG, (hipStream_t)stream, 2.0 * pixels * cs * 16 * cb, 4.0 * (4.0 * pixels * cb + 16.0 * cs * cb + (double)pixels * cs));
    const unsigned mtiles = (unsigned)((pixels + 127) / 128);
    // outputs of 32 columns: all four parity classes per workgroup (class by class a 128 x 32 tile has too few MFMAs per
    // staged value to pay for splitting it: 266 against 252 us of the fp32 form)
    if (precision == 1 && cb <= 32)                // rounding to bf16 costs a tenth of the exact split: class by class, 128 x 32 tiles
        k_conv_split<CONV_UP, 1, 1, 1, 1><<<dim3(1, mtiles, 4), dim3(256), 0, (hipStream_t)stream>>>(g);
    else if (precision == 1) {
        if (conv_wide_tiles(cb)) k_conv_split<CONV_UP, 2, 2, 2, 1><<<dim3(cb / 128, mtiles, 4), dim3(256), 0, (hipStream_t)stream>>>(g);
        else k_conv_split<CONV_UP, 2, 1, 2, 1><<<dim3((cb + 63) / 64, mtiles, 4), dim3(256), 0, (hipStream_t)stream>>>(g);
    } else if (conv_split_form() && cb <= 32 && std::getenv("DRM_CONV_NO_UP4") == nullptr)
        k_conv_up4_split<<<dim3(mtiles), dim3(256), 0, (hipStream_t)stream>>>(g);
    else if (conv_split_form() && cb > 32) {
        if (conv_wide_tiles(cb)) k_conv_split<CONV_UP, 2, 2><<<dim3(cb / 128, mtiles, 4), dim3(256), 0, (hipStream_t)stream>>>(g);
        else k_conv_split<CONV_UP, 2, 1><<<dim3((cb + 63) / 64, mtiles, 4), dim3(256), 0, (hipStream_t)stream>>>(g);
    } else if (cb <= 32 && std::getenv("DRM_CONV_NO_TALL_TILE") == nullptr)   // outputs of 32 columns: waves 4 x 1 over a 128 x 32 tile instead of half-empty 64-column tiles
        // (a 256 x 32 tile moves fewer operand bytes per product but its 92 KB of LDS leave one workgroup per CU: 312 against 256 us)
        k_conv_gemm<CONV_UP, 1, 1, 1><<<dim3(1, (unsigned)((pixels + 127) / 128), 4), dim3(256), 0, (hipStream_t)stream>>>(g);
    else if (conv_wide_tiles(cb))
        k_conv_gemm<CONV_UP, 2, 2><<<dim3(cb / 128, (unsigned)((pixels + 127) / 128), 4), dim3(256), 0, (hipStream_t)stream>>>(g);
    else
        k_conv_gemm<CONV_UP, 2, 1><<<dim3((cb + 63) / 64, (unsigned)((pixels + 127) / 128), 4), dim3(256), 0, (hipStream_t)stream>>>(g);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" size_t drm_conv4s2_wgrad_workspace_bytes(int64_t NI, int32_t Hs, int32_t Ws, int32_t cs, int32_t cb) {
    if (NI < 1 || Hs < 1 || Ws < 1 || cs < 1 || cb < 1) return 0;
    const int s = wgrad_splits(NI * Hs * Ws, cs, cb);
    return s > 1 ? (size_t)s * cs * 16 * cb * sizeof(float) : 0;
}

extern "C" int drm_conv4s2_wgrad_prec(const float* small, const float* big, float* kernel_grad_out, int64_t NI, int32_t Hs, int32_t Ws,
                                      int32_t cs, int32_t cb, void* workspace, size_t workspace_bytes, int32_t precision,
                                      drm_stream_t stream);
extern "C" int drm_conv4s2_wgrad(const float* small, const float* big, float* kernel_grad_out, int64_t NI, int32_t Hs, int32_t Ws,
                                 int32_t cs, int32_t cb, void* workspace, size_t workspace_bytes, drm_stream_t stream) {
    return drm_conv4s2_wgrad_prec(small, big, kernel_grad_out, NI, Hs, Ws, cs, cb, workspace, workspace_bytes, 0, stream);
}

extern "C" int drm_conv4s2_wgrad_prec(const float* small, const float* big, float* kernel_grad_out, int64_t NI, int32_t Hs, int32_t Ws,
                                      int32_t cs, int32_t cb, void* workspace, size_t workspace_bytes, int32_t precision,
                                      drm_stream_t stream) {
    if (int rc = check_conv(small, big, kernel_grad_out, NI, Hs, Ws, cs, cb)) return rc;
    const int64_t pixels = NI * Hs * Ws;
    const int splits = wgrad_splits(pixels, cs, cb);
    const int nk = (int)((pixels + CV_BK - 1) / CV_BK);
    ConvArgs g{big, small, nullptr, kernel_grad_out, (int)NI, Hs, Ws, cs, cb, (nk + splits - 1) / splits, nullptr};
    if (splits > 1) {
        const size_t need = (size_t)splits * cs * 16 * cb * sizeof(float);
        DRM_REQUIRE(workspace && workspace_bytes >= need && aligned16(workspace), DRM_EINVAL, "workspace too small: need %zu bytes", need);
        g.partial = (float*)workspace;
    }
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof(PC_LINEAR_BIG, s, 2.0 * pixels * cs * 16 * cb, 4.0 * (4.0 * pixels * cb + 16.0 * cs * cb + (double)pixels * cs));
        if (precision == 1 && conv_split_form()) {       // (same slices as the split form: the workspace size does not depend on the precision)
            if (conv_wide_tiles(cs)) k_conv_split<CONV_WGRAD, 2, 2, 2, 1><<<dim3((16 * cb + 127) / 128, cs / 128, splits), dim3(256), 0, s>>>(g);
            else k_conv_split<CONV_WGRAD, 1, 2, 2, 1><<<dim3((16 * cb + 127) / 128, (cs + 63) / 64, splits), dim3(256), 0, s>>>(g);
        } else if (conv_split_form()) {
            if (conv_wide_tiles(cs)) k_conv_split<CONV_WGRAD, 2, 2><<<dim3((16 * cb + 127) / 128, cs / 128, splits), dim3(256), 0, s>>>(g);
            else k_conv_split<CONV_WGRAD, 1, 2><<<dim3((16 * cb + 127) / 128, (cs + 63) / 64, splits), dim3(256), 0, s>>>(g);
        } else if (conv_wide_tiles(cs)) k_conv_gemm<CONV_WGRAD, 2, 2><<<dim3((16 * cb + 127) / 128, cs / 128, splits), dim3(256), 0, s>>>(g);
        else k_conv_gemm<CONV_WGRAD, 1, 2><<<dim3((16 * cb + 127) / 128, (cs + 63) / 64, splits), dim3(256), 0, s>>>(g);
        DRM_LAUNCH_CHECK();
    }
    if (splits > 1) {
        const int64_t total4 = (int64_t)cs * 16 * cb / 4;
        ProfScope prof(PC_ELEMENTWISE, s, (double)splits * total4 * 4, 16.0 * (splits + 1.0) * total4);
        k_conv_reduce<<<dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, s>>>(g.partial, splits, total4, kernel_grad_out);
        DRM_LAUNCH_CHECK();
    }
    return DRM_OK;
}

// ---- large Linear gradients on the split kernel's staging ----------------------------------------------------------------
static int mat_tn_splits(int M, int N, int K) {
    const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    const int nk = (K + CV_BK - 1) / CV_BK;
    int s = 1;
    while (tiles * s < 256 && s * 2 <= 64 && nk / (s * 2) >= 8) s *= 2;
    return s;
}

extern "C" size_t drm_gemm_split_workspace_bytes(int32_t trans_a, int32_t M, int32_t N, int32_t K) {
    if (!trans_a || M < 1 || N < 1 || K < 1) return 0;
    const int s = mat_tn_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int drm_gemm_split_f32(int32_t trans_a, int32_t bf16_inputs, int32_t M, int32_t N, int32_t K, const float* A, int64_t lda,
                                  const float* B, int64_t ldb, float* C, int64_t ldc, void* workspace, size_t workspace_bytes,
                                  drm_stream_t stream) {
    DRM_REQUIRE(A && B && C, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(M >= 1 && N >= 4 && K >= 1 && N % 4 == 0 && ldb >= N && ldb % 4 == 0 && ldc >= N && ldc % 4 == 0, DRM_EINVAL,
                "M=%d N=%d K=%d ldb=%lld ldc=%lld (N and the leading dimensions multiples of 4)", M, N, K, (long long)ldb, (long long)ldc);
    DRM_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), DRM_EINVAL, "operands must be 16-byte aligned");
    DRM_REQUIRE((int64_t)M * N < (1ll << 31) && (int64_t)K * (M > N ? M : N) < (1ll << 40), DRM_EINVAL, "extents too large");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PC_LINEAR_BIG, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    if (trans_a) {                                  // C = A^T B, A (K, M)
        DRM_REQUIRE(M % 4 == 0 && lda >= M && lda % 4 == 0, DRM_EINVAL, "M=%d and lda=%lld must be multiples of 4", M, (long long)lda);
        const int splits = mat_tn_splits(M, N, K);
        const int nk = (K + CV_BK - 1) / CV_BK;
        ConvArgs g{B, A, nullptr, C, K, 1, 1, M, N, (nk + splits - 1) / splits, nullptr};
        g.lda = lda; g.ldb = ldb; g.ldc = ldc;
        if (splits > 1) {
            const size_t need = (size_t)splits * M * N * sizeof(float);
            DRM_REQUIRE(workspace && workspace_bytes >= need && aligned16(workspace), DRM_EINVAL, "workspace too small: need %zu bytes", need);
            DRM_REQUIRE(ldc == N, DRM_EINVAL, "a split reduction writes a contiguous result (ldc = N)");
            g.partial = (float*)workspace;
        }
        if (bf16_inputs) k_conv_split<MAT_TN, 2, 2, 2, 1><<<dim3((N + 127) / 128, (M + 127) / 128, splits), dim3(256), 0, s>>>(g);
        else k_conv_split<MAT_TN, 2, 2><<<dim3((N + 127) / 128, (M + 127) / 128, splits), dim3(256), 0, s>>>(g);
        DRM_LAUNCH_CHECK();
        if (splits > 1) {
            const int64_t total4 = (int64_t)M * N / 4;
            k_conv_reduce<<<dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, s>>>(g.partial, splits, total4, C);
            DRM_LAUNCH_CHECK();
        }
    } else {                                        // C = A B, A (M, K)
        DRM_REQUIRE(K % CV_BK == 0 && lda >= K && lda % 4 == 0, DRM_EINVAL, "K=%d must be a multiple of 32, lda=%lld of 4", K, (long long)lda);
        ConvArgs g{nullptr, A, B, C, M, 1, 1, K, N, 0, nullptr};
        g.lda = lda; g.ldb = ldb; g.ldc = ldc;
        if (bf16_inputs) k_conv_split<MAT_NT, 2, 2, 2, 1><<<dim3((N + 127) / 128, (M + 127) / 128, 1), dim3(256), 0, s>>>(g);
        else k_conv_split<MAT_NT, 2, 2><<<dim3((N + 127) / 128, (M + 127) / 128, 1), dim3(256), 0, s>>>(g);
        DRM_LAUNCH_CHECK();
    }
    return DRM_OK;
}
