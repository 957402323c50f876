// General fp32 matrix product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) for the batched (B*T)-row side
// of the world-model update: the Linear layers of the prior / reward / continue / decoder / encoder heads
// (lib/world_model.py:129-134, :153-160, :203-217, :247-261, :283-288), their input and weight gradients, and the
// weight-gradient products of the observe recurrence (lib/world_model.py:39-71).
//
//   C (M, N) = op(A) . op(B) (+ bias[n])
//   TA = 0: A is (M, K), k contiguous         TA = 1: A is (K, M), m contiguous   (a gradient read as dY^T)
//   TB = 0: B is (N, K), k contiguous (torch Linear weight)      TB = 1: B is (K, N), n contiguous
// so forward = (0, 0), input gradient dX = dY . W = (0, 1), weight gradient dW = dY^T . X = (1, 1).
//
// Workgroup = 4 waves = a 64 x 64 output tile (2 x 2 waves of one 32 x 32 accumulator), K in steps of 32 through two
// LDS buffers with two register stages of global loads in flight.  Operands whose reduction index is contiguous in
// memory are staged as [row][k] (row stride 36 words: conflict-free 16-byte fragment reads); operands whose reduction
// index is the slow one are staged as [k][row] (row stride 68 words) and read one word per lane and MFMA (the lanes of
// a half-wave read 32 consecutive words).  Problems with few output tiles split K over blockIdx.z into partial tiles
// in a workspace, summed in a fixed order by a second kernel (no atomics: results do not depend on timing).
#include "common.hpp"
#include <cstdlib>
#include "../../include/dreamer_amd.h"

namespace drm {

constexpr int GM_BM = 64, GM_BN = 64, GM_BK = 32, GM_LDK = 36, GM_LDT = 68;
// words per operand stage: the larger of the two images (64 rows x 36, 32 k x 68)
constexpr int GM_STAGE_WORDS = 64 * GM_LDK > GM_BK * GM_LDT ? 64 * GM_LDK : GM_BK * GM_LDT;
static_assert(4 * GM_STAGE_WORDS >= GM_BM * (GM_BN + 4), "the output tile must fit the operand stages");

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* bias;
    float* C; int64_t ldc;
    int M, N, K;
    int ksteps_per_split;      // K-steps (of 32) per blockIdx.z
    float* partial;            // (splits, M, N) when splits > 1
};

template <bool T>
struct Stage {
    // global -> registers: two f32x4 per thread per operand and K-step
    f32x4 v[2];
};

// loads one 64-row x 32-k operand tile slice for this thread
template <bool T>
__device__ __forceinline__ void gm_load(Stage<T>& s, const float* __restrict__ P, int64_t ld, int row0, int rows, int k0, int K,
                                        int tid) {
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!T) {
        // [row][k]: 8 threads cover the 32 k of a row; rows tid>>3 and (tid>>3) + 32
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = row0 + (tid >> 3) + 32 * i;
            s.v[i] = (r < rows && k0 + kq < K) ? *reinterpret_cast<const f32x4*>(P + (int64_t)r * ld + k0 + kq) : zero;
        }
    } else {
        // [k][row]: 16 threads cover the 64 rows of one k; k = tid>>4 and (tid>>4) + 16
        const int r4 = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + (tid >> 4) + 16 * i;
            const int r = row0 + r4;
            if (k < K && r + 3 < rows) {
                s.v[i] = *reinterpret_cast<const f32x4*>(P + (int64_t)k * ld + r);
            } else if (k < K && r < rows) {
                const float* p = P + (int64_t)k * ld + r;
                s.v[i] = f32x4{p[0], r + 1 < rows ? p[1] : 0.f, r + 2 < rows ? p[2] : 0.f, 0.f};
            } else {
                s.v[i] = zero;
            }
        }
    }
}

template <bool T>
__device__ __forceinline__ void gm_store(const Stage<T>& s, float* __restrict__ buf, int tid) {
    if constexpr (!T) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(buf + ((tid >> 3) + 32 * i) * GM_LDK + kq) = s.v[i];
    } else {
        const int r4 = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(buf + ((tid >> 4) + 16 * i) * GM_LDT + r4) = s.v[i];
    }
}

// fragment of group g (k = g*8 + lh*4 + s, s = 0..3) for the row `row` of a 64-row operand tile
template <bool T>
__device__ __forceinline__ f32x4 gm_frag(const float* __restrict__ buf, int row, int g, int lh) {
    if constexpr (!T) {
        return *reinterpret_cast<const f32x4*>(buf + row * GM_LDK + g * 8 + lh * 4);
    } else {
        const float* p = buf + (g * 8 + lh * 4) * GM_LDT + row;
        return f32x4{p[0], p[GM_LDT], p[2 * GM_LDT], p[3 * GM_LDT]};
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_f32(const GemmArgs G) {
    __shared__ __attribute__((aligned(16))) float smem[4 * GM_STAGE_WORDS];   // A0 A1 B0 B1; later the 64 x 68 output tile
    float* sA[2] = {smem, smem + GM_STAGE_WORDS};
    float* sB[2] = {smem + 2 * GM_STAGE_WORDS, smem + 3 * GM_STAGE_WORDS};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    const int nk_all = (G.K + GM_BK - 1) / GM_BK;
    const int ks0 = blockIdx.z * G.ksteps_per_split;
    const int ks1 = min(ks0 + G.ksteps_per_split, nk_all);
    const int nk = ks1 - ks0;

    Stage<TA> a0, a1;
    Stage<TB> b0, b1;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    a1.v[0] = a1.v[1] = b1.v[0] = b1.v[1] = zero;
    a0.v[0] = a0.v[1] = b0.v[0] = b0.v[1] = zero;
    if (nk > 0) { gm_load<TA>(a0, G.A, G.lda, m0, G.M, ks0 * GM_BK, G.K, tid); gm_load<TB>(b0, G.B, G.ldb, n0, G.N, ks0 * GM_BK, G.K, tid); }
    if (nk > 1) { gm_load<TA>(a1, G.A, G.lda, m0, G.M, (ks0 + 1) * GM_BK, G.K, tid); gm_load<TB>(b1, G.B, G.ldb, n0, G.N, (ks0 + 1) * GM_BK, G.K, tid); }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

#define GM_STEP(buf)                                                                                  \
    {                                                                                                 \
        f32x4 fa0 = gm_frag<TA>(sA[buf], wm * 32 + li, 0, lh), fb0 = gm_frag<TB>(sB[buf], wn * 32 + li, 0, lh); \
        f32x4 fa1 = gm_frag<TA>(sA[buf], wm * 32 + li, 1, lh), fb1 = gm_frag<TB>(sB[buf], wn * 32 + li, 1, lh); \
        _Pragma("unroll") for (int g = 0; g < 4; g += 2) {                                            \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.x, fb0.x, acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.y, fb0.y, acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.z, fb0.z, acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.w, fb0.w, acc, 0, 0, 0);                   \
            if (g + 2 < 4) { fa0 = gm_frag<TA>(sA[buf], wm * 32 + li, g + 2, lh); fb0 = gm_frag<TB>(sB[buf], wn * 32 + li, g + 2, lh); } \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.x, fb1.x, acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.y, fb1.y, acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.z, fb1.z, acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.w, fb1.w, acc, 0, 0, 0);                   \
            if (g + 3 < 4) { fa1 = gm_frag<TA>(sA[buf], wm * 32 + li, g + 3, lh); fb1 = gm_frag<TB>(sB[buf], wn * 32 + li, g + 3, lh); } \
        }                                                                                             \
    }

    for (int ks = 0; ks < nk; ks += 2) {
        gm_store<TA>(a0, sA[0], tid); gm_store<TB>(b0, sB[0], tid);
        __syncthreads();
        if (ks + 2 < nk) { gm_load<TA>(a0, G.A, G.lda, m0, G.M, (ks0 + ks + 2) * GM_BK, G.K, tid); gm_load<TB>(b0, G.B, G.ldb, n0, G.N, (ks0 + ks + 2) * GM_BK, G.K, tid); }
        GM_STEP(0)
        if (ks + 1 < nk) {
            gm_store<TA>(a1, sA[1], tid); gm_store<TB>(b1, sB[1], tid);
            __syncthreads();
            if (ks + 3 < nk) { gm_load<TA>(a1, G.A, G.lda, m0, G.M, (ks0 + ks + 3) * GM_BK, G.K, tid); gm_load<TB>(b1, G.B, G.ldb, n0, G.N, (ks0 + ks + 3) * GM_BK, G.K, tid); }
            GM_STEP(1)
        }
    }
#undef GM_STEP

    // epilogue through LDS: every thread then owns 4 consecutive columns of a row (16-byte stores)
    __syncthreads();
    float (*sC)[GM_BN + 4] = reinterpret_cast<float (*)[GM_BN + 4]>(smem);
#pragma unroll
    for (int i = 0; i < 16; ++i) sC[wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh][wn * 32 + li] = acc[i];
    __syncthreads();
    const int c4 = (tid & 15) * 4, n = n0 + c4;
    if (n >= G.N) return;
    const bool split = G.partial != nullptr;
    float* out = split ? G.partial + (int64_t)blockIdx.z * G.M * G.N : G.C;
    const int64_t ldo = split ? G.N : G.ldc;
    const bool vec = n + 3 < G.N && (ldo & 3) == 0;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (G.bias && !split) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < G.N) bias[j] = G.bias[n + j];
    }
    for (int r = tid >> 4; r < GM_BM; r += 16) {
        const int m = m0 + r;
        if (m >= G.M) break;
        f32x4 y = *reinterpret_cast<const f32x4*>(&sC[r][c4]);
        y.x += bias[0]; y.y += bias[1]; y.z += bias[2]; y.w += bias[3];
        float* o = out + (int64_t)m * ldo + n;
        if (vec) *reinterpret_cast<f32x4*>(o) = y;
        else { o[0] = y.x; if (n + 1 < G.N) o[1] = y.y; if (n + 2 < G.N) o[2] = y.z; if (n + 3 < G.N) o[3] = y.w; }
    }
}

// C[m][n] = bias[n] + sum_z partial[z][m][n], z ascending
__global__ __launch_bounds__(256) void k_gemm_reduce(const float* __restrict__ partial, int splits, int M, int N,
                                                      const float* __restrict__ bias, float* __restrict__ C, int64_t ldc) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
        float s = partial[i];
        for (int z = 1; z < splits; ++z) s += partial[(int64_t)z * total + i];
        if (bias) s += bias[n];
        C[(int64_t)m * ldc + n] = s;
    }
}

static int gemm_splits(int M, int N, int K) {
    const int64_t tiles = (int64_t)((M + GM_BM - 1) / GM_BM) * ((N + GM_BN - 1) / GM_BN);
    const int nk = (K + GM_BK - 1) / GM_BK;
    // one workgroup per CU, two for long reductions (K >= 2048: 1024 x 1536 x 4096 181 -> 128 us, 1024 x 1024 x 4096 104 -> 90 us;
    // short ones lose more to the summing kernel than they gain); at least 4 K-steps per slice
    const int fill = nk >= 64 ? 512 : 256;
    int s = 1;
    while (tiles * s < fill && s * 2 <= 16 && nk / (s * 2) >= 4) s *= 2;
    return s;
}

// Column sums of a row-major matrix: out[c] = sum_r X[r, c] (bias gradients of the batched Linear layers, slice sums of
// the LayerNorm parameter gradients).  A workgroup owns 16 columns: 4 threads x 16 bytes across, 64 row groups down; a
// thread adds every 64th row, then the row groups are added through LDS in a fixed order (same bits every time).
// Narrow strips on purpose: a 1024 x 512 matrix becomes 32 workgroups of 16 loads per thread instead of 2 of 256.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ X, int64_t ldx, int rows, int cols, float* __restrict__ out) {
    __shared__ f32x4 part[64][4];
    const int tid = threadIdx.x, cq = tid & 3, rg = tid >> 2;
    const int c = blockIdx.x * 16 + cq * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < cols) {                               // cols % 4 == 0
        const float* p = X + (int64_t)rg * ldx + c;
        int r = rg;
#pragma unroll 1
        for (; r + 7 * 64 < rows; r += 8 * 64) {   // eight independent loads in flight
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)u * 64 * ldx);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
            p += (int64_t)8 * 64 * ldx;
        }
        for (; r < rows; r += 64) { acc += *reinterpret_cast<const f32x4*>(p); p += (int64_t)64 * ldx; }
    }
    part[rg][cq] = acc;
    __syncthreads();
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) {
        if (rg < h) part[rg][cq] += part[rg + h][cq];
        __syncthreads();
    }
    if (rg == 0 && c < cols) *reinterpret_cast<f32x4*>(out + c) = part[0][cq];
}

}  // namespace drm

using namespace drm;

extern "C" size_t drm_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    const int s = gemm_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int drm_gemm_f32(int32_t trans_a, int32_t trans_b, int32_t M, int32_t N, int32_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc, void* workspace,
                            size_t workspace_bytes, drm_stream_t stream) {
    DRM_REQUIRE(A && B && C, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(M >= 1 && N >= 1 && K >= 1, DRM_EINVAL, "M=%d N=%d K=%d", M, N, K);
    DRM_REQUIRE(aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0, DRM_EINVAL,
                "operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    DRM_REQUIRE(trans_a || K % 4 == 0, DRM_EINVAL, "K=%d must be a multiple of 4 for a k-contiguous A", K);
    DRM_REQUIRE(trans_b || K % 4 == 0, DRM_EINVAL, "K=%d must be a multiple of 4 for a k-contiguous B", K);
    DRM_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? N : K) && ldc >= N, DRM_EINVAL, "leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    const int splits = gemm_splits(M, N, K);
    const int nk = (K + GM_BK - 1) / GM_BK;
    GemmArgs g{A, lda, B, ldb, bias, C, ldc, M, N, K, (nk + splits - 1) / splits, nullptr};
    if (splits > 1) {
        DRM_REQUIRE(workspace && workspace_bytes >= (size_t)splits * M * N * sizeof(float) && aligned16(workspace), DRM_EINVAL,
                    "workspace too small: need %zu bytes", (size_t)splits * M * N * sizeof(float));
        g.partial = (float*)workspace;
    }
    const dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM, splits);
    {
        ProfScope prof(PC_LINEAR_BIG, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        if (!trans_a && !trans_b) k_gemm_f32<false, false><<<grid, dim3(256), 0, s>>>(g);
        else if (!trans_a && trans_b) k_gemm_f32<false, true><<<grid, dim3(256), 0, s>>>(g);
        else if (trans_a && !trans_b) k_gemm_f32<true, false><<<grid, dim3(256), 0, s>>>(g);
        else k_gemm_f32<true, true><<<grid, dim3(256), 0, s>>>(g);
        DRM_LAUNCH_CHECK();
    }
    if (splits > 1) {
        const int64_t total = (int64_t)M * N;
        const int64_t blocks = (total + 255) / 256;
        ProfScope prof(PC_ELEMENTWISE, s, (double)splits * total, 4.0 * (splits + 1.0) * total);
        k_gemm_reduce<<<dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s>>>(g.partial, splits, M, N, bias, C, ldc);
        DRM_LAUNCH_CHECK();
    }
    return DRM_OK;
}

extern "C" int drm_colsum_f32(const float* x, int64_t ldx, int32_t rows, int32_t cols, float* out, drm_stream_t stream) {
    DRM_REQUIRE(x && out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(rows >= 1 && cols >= 4 && cols % 4 == 0 && ldx >= cols && ldx % 4 == 0, DRM_EINVAL, "rows=%d cols=%d ldx=%lld (multiples of 4)",
                rows, cols, (long long)ldx);
    DRM_REQUIRE(aligned16(x) && aligned16(out), DRM_EINVAL, "tensors must be 16-byte aligned");
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, (double)rows * cols, 4.0 * ((double)rows * cols + cols));
    k_colsum<<<dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, (hipStream_t)stream>>>(x, ldx, rows, cols, out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
