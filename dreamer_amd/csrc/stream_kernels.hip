// HBM-bound kernels: lambda-return scan, replay sequence gather, standalone categorical sampling.
#include "common.hpp"

namespace drm {

// ------------------------------------------------------------------------------------------------
// lambda-return.  One block owns ROWS consecutive trajectories: their R/C/V values are one
// contiguous run of ROWS*H floats in each input, so global loads and stores are fully coalesced;
// the reverse scan itself runs from LDS, one lane per trajectory, row stride padded to an odd
// number of words so the 64 lanes of a wave hit 64 different banks.
// ------------------------------------------------------------------------------------------------
template <int ROWS>
__global__ __launch_bounds__(ROWS) void k_lambda_return(const float* __restrict__ R, const float* __restrict__ C,
                                                          const float* __restrict__ V,
                                                          const float* __restrict__ VL, float* __restrict__ out,
                                                          int64_t B, int H, float gamma, float one_minus_lam,
                                                          float lam) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HP = H | 1;
    float* sR = sm;
    float* sC = sm + ROWS * HP;
    float* sV = sm + 2 * ROWS * HP;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int rows = (int)min((int64_t)ROWS, B - row0);
    const int64_t base = row0 * H;
    const int n = rows * H;
    for (int i = threadIdx.x; i < n; i += ROWS) {
        const int r = i / H, t = i - r * H;
        sR[r * HP + t] = R[base + i];
        sC[r * HP + t] = C[base + i];
        sV[r * HP + t] = V[base + i];
    }
    __syncthreads();
    if ((int)threadIdx.x < rows) {
        float nxt = VL[row0 + threadIdx.x];
        float* r = sR + threadIdx.x * HP;
        const float* c = sC + threadIdx.x * HP;
        const float* v = sV + threadIdx.x * HP;
        for (int t = H - 1; t >= 0; --t) {
            // every operation rounded on its own, in the order of the eager fp32 expression
            const float disc = __fmul_rn(gamma, __fsub_rn(1.0f, c[t]));
            const float mix = __fadd_rn(__fmul_rn(one_minus_lam, v[t]), __fmul_rn(lam, nxt));
            nxt = __fadd_rn(r[t], __fmul_rn(disc, mix));
            r[t] = nxt;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += ROWS) {
        const int r = i / H, t = i - r * H;
        out[base + i] = sR[r * HP + t];
    }
}

// Long horizons (rows no longer fit LDS): one lane per trajectory straight from global memory.
__global__ __launch_bounds__(64) void k_lambda_return_long(const float* __restrict__ R, const float* __restrict__ C,
                                                           const float* __restrict__ V, const float* __restrict__ VL,
                                                           float* __restrict__ out, int64_t B, int H, float gamma,
                                                           float one_minus_lam, float lam) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float nxt = VL[b];
    for (int t = H - 1; t >= 0; --t) {
        const int64_t i = b * H + t;
        const float disc = __fmul_rn(gamma, __fsub_rn(1.0f, C[i]));
        const float mix = __fadd_rn(__fmul_rn(one_minus_lam, V[i]), __fmul_rn(lam, nxt));
        nxt = __fadd_rn(R[i], __fmul_rn(disc, mix));
        out[i] = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// replay sequence gather.  One block per (b, t) frame: 16-byte lane accesses, consecutive lanes on
// consecutive 16-byte chunks (1 KiB per wave instruction).  Lane 0 moves the three scalars.
// The optional fp32 image is produced in a second sweep over the frame with 4-byte loads and
// 16-byte stores so that the stores are contiguous across the wave as well.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seq_gather(drm_ring ring, const int64_t* __restrict__ starts, int T,
                                                     uint8_t* __restrict__ obs_out, int64_t* __restrict__ act_out,
                                                     float* __restrict__ rew_out, uint8_t* __restrict__ cont_out,
                                                     float* __restrict__ obs_f32) {
    const int64_t frame = blockIdx.x;  // b*T + t
    const int b = (int)(frame / T), t = (int)(frame - (int64_t)b * T);
    const int64_t src = (starts[b] + t) % ring.capacity;
    const uint4* s16 = reinterpret_cast<const uint4*>(ring.observations + src * ring.frame_bytes);
    uint4* d16 = reinterpret_cast<uint4*>(obs_out + frame * ring.frame_bytes);
    const int n16 = (int)(ring.frame_bytes >> 4);
    for (int i = threadIdx.x; i < n16; i += 256) d16[i] = s16[i];
    if (threadIdx.x == 0) {
        act_out[frame] = ring.actions[src];
        rew_out[frame] = ring.rewards[src];
        cont_out[frame] = ring.continues[src];
    }
    if (obs_f32 != nullptr) {
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s16);
        float4* d = reinterpret_cast<float4*>(obs_f32 + frame * ring.frame_bytes);
        const int n4 = (int)(ring.frame_bytes >> 2);
        for (int i = threadIdx.x; i < n4; i += 256) {
            const uint32_t w = s4[i];
            float4 f;
            f.x = __fdiv_rn((float)(w & 0xffu), 255.0f);
            f.y = __fdiv_rn((float)((w >> 8) & 0xffu), 255.0f);
            f.z = __fdiv_rn((float)((w >> 16) & 0xffu), 255.0f);
            f.w = __fdiv_rn((float)(w >> 24), 255.0f);
            d[i] = f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// straight-through categorical sampling over K = 32 classes.  Eight lanes own one categorical
// (four classes each, one 16-byte access per tensor), so a wave handles eight categoricals per
// pass and every reduction is three xor-shuffles inside the 8-lane group.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_st(const float* __restrict__ logits, const float* __restrict__ expo,
                                                    int64_t rows, float log1m_eps, float log_eps_over_k,
                                                    int use_unimix, float* __restrict__ z_out,
                                                    uint8_t* __restrict__ idx_out, float* __restrict__ y_out) {
    const int sub = threadIdx.x & 7;
    const int64_t stride = (int64_t)gridDim.x * 32;
    for (int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); row < rows; row += stride) {
        const float4 lg = *reinterpret_cast<const float4*>(logits + row * 32 + sub * 4);
        const float4 ex = *reinterpret_cast<const float4*>(expo + row * 32 + sub * 4);
        float v[4] = {lg.x, lg.y, lg.z, lg.w};
        const float e[4] = {ex.x, ex.y, ex.z, ex.w};
        if (use_unimix) {
            const float mx = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += expf(v[j] - mx);
            const float lse = logf(group_sum<8>(s));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = logaddexp_f((v[j] - mx - lse) + log1m_eps, log_eps_over_k);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] - logf(e[j]);
        const float smax = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float y[4], s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { y[j] = expf(v[j] - smax); s += y[j]; }
        s = group_sum<8>(s);
        float best = -1.f;
        int bi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = y[j] / s;
            if (y[j] > best) { best = y[j]; bi = sub * 4 + j; }
        }
        group_argmax<8>(best, bi);
        if (y_out) *reinterpret_cast<float4*>(y_out + row * 32 + sub * 4) = make_float4(y[0], y[1], y[2], y[3]);
        if (z_out) {
            const int k0 = sub * 4;
            *reinterpret_cast<float4*>(z_out + row * 32 + k0) =
                make_float4(bi == k0 ? 1.f : 0.f, bi == k0 + 1 ? 1.f : 0.f, bi == k0 + 2 ? 1.f : 0.f,
                            bi == k0 + 3 ? 1.f : 0.f);
        }
        if (idx_out && sub == 0) idx_out[row] = (uint8_t)bi;
    }
}

__global__ __launch_bounds__(256) void k_onehot_index(const float* __restrict__ z, int64_t rows, int K,
                                                       uint8_t* __restrict__ idx_out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* p = z + row * K;
    float best = p[0];
    int bi = 0;
    for (int k = 1; k < K; ++k)
        if (p[k] > best) { best = p[k]; bi = k; }
    idx_out[row] = (uint8_t)bi;
}

}  // namespace drm

using namespace drm;

extern "C" int drm_lambda_return(const float* R, const float* C, const float* V, const float* V_last, float* out,
                                 int64_t B, int32_t H, double gamma, double lam, drm_stream_t stream) {
    DRM_REQUIRE(R && C && V && V_last && out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(B >= 0 && H >= 1, DRM_EINVAL, "bad shape B=%lld H=%d", (long long)B, H);
    if (B == 0) return DRM_OK;
    const int HP = H | 1;
    const float g = (float)gamma, oml = (float)(1.0 - lam), l = (float)lam;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PC_LAMBDA, s, 6.0 * B * H, 16.0 * B * H + 4.0 * B);
    if ((size_t)3 * 256 * HP * 4 <= 64 * 1024) {
        const size_t sh = (size_t)3 * 256 * HP * 4;
        k_lambda_return<256><<<dim3((unsigned)((B + 255) / 256)), dim3(256), sh, s>>>(R, C, V, V_last, out, B, H, g, oml, l);
    } else if ((size_t)3 * 64 * HP * 4 <= 64 * 1024) {
        const size_t sh = (size_t)3 * 64 * HP * 4;
        k_lambda_return<64><<<dim3((unsigned)((B + 63) / 64)), dim3(64), sh, s>>>(R, C, V, V_last, out, B, H, g, oml, l);
    } else {
        k_lambda_return_long<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s>>>(R, C, V, V_last, out, B, H, g, oml, l);
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_seq_gather(const drm_ring* ring, const int64_t* starts, int32_t B, int32_t T, uint8_t* obs_out,
                              int64_t* actions_out, float* rewards_out, uint8_t* continues_out, float* obs_f32_out,
                              drm_stream_t stream) {
    DRM_REQUIRE(ring && starts && obs_out && actions_out && rewards_out && continues_out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(ring->observations && ring->actions && ring->rewards && ring->continues, DRM_EINVAL, "null ring storage");
    DRM_REQUIRE(B >= 0 && T >= 1 && ring->capacity >= 1, DRM_EINVAL, "bad shape B=%d T=%d capacity=%lld", B, T,
                (long long)ring->capacity);
    DRM_REQUIRE(ring->frame_bytes > 0 && ring->frame_bytes % 16 == 0, DRM_ENOTSUP, "frame_bytes %lld not a multiple of 16",
                (long long)ring->frame_bytes);
    DRM_REQUIRE(aligned16(ring->observations) && aligned16(obs_out) && aligned16(obs_f32_out), DRM_EINVAL,
                "observation pointers must be 16-byte aligned");
    if (B == 0) return DRM_OK;
    const double units = (double)B * T;
    ProfScope prof(PC_GATHER, (hipStream_t)stream, 0.0,
                   units * (2.0 * (ring->frame_bytes + 8 + 4 + 1) + (obs_f32_out ? 4.0 * ring->frame_bytes : 0.0)));
    k_seq_gather<<<dim3((unsigned)((int64_t)B * T)), dim3(256), 0, (hipStream_t)stream>>>(
        *ring, starts, T, obs_out, actions_out, rewards_out, continues_out, obs_f32_out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_sample_categorical_st(const float* logits, const float* expo, int64_t rows, int32_t K,
                                         float unimix_eps, float* z_out, uint8_t* idx_out, float* y_out,
                                         drm_stream_t stream) {
    DRM_REQUIRE(logits && expo, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(K == 32, DRM_ENOTSUP, "classes per latent must be 32, got %d", K);
    DRM_REQUIRE(rows >= 0, DRM_EINVAL, "bad rows");
    DRM_REQUIRE(aligned16(logits) && aligned16(expo) && aligned16(z_out) && aligned16(y_out), DRM_EINVAL,
                "pointers must be 16-byte aligned");
    if (rows == 0) return DRM_OK;
    const int use = unimix_eps >= 0.f;
    const float l1m = use ? (float)log1p(-(double)unimix_eps) : 0.f;
    const float lek = use ? (float)(log((double)unimix_eps) - log(32.0)) : 0.f;
    const int64_t blocks = (rows + 31) / 32;
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    ProfScope prof(PC_SAMPLE, (hipStream_t)stream, 40.0 * rows * 32,
                   (double)rows * (256.0 + (z_out ? 128.0 : 0.0) + (y_out ? 128.0 : 0.0) + (idx_out ? 1.0 : 0.0)));
    k_sample_st<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(logits, expo, rows, l1m, lek, use, z_out, idx_out, y_out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_onehot_to_index(const float* z, int64_t rows, int32_t K, uint8_t* idx_out, drm_stream_t stream) {
    DRM_REQUIRE(z && idx_out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(K >= 1 && K <= 256 && rows >= 0, DRM_EINVAL, "bad shape");
    if (rows == 0) return DRM_OK;
    k_onehot_index<<<dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(z, rows, K, idx_out);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

// ------------------------------------------------------------------------------------------------
// two device-to-device copies in one launch (start-state staging of a graph-replayed block: the
// recorded launches read fixed addresses, and each extra copy launch is a boundary on a path that
// is a few hundred microseconds long)
// ------------------------------------------------------------------------------------------------
namespace drm {
__global__ __launch_bounds__(256) void k_copy_pair(const uint8_t* __restrict__ a_src, uint8_t* __restrict__ a_dst, size_t a_bytes,
                                                    const uint8_t* __restrict__ b_src, uint8_t* __restrict__ b_dst, size_t b_bytes,
                                                    int vec) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    if (vec) {   // all pointers and sizes are multiples of 16
        const size_t na = a_bytes / 16, nb = b_bytes / 16;
        for (size_t i = tid; i < na + nb; i += step) {
            if (i < na) reinterpret_cast<uint4*>(a_dst)[i] = reinterpret_cast<const uint4*>(a_src)[i];
            else reinterpret_cast<uint4*>(b_dst)[i - na] = reinterpret_cast<const uint4*>(b_src)[i - na];
        }
    } else {
        for (size_t i = tid; i < a_bytes + b_bytes; i += step) {
            if (i < a_bytes) a_dst[i] = a_src[i];
            else b_dst[i - a_bytes] = b_src[i - a_bytes];
        }
    }
}
}  // namespace drm

extern "C" int drm_copy_pair(const void* a_src, void* a_dst, size_t a_bytes, const void* b_src, void* b_dst, size_t b_bytes,
                             drm_stream_t stream) {
    DRM_REQUIRE((a_bytes == 0 || (a_src && a_dst)) && (b_bytes == 0 || (b_src && b_dst)), DRM_EINVAL, "null pointer");
    if (a_bytes + b_bytes == 0) return DRM_OK;
    const bool vec = aligned16(a_src) && aligned16(a_dst) && aligned16(b_src) && aligned16(b_dst) && a_bytes % 16 == 0 &&
                     b_bytes % 16 == 0;
    const size_t units = vec ? (a_bytes + b_bytes) / 16 : a_bytes + b_bytes;
    size_t blocks = (units + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 0.0, 2.0 * (double)(a_bytes + b_bytes));
    k_copy_pair<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
        (const uint8_t*)a_src, (uint8_t*)a_dst, a_bytes, (const uint8_t*)b_src, (uint8_t*)b_dst, b_bytes, vec ? 1 : 0);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
