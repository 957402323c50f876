// HBM-bound kernels: lambda-return scan, replay sequence gather, standalone categorical sampling.
#include "common.hpp"

namespace drm {

// Streamed once: these kernels' loads and stores carry the non-temporal hint (nothing they touch is read again
// soon, so the lines need not displace what the caches hold).
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stnt4(float* p, const float4& v) {
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
}

// ------------------------------------------------------------------------------------------------
// lambda-return (lib/trainer.py:217-224).  One block of 256 threads owns ROWS consecutive trajectories: their
// R/C/V values are one contiguous run of ROWS*H floats in each input, moved with 16-byte accesses, four per
// array in flight per thread; the reverse scan itself runs from LDS, one lane per trajectory, with an odd
// row stride so the 64 lanes of a wave hit different banks.  ROWS shrinks as the horizon grows (256 down to 16
// trajectories per block) so that the three staged arrays stay within 64 KB and two blocks share a CU.  The two products that do not depend on the
// recurrence, gamma*(1-c) and (1-lambda)*v, are formed while loading; every operation is rounded on its
// own, in the order of the eager fp32 expression, so results are bit-identical to the reference loop.
// ------------------------------------------------------------------------------------------------
template <int ROWS>
__global__ __launch_bounds__(256) void k_lambda_return(const float* __restrict__ R, const float* __restrict__ C,
                                                          const float* __restrict__ V,
                                                          const float* __restrict__ VL, float* __restrict__ out,
                                                          int64_t B, int H, float gamma, float one_minus_lam,
                                                          float lam, int vec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HS = H | 1;                     // LDS row stride; odd H: the tile is stored exactly as it lies in memory
    float* sA = sm;                           // r, then the returns
    float* sD = sm + ROWS * HS;               // gamma * (1 - c)
    float* sM = sm + 2 * ROWS * HS;           // (1 - lambda) * v
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int rows = (int)min((int64_t)ROWS, B - row0);
    const int64_t base = row0 * H;            // a multiple of 4 floats (ROWS is)
    const int n = rows * H;
    float nxt = tid < rows ? VL[row0 + tid] : 0.f;
    const int n4 = vec ? n >> 2 : 0;
    const float4* R4 = reinterpret_cast<const float4*>(R + base);
    const float4* C4 = reinterpret_cast<const float4*>(C + base);
    const float4* V4 = reinterpret_cast<const float4*>(V + base);
    constexpr int U = 4, NT = 256;
    for (int k0 = 0; k0 < n4; k0 += NT * U) {
        float4 r[U], c[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(k0 + u * NT + tid, n4 - 1);
            r[u] = ldnt4(reinterpret_cast<const float*>(R4 + i)); c[u] = ldnt4(reinterpret_cast<const float*>(C4 + i));
            v[u] = ldnt4(reinterpret_cast<const float*>(V4 + i));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = k0 + u * NT + tid;
            if (i >= n4) continue;
            const float a4[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
            const float d4[4] = {__fmul_rn(gamma, __fsub_rn(1.0f, c[u].x)), __fmul_rn(gamma, __fsub_rn(1.0f, c[u].y)),
                                 __fmul_rn(gamma, __fsub_rn(1.0f, c[u].z)), __fmul_rn(gamma, __fsub_rn(1.0f, c[u].w))};
            const float m4[4] = {__fmul_rn(one_minus_lam, v[u].x), __fmul_rn(one_minus_lam, v[u].y),
                                 __fmul_rn(one_minus_lam, v[u].z), __fmul_rn(one_minus_lam, v[u].w)};
            if (HS == H) {
                reinterpret_cast<float4*>(sA)[i] = r[u];
                reinterpret_cast<float4*>(sD)[i] = make_float4(d4[0], d4[1], d4[2], d4[3]);
                reinterpret_cast<float4*>(sM)[i] = make_float4(m4[0], m4[1], m4[2], m4[3]);
            } else {
                int rr = (i * 4) / H, t = i * 4 - rr * H;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = rr * HS + t;
                    sA[pos] = a4[j]; sD[pos] = d4[j]; sM[pos] = m4[j];
                    if (++t == H) { t = 0; ++rr; }
                }
            }
        }
    }
    for (int e = n4 * 4 + tid; e < n; e += NT) {         // unaligned inputs, or the last elements of a short tile
        const int rr = e / H, pos = rr * HS + (e - rr * H);
        sA[pos] = R[base + e];
        sD[pos] = __fmul_rn(gamma, __fsub_rn(1.0f, C[base + e]));
        sM[pos] = __fmul_rn(one_minus_lam, V[base + e]);
    }
    __syncthreads();
    if (tid < rows) {
        float* a = sA + tid * HS;
        const float* d = sD + tid * HS;
        const float* m = sM + tid * HS;
#pragma unroll 4
        for (int t = H - 1; t >= 0; --t) {
            nxt = __fadd_rn(a[t], __fmul_rn(d[t], __fadd_rn(m[t], __fmul_rn(lam, nxt))));
            a[t] = nxt;
        }
    }
    __syncthreads();
    float4* O4 = reinterpret_cast<float4*>(out + base);
    for (int i = tid; i < n4; i += NT) {
        if (HS == H) {
            stnt4(reinterpret_cast<float*>(O4 + i), reinterpret_cast<const float4*>(sA)[i]);
        } else {
            int rr = (i * 4) / H, t = i * 4 - rr * H;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = sA[rr * HS + t];
                if (++t == H) { t = 0; ++rr; }
            }
            stnt4(reinterpret_cast<float*>(O4 + i), make_float4(o[0], o[1], o[2], o[3]));
        }
    }
    for (int e = n4 * 4 + tid; e < n; e += NT) {
        const int rr = e / H;
        out[base + e] = sA[rr * HS + (e - rr * H)];
    }
}

// Longer horizons: one wave owns 64 trajectories and walks their time axis backwards in chunks of 32 steps
// (128 contiguous bytes per trajectory and array), carrying the running return in a register.
__global__ __launch_bounds__(64) void k_lambda_return_chunked(const float* __restrict__ R, const float* __restrict__ C,
                                                              const float* __restrict__ V, const float* __restrict__ VL,
                                                              float* __restrict__ out, int64_t B, int H, float gamma,
                                                              float one_minus_lam, float lam) {
    constexpr int TC = 32, TS = 33;
    __shared__ float sA[64 * TS], sD[64 * TS], sM[64 * TS];
    const int lane = threadIdx.x, tl = lane & 31, rh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)min((int64_t)64, B - row0);
    float nxt = lane < rows ? VL[row0 + lane] : 0.f;
    for (int t1 = H; t1 > 0; t1 -= TC) {
        const int t0 = max(t1 - TC, 0), w = t1 - t0;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const int rr = 2 * k + rh;
            if (rr < rows && tl < w) {
                const int64_t idx = (row0 + rr) * H + t0 + tl;
                const int pos = rr * TS + tl;
                sA[pos] = R[idx];
                sD[pos] = __fmul_rn(gamma, __fsub_rn(1.0f, C[idx]));
                sM[pos] = __fmul_rn(one_minus_lam, V[idx]);
            }
        }
        __syncthreads();
        if (lane < rows) {
            float* a = sA + lane * TS;
            const float* d = sD + lane * TS;
            const float* m = sM + lane * TS;
#pragma unroll 4
            for (int t = w - 1; t >= 0; --t) {
                nxt = __fadd_rn(a[t], __fmul_rn(d[t], __fadd_rn(m[t], __fmul_rn(lam, nxt))));
                a[t] = nxt;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const int rr = 2 * k + rh;
            if (rr < rows && tl < w) out[(row0 + rr) * H + t0 + tl] = sA[rr * TS + tl];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// replay sequence gather (lib/replay_buffer.py:46-85).  One block per (b, t) frame.  Bytes only: 16-byte
// lane accesses, consecutive lanes on consecutive chunks (1 KiB per wave instruction), four loads in flight
// per thread before the first store.  With the fp32 image (lib/trainer.py:163, obs.float() / 255) the frame is
// read once in 4-byte words (256 contiguous bytes per wave instruction); each word is stored as it is and
// as one 16-byte group of four floats, so both outputs are written contiguously across the wave.
// Lane 0 moves the three scalars.
// ------------------------------------------------------------------------------------------------
// x / 255 correctly rounded for x = 0..255 without the division sequence: q = x * fl(1/255), one exact
// residual and one correction (equal to the IEEE quotient for all 256 inputs)
__device__ __forceinline__ float byte_over_255(uint32_t x) {
    const float r = 1.0f / 255.0f;
    const float f = (float)x;
    const float q = __fmul_rn(f, r);
    return fmaf(fmaf(-q, 255.0f, f), r, q);
}

__global__ __launch_bounds__(256) void k_seq_gather(drm_ring ring, const int64_t* __restrict__ starts, int T,
                                                     uint8_t* __restrict__ obs_out, int64_t* __restrict__ act_out,
                                                     float* __restrict__ rew_out, uint8_t* __restrict__ cont_out,
                                                     float* __restrict__ obs_f32) {
    const int64_t frame = blockIdx.x;  // b*T + t
    const int b = (int)(frame / T), t = (int)(frame - (int64_t)b * T);
    const int64_t src = (starts[b] + t) % ring.capacity;
    const int tid = threadIdx.x;
    constexpr int U = 4;
    if (obs_f32 == nullptr) {
        const u32x4s* s16 = reinterpret_cast<const u32x4s*>(ring.observations + src * ring.frame_bytes);
        u32x4s* d16 = reinterpret_cast<u32x4s*>(obs_out + frame * ring.frame_bytes);
        const int n16 = (int)(ring.frame_bytes >> 4);
        for (int k0 = 0; k0 < n16; k0 += 256 * U) {
            u32x4s v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(s16 + min(k0 + u * 256 + tid, n16 - 1));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = k0 + u * 256 + tid;
                if (i < n16) __builtin_nontemporal_store(v[u], d16 + i);
            }
        }
    } else {
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(ring.observations + src * ring.frame_bytes);
        uint32_t* d4 = reinterpret_cast<uint32_t*>(obs_out + frame * ring.frame_bytes);
        float4* d = reinterpret_cast<float4*>(obs_f32 + frame * ring.frame_bytes);
        const int n4 = (int)(ring.frame_bytes >> 2);
        for (int k0 = 0; k0 < n4; k0 += 256 * U) {
            uint32_t w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(s4 + min(k0 + u * 256 + tid, n4 - 1));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = k0 + u * 256 + tid;
                if (i < n4) {
                    __builtin_nontemporal_store(w[u], d4 + i);
                    stnt4(reinterpret_cast<float*>(d + i), make_float4(byte_over_255(w[u] & 0xffu), byte_over_255((w[u] >> 8) & 0xffu),
                                                                         byte_over_255((w[u] >> 16) & 0xffu), byte_over_255(w[u] >> 24)));
                }
            }
        }
    }
    if (tid == 0) {
        act_out[frame] = ring.actions[src];
        rew_out[frame] = ring.rewards[src];
        cont_out[frame] = ring.continues[src];
    }
}

// ------------------------------------------------------------------------------------------------
// straight-through categorical sampling over K = 32 classes.  Eight lanes own one categorical
// (four classes each, one 16-byte access per tensor), so a wave handles eight categoricals per
// pass and every reduction is three xor-shuffles inside the 8-lane group.
// softmax(log_unimix(logits) - log E)_k = w_k / sum_j w_j with w_k = ((1-eps) softmax(logits)_k + eps/K) / E_k
// (lib/world_model.py:459-461 with lib/utils.py:20-31 and F.gumbel_softmax written out): one exponential
// and one division per class, no logarithm; the kernel then runs at memory speed instead of ALU speed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_st(const float* __restrict__ logits, const float* __restrict__ expo,
                                                    int64_t rows, float one_m_eps, float eps_over_k,
                                                    int use_unimix, float* __restrict__ z_out,
                                                    uint8_t* __restrict__ idx_out, float* __restrict__ y_out) {
    const int sub = threadIdx.x & 7;
    const int64_t stride = (int64_t)gridDim.x * 32;
    for (int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); row < rows; row += stride) {
        const float4 lg = ldnt4(logits + row * 32 + sub * 4);
        const float4 ex = ldnt4(expo + row * 32 + sub * 4);
        const float v[4] = {lg.x, lg.y, lg.z, lg.w};
        const float e[4] = {ex.x, ex.y, ex.z, ex.w};
        const float mx = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float w[4], s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = expf(v[j] - mx); s += w[j]; }
        if (use_unimix) {
            const float scale = one_m_eps * __builtin_amdgcn_rcpf(group_sum<8>(s));
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = fmaf(w[j], scale, eps_over_k);
        }
        float best = -1.f;
        int bi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = w[j] * __builtin_amdgcn_rcpf(fmaxf(e[j], 1e-30f));   // reciprocal within 1 ulp
            if (w[j] > best) { best = w[j]; bi = sub * 4 + j; }
        }
        group_argmax<8>(best, bi);
        if (y_out) {
            const float inv = __builtin_amdgcn_rcpf(group_sum<8>((w[0] + w[1]) + (w[2] + w[3])));
            stnt4(y_out + row * 32 + sub * 4, make_float4(w[0] * inv, w[1] * inv, w[2] * inv, w[3] * inv));
        }
        if (z_out) {
            const int k0 = sub * 4;
            stnt4(z_out + row * 32 + k0, make_float4(bi == k0 ? 1.f : 0.f, bi == k0 + 1 ? 1.f : 0.f, bi == k0 + 2 ? 1.f : 0.f,
                                                     bi == k0 + 3 ? 1.f : 0.f));
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
    const int HS = H | 1;
    const float g = (float)gamma, oml = (float)(1.0 - lam), l = (float)lam;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PC_LAMBDA, s, 6.0 * B * H, 16.0 * B * H + 4.0 * B);
    const int vec = aligned16(R) && aligned16(C) && aligned16(V) && aligned16(out);
    const size_t per_row = (size_t)3 * HS * 4;       // staged bytes per trajectory
#define DRM_LAMBDA(ROWS) k_lambda_return<ROWS><<<dim3((unsigned)((B + ROWS - 1) / ROWS)), dim3(256), ROWS * per_row, s>>>(R, C, V, V_last, out, B, H, g, oml, l, vec)
    if (256 * per_row <= 64 * 1024) DRM_LAMBDA(256);
    else if (128 * per_row <= 64 * 1024) DRM_LAMBDA(128);
    else if (64 * per_row <= 64 * 1024) DRM_LAMBDA(64);
    else if (32 * per_row <= 64 * 1024) DRM_LAMBDA(32);
    else if (16 * per_row <= 64 * 1024) DRM_LAMBDA(16);
    else {
        k_lambda_return_chunked<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s>>>(R, C, V, V_last, out, B, H, g, oml, l);
    }
#undef DRM_LAMBDA
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
    const float ome = use ? 1.0f - unimix_eps : 1.f;
    const float eok = use ? unimix_eps / 32.0f : 0.f;
    const int64_t blocks = (rows + 31) / 32;
    // one pass per block while the grid fits: consecutive blocks then sweep memory as one moving window
    const unsigned grid = (unsigned)(blocks < (1 << 24) ? blocks : (1 << 24));
    ProfScope prof(PC_SAMPLE, (hipStream_t)stream, 12.0 * rows * 32,
                   (double)rows * (256.0 + (z_out ? 128.0 : 0.0) + (y_out ? 128.0 : 0.0) + (idx_out ? 1.0 : 0.0)));
    k_sample_st<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(logits, expo, rows, ome, eok, use, z_out, idx_out, y_out);
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
                                                    int vec, unsigned long long* __restrict__ rng_state,
                                                    unsigned long long rng_counter) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    if (tid == 0 && rng_state != nullptr) *rng_state = rng_counter;
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

extern "C" int drm_copy_pair_rng(const void* a_src, void* a_dst, size_t a_bytes, const void* b_src, void* b_dst, size_t b_bytes,
                                 uint64_t* rng_state, uint64_t rng_counter, drm_stream_t stream) {
    DRM_REQUIRE((a_bytes == 0 || (a_src && a_dst)) && (b_bytes == 0 || (b_src && b_dst)), DRM_EINVAL, "null pointer");
    if (a_bytes + b_bytes == 0 && rng_state == nullptr) return DRM_OK;
    const bool vec = aligned16(a_src) && aligned16(a_dst) && aligned16(b_src) && aligned16(b_dst) && a_bytes % 16 == 0 &&
                     b_bytes % 16 == 0;
    const size_t units = vec ? (a_bytes + b_bytes) / 16 : a_bytes + b_bytes;
    size_t blocks = (units + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    ProfScope prof(PC_ELEMENTWISE, (hipStream_t)stream, 0.0, 2.0 * (double)(a_bytes + b_bytes));
    k_copy_pair<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
        (const uint8_t*)a_src, (uint8_t*)a_dst, a_bytes, (const uint8_t*)b_src, (uint8_t*)b_dst, b_bytes, vec ? 1 : 0,
        (unsigned long long*)rng_state, (unsigned long long)rng_counter);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_copy_pair(const void* a_src, void* a_dst, size_t a_bytes, const void* b_src, void* b_dst, size_t b_bytes,
                             drm_stream_t stream) {
    if (a_bytes + b_bytes == 0) return DRM_OK;
    return drm_copy_pair_rng(a_src, a_dst, a_bytes, b_src, b_dst, b_bytes, nullptr, 0, stream);
}
