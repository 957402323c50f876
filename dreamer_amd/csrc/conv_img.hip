// The image-side ends of the convolution towers: a stride-2 4x4 layer (padding 1) between an image with a handful of
// channels (N, Cimg <= 4, 2H, 2W; plain NCHW memory) and a feature map at half the resolution (N, H, W, Cs; channels-last
// memory, Cs a multiple of 32).  Both nn.Conv2d(Cimg -> Cs) (lib/world_model.py:94-97, the first encoder layer) and
// nn.ConvTranspose2d(Cs -> Cimg) (lib/world_model.py:298-302, the last decoder layer) tie small[i][j] to
// big[2i - 1 + ky][2j - 1 + kx] through a weight laid out [cs][cimg][ky][kx], so three kernels cover what the two layers
// need:
//
//   DOWN  : small = conv(big)            first encoder layer forward (the transposed layer's input gradient is the
//                                        direct kernel of convt.hip)
//   UP    : big = convT(small) + bias    last decoder layer forward from a channels-last input
//   WGRAD : dW[cs][cimg][ky][kx] = sum over images and positions of small * big      (both layers' weight gradient)
//
// DOWN and WGRAD are products with one long and two short extents (positions x 16 Cimg x 32): fp32 matrix instructions
// (v_mfma_f32_32x32x2_f32) fed straight from global memory -- the feature-map operand is 128 contiguous bytes per
// position, the image operand a gather of 4-byte values that stay in the caches (the image is read 4 times over).
// UP has 4 Cimg <= 16 results per position and runs on the vector ALUs: eight lanes share a position (four channels
// each, 16-byte reads that cover a position's 128-byte channel vector), weights come from LDS, partial sums meet in a
// three-level DPP reduction.  All three are streams at the training shape (134 MB of features, 50 MB of image).
#include "common.hpp"

namespace drm {

// ---- DOWN ----------------------------------------------------------------------------------------------------------
// Wave = 32 consecutive positions x 32 output channels; K = 16 Cimg image values per position, two per matrix
// instruction.  The weights of the wave's 32 channels sit in 8 Cimg registers per lane for the whole kernel.
template <int CIMG>
__global__ __launch_bounds__(256) void k_img_down(const float* __restrict__ big, const float* __restrict__ Wt,
                                                   float* __restrict__ small, int64_t P, int Cs, int H, int W) {
    constexpr int KK = 16 * CIMG, KS = KK / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, half = lane >> 5;
    const int cs0 = blockIdx.y * 32;
    float bw[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bw[s] = Wt[(int64_t)(cs0 + m) * KK + 2 * s + half];
    const int H2 = 2 * H, W2 = 2 * W;
    const int64_t HW = (int64_t)H * W, ntiles = (P + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t pos = tile * 32 + m;
        const bool valid = pos < P;
        const int64_t pc = valid ? pos : P - 1;
        const int64_t n = pc / HW;
        const int r = (int)(pc - n * HW);
        const int i = r / W, j = r - i * W;
        const float* img = big + n * (int64_t)CIMG * H2 * W2;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // kk = 2 s + half: the two values of a step differ in kx only
            const int cimg = (2 * s) >> 4, ky = ((2 * s) >> 2) & 3, kx = ((2 * s) & 3) + half;
            const int y = 2 * i - 1 + ky, x = 2 * j - 1 + kx;
            const bool ok = valid && (unsigned)y < (unsigned)H2 && (unsigned)x < (unsigned)W2;
            const float v = img[((int64_t)cimg * H2 + (ok ? y : 0)) * W2 + (ok ? x : 0)];     // clamped address, masked value
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? v : 0.f, bw[s], acc, 0, 0, 0);
        }
        // C layout: column (channel) = lane & 31, row (position) = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t p = tile * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (p < P) small[p * Cs + cs0 + m] = acc[q];
        }
    }
}

// ---- WGRAD ---------------------------------------------------------------------------------------------------------
// Wave = a contiguous run of positions (two per matrix instruction) accumulated into a 32 channels x 16 Cimg tile;
// four steps' operands are requested before the first of their matrix instructions.  The four waves of a workgroup
// add up in LDS one after another and the workgroup writes one partial tile, summed by k_img_wgrad_reduce in a fixed
// order.
template <int CIMG>
__global__ __launch_bounds__(256) void k_img_wgrad(const float* __restrict__ small, const float* __restrict__ big,
                                                    float* __restrict__ part, int64_t P, int Cs, int H, int W,
                                                    int64_t steps_per_wave) {
    constexpr int KK = 16 * CIMG, NT = (KK + 31) / 32, U = 4;
    __shared__ float red[32][NT * 32 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, half = lane >> 5;
    const int cs0 = blockIdx.y * 32;
    const int H2 = 2 * H, W2 = 2 * W;
    const int64_t HW = (int64_t)H * W;
    // image-side value this lane gathers for each column tile: kk = 32 t + m
    int cimg[NT], ky[NT], kx[NT];
    bool kv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int kk = 32 * t + m;
        kv[t] = kk < KK;
        const int kc = kv[t] ? kk : 0;
        cimg[t] = kc >> 4; ky[t] = (kc >> 2) & 3; kx[t] = kc & 3;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const int64_t nsteps = (P + 1) / 2;
    const int64_t st0 = ((int64_t)blockIdx.x * 4 + wave) * steps_per_wave;
    const int64_t st1 = st0 + steps_per_wave < nsteps ? st0 + steps_per_wave : nsteps;
    if (st0 < st1) {        // (uniform over the wave)
        // two positions per step: (image, row, column) advance by a fixed amount with one carry each
        const int dn = (int)(2 / HW), rem = (int)(2 % HW), di = rem / W, dj = rem - di * W;
        int64_t pos = 2 * st0 + half;
        const int64_t pcl = pos < P ? pos : P - 1;
        int n = (int)(pcl / HW);
        const int r0 = (int)(pcl - (int64_t)n * HW);
        int i = r0 / W, j = r0 - i * W;
        const int nlast = (int)((P - 1) / HW);
        for (int64_t st = st0; st < st1; st += U) {
            float a[U], b[U][NT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = pos < P && st + u < st1;
                const int nn = n <= nlast ? n : nlast;              // the (masked) image pointer stays inside the tensor
                const float* img = big + (int64_t)nn * CIMG * H2 * W2;
                const float av = small[(valid ? pos : P - 1) * Cs + cs0 + m];
                a[u] = valid ? av : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int y = 2 * i - 1 + ky[t], x = 2 * j - 1 + kx[t];
                    const bool ok = valid && kv[t] && (unsigned)y < (unsigned)H2 && (unsigned)x < (unsigned)W2;
                    const float v = img[((int64_t)cimg[t] * H2 + (ok ? y : 0)) * W2 + (ok ? x : 0)];
                    b[u][t] = ok ? v : 0.f;
                }
                pos += 2;
                j += dj;
                const int cj = j >= W ? 1 : 0;
                j -= cj * W;
                i += di + cj;
                const int ci = i >= H ? 1 : 0;
                i -= ci * H;
                n += dn + ci;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    // accumulator layout: column (kk) = lane & 31, row (channel) = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float* cell = &red[(q & 3) + 8 * (q >> 2) + 4 * half][32 * t + m];
                    *cell = w == 0 ? acc[t][q] : *cell + acc[t][q];
                }
        }
        __syncthreads();
    }
    float* out = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 32 * KK;
    for (int e = threadIdx.x; e < 32 * KK; e += 256) {
        const int c = e / KK, kk = e - c * KK;
        out[e] = red[c][kk];
    }
}

// dW[(cs block, c, kk)] = sum over the workgroups' partial tiles, in workgroup order
__global__ __launch_bounds__(256) void k_img_wgrad_reduce(const float* __restrict__ part, int nparts, int64_t total,
                                                           float* __restrict__ dW) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * total + e];
    dW[e] = s;
}

// ---- UP ------------------------------------------------------------------------------------------------------------
// Workgroup = 32 positions of one image x 8 lanes each; lane q of a position owns channels 4 q .. 4 q + 3 of every
// block of 32.  A position forms the 2 x 2 output pixels it centres, for every image channel: output row 2i takes
// kernel rows 1 (input row i) and 3 (row i - 1), row 2i + 1 takes kernel rows 0 (row i + 1) and 2 (row i); columns
// likewise.
template <int CIMG>
__global__ __launch_bounds__(256) void k_img_up(const float* __restrict__ small, const float* __restrict__ Wt,
                                                 const float* __restrict__ bias, float* __restrict__ big, int Cs, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sw[];       // [cs][cimg][ky][kx]
    const int tid = threadIdx.x;
    for (int e = tid; e < Cs * CIMG * 16; e += 256) sw[e] = Wt[e];
    __syncthreads();
    const int q = tid & 7;
    const int HW = H * W;
    const int pos = blockIdx.x * 32 + (tid >> 3);
    const bool valid = pos < HW;
    const int pc = valid ? pos : HW - 1;
    const int i = pc / W, j = pc - i * W;
    const int64_t n = blockIdx.y;
    const float* x = small + n * (int64_t)HW * Cs;
    // the 3 x 3 neighbourhood: clamped addresses, validity as multipliers of 0 / 1 folded into a mask
    int off[3][3];
    bool ok[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ii = i + a - 1, jj = j + b - 1;
            ok[a][b] = valid && (unsigned)ii < (unsigned)H && (unsigned)jj < (unsigned)W;
            off[a][b] = ok[a][b] ? (ii * W + jj) * Cs : pc * Cs;
        }
    float acc[CIMG][2][2];
#pragma unroll
    for (int c = 0; c < CIMG; ++c) { acc[c][0][0] = 0.f; acc[c][0][1] = 0.f; acc[c][1][0] = 0.f; acc[c][1][1] = 0.f; }
    for (int c0 = 4 * q; c0 < Cs; c0 += 32) {
        f32x4 v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(x + off[a][b] + c0);
                v[a][b] = ok[a][b] ? t : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const float* wc = sw + (c0 + ch) * CIMG * 16;
#pragma unroll
            for (int c = 0; c < CIMG; ++c)
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wc + c * 16 + ky * 4);
                    const int a = (ky & 1) ? 0 : 1;
                    const int vr = ky == 0 ? 2 : (ky == 3 ? 0 : 1);
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int b = (kx & 1) ? 0 : 1;
                        const int vc = kx == 0 ? 2 : (kx == 3 ? 0 : 1);
                        acc[c][a][b] = fmaf(v[vr][vc][ch], w4[kx], acc[c][a][b]);
                    }
                }
        }
    }
    // the eight lanes of a position add up; lane q < 2 CIMG stores output row (channel q >> 1, row parity q & 1)
    float2 mine = make_float2(0.f, 0.f);
#pragma unroll
    for (int c = 0; c < CIMG; ++c)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float s0 = group_sum<8>(acc[c][a][0]), s1 = group_sum<8>(acc[c][a][1]);
            if (q == 2 * c + a) mine = make_float2(s0, s1);
        }
    if (valid && q < 2 * CIMG) {
        const int c = q >> 1, a = q & 1;
        const float bv = bias ? bias[c] : 0.f;
        const int H2 = 2 * H, W2 = 2 * W;
        float* y = big + ((n * CIMG + c) * H2 + 2 * i + a) * (int64_t)W2 + 2 * j;
        *reinterpret_cast<float2*>(y) = make_float2(mine.x + bv, mine.y + bv);
    }
}

static int img_check(int64_t N, int Cs, int Cimg, int H, int W) {
    DRM_REQUIRE(N >= 1 && N < 65536 && H >= 1 && W >= 1 && H <= 4096 && W <= 4096, DRM_EINVAL, "bad shape N=%lld H=%d W=%d",
                (long long)N, H, W);
    DRM_REQUIRE(Cs >= 32 && Cs % 32 == 0 && Cs <= 512, DRM_ENOTSUP, "feature channels must be a multiple of 32, at most 512 (got %d)", Cs);
    DRM_REQUIRE(Cimg >= 1 && Cimg <= 4, DRM_ENOTSUP, "built for at most 4 image channels (got %d)", Cimg);
    DRM_REQUIRE(N * H * W < ((int64_t)1 << 31), DRM_ENOTSUP, "too many positions");
    return DRM_OK;
}

static int cu_count_img() {
    static int n = [] {
        int dev = 0; hipDeviceProp_t p{};
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) return 256;
        return p.multiProcessorCount;
    }();
    return n;
}

// workgroups of the weight-gradient launch (per block of 32 feature channels) for P positions
static int wgrad_blocks(int64_t P) {
    const int64_t nsteps = (P + 1) / 2;
    int64_t blocks = (nsteps + 4 * 64 - 1) / (4 * 64);        // at least 64 steps per wave
    const int64_t cap = 8 * (int64_t)cu_count_img();        // eight workgroups per CU: the loop hides its gathers behind other waves
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace drm

using namespace drm;

extern "C" int drm_conv_img_down(const float* big, const float* weight, float* small_out, int64_t N, int32_t Cs, int32_t Cimg,
                                 int32_t H, int32_t W, drm_stream_t stream) {
    DRM_REQUIRE(big && weight && small_out, DRM_EINVAL, "null pointer");
    if (int rc = img_check(N, Cs, Cimg, H, W)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t P = N * H * W, ntiles = (P + 31) / 32;
    int64_t gx = (ntiles + 3) / 4;
    const int64_t cap = 8 * (int64_t)cu_count_img();
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)(Cs / 32)), block(256);
    ProfScope prof(PC_ELEMENTWISE, s, 2.0 * P * Cs * 16.0 * Cimg, 4.0 * ((double)P * Cs + 4.0 * P * Cimg + 16.0 * Cs * Cimg));
    switch (Cimg) {
        case 1: k_img_down<1><<<grid, block, 0, s>>>(big, weight, small_out, P, Cs, H, W); break;
        case 2: k_img_down<2><<<grid, block, 0, s>>>(big, weight, small_out, P, Cs, H, W); break;
        case 3: k_img_down<3><<<grid, block, 0, s>>>(big, weight, small_out, P, Cs, H, W); break;
        default: k_img_down<4><<<grid, block, 0, s>>>(big, weight, small_out, P, Cs, H, W); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_conv_img_up(const float* small_in, const float* weight, const float* bias, float* big_out, int64_t N, int32_t Cs,
                               int32_t Cimg, int32_t H, int32_t W, drm_stream_t stream) {
    DRM_REQUIRE(small_in && weight && big_out, DRM_EINVAL, "null pointer");
    if (int rc = img_check(N, Cs, Cimg, H, W)) return rc;
    DRM_REQUIRE(aligned16(small_in) && (reinterpret_cast<uintptr_t>(big_out) & 7u) == 0, DRM_EINVAL,
                "the feature map must be 16-byte aligned, the image 8-byte aligned");
    DRM_REQUIRE((int64_t)H * W * Cs < ((int64_t)1 << 31), DRM_ENOTSUP, "feature map of one image too large");
    DRM_REQUIRE(Cs <= 128, DRM_ENOTSUP, "the weights of at most 128 feature channels fit the kernel's LDS image (got %d)", Cs);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((H * W + 31) / 32), (unsigned)N), block(256);
    const size_t lds = (size_t)Cs * Cimg * 16 * sizeof(float);
    const double out = (double)N * Cimg * 4.0 * H * W;
    ProfScope prof(PC_ELEMENTWISE, s, 2.0 * out * Cs * 4.0, 4.0 * ((double)N * Cs * H * W + out + 16.0 * Cs * Cimg));
    switch (Cimg) {
        case 1: k_img_up<1><<<grid, block, lds, s>>>(small_in, weight, bias, big_out, Cs, H, W); break;
        case 2: k_img_up<2><<<grid, block, lds, s>>>(small_in, weight, bias, big_out, Cs, H, W); break;
        case 3: k_img_up<3><<<grid, block, lds, s>>>(small_in, weight, bias, big_out, Cs, H, W); break;
        default: k_img_up<4><<<grid, block, lds, s>>>(small_in, weight, bias, big_out, Cs, H, W); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" size_t drm_conv_img_wgrad_workspace_bytes(int64_t N, int32_t Cs, int32_t Cimg, int32_t H, int32_t W) {
    if (N < 1 || H < 1 || W < 1 || Cs < 32 || Cimg < 1) return 0;
    return (size_t)wgrad_blocks(N * H * W) * (size_t)Cs * 16 * (size_t)Cimg * sizeof(float);
}

extern "C" int drm_conv_img_wgrad(const float* small_in, const float* big, float* grad_weight, void* workspace,
                                  size_t workspace_bytes, int64_t N, int32_t Cs, int32_t Cimg, int32_t H, int32_t W,
                                  drm_stream_t stream) {
    DRM_REQUIRE(small_in && big && grad_weight && workspace, DRM_EINVAL, "null pointer");
    if (int rc = img_check(N, Cs, Cimg, H, W)) return rc;
    const int64_t P = N * H * W;
    const int blocks = wgrad_blocks(P);
    const int64_t total = (int64_t)Cs * 16 * Cimg;
    DRM_REQUIRE(workspace_bytes >= (size_t)blocks * (size_t)total * sizeof(float), DRM_EINVAL, "workspace too small: need %zu bytes",
                (size_t)blocks * (size_t)total * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nsteps = (P + 1) / 2;
    const int64_t spw = (nsteps + (int64_t)blocks * 4 - 1) / ((int64_t)blocks * 4);
    const dim3 grid((unsigned)blocks, (unsigned)(Cs / 32)), block(256);
    float* part = (float*)workspace;
    {
        ProfScope prof(PC_ELEMENTWISE, s, 2.0 * P * Cs * 16.0 * Cimg, 4.0 * ((double)P * Cs + 4.0 * P * Cimg));
        switch (Cimg) {
            case 1: k_img_wgrad<1><<<grid, block, 0, s>>>(small_in, big, part, P, Cs, H, W, spw); break;
            case 2: k_img_wgrad<2><<<grid, block, 0, s>>>(small_in, big, part, P, Cs, H, W, spw); break;
            case 3: k_img_wgrad<3><<<grid, block, 0, s>>>(small_in, big, part, P, Cs, H, W, spw); break;
            default: k_img_wgrad<4><<<grid, block, 0, s>>>(small_in, big, part, P, Cs, H, W, spw); break;
        }
        DRM_LAUNCH_CHECK();
    }
    ProfScope prof(PC_ELEMENTWISE, s, (double)blocks * total, 4.0 * ((double)blocks + 1.0) * total);
    k_img_wgrad_reduce<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(part, blocks, total, grad_weight);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
