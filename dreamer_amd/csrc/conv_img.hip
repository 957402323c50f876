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
// UP has 4 Cimg <= 16 results per position and runs on the vector ALUs, one thread per position, the features staged
// through LDS and the weights as scalar operands.  All three are streams at the training shape (134 MB of features,
// 50 MB of image).
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
    const int H2 = 2 * H, W2 = 2 * W, imgsz = CIMG * H2 * W2, Hm1 = H - 1, Wm1 = W - 1;
    const int64_t HW = (int64_t)H * W, ntiles = (P + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t pos = tile * 32 + m;
        const bool valid = pos < P;
        const int64_t pc = valid ? pos : P - 1;
        const int n = (int)(pc / HW);
        const int r = (int)(pc - (int64_t)n * HW);
        const int i = r / W, j = r - i * W;
        // offsets (32-bit, from the tensor's start) of the gathers: pixel (2i, 2j) of channel 0 is always a valid address;
        // edge bits: 0 top row, 1 bottom row, 2 left column, 3 right column, 4 position past the end
        const int rc = n * imgsz + 2 * i * W2 + 2 * j;
        const unsigned e = (i == 0 ? 1u : 0u) | (i == Hm1 ? 2u : 0u) | (j == 0 ? 4u : 0u) | (j == Wm1 ? 8u : 0u) | (valid ? 0u : 16u);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        // every gather of the tile is requested before the first matrix instruction (the fence keeps the scheduler from
        // pairing each load with its consumer, which would expose the full memory latency once per step)
        float av[KS];
        unsigned okm = 0u;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // kk = 2 s + half: the two values of a step differ in kx only
            const int cimg = (2 * s) >> 4, ky = ((2 * s) >> 2) & 3, kx0 = (2 * s) & 3;      // kx = kx0 + half
            const unsigned em = (ky == 0 ? 1u : 0u) | (ky == 3 ? 2u : 0u) | 16u;
            // kx = 0 only for (kx0 = 0, half = 0); kx = 3 only for (kx0 = 2, half = 1)
            const unsigned emx = kx0 == 0 ? (half == 0 ? 4u : 0u) : (half == 1 ? 8u : 0u);
            const bool out = ((em | emx) & e) != 0u;
            av[s] = big[out ? rc : rc + (cimg * H2 + ky - 1) * W2 + kx0 + half - 1];
            okm |= (out ? 0u : 1u) << s;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(((okm >> s) & 1u) ? av[s] : 0.f, bw[s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
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
// eight steps' operands are requested before the first of their matrix instructions.  The eight waves of a workgroup
// add up in LDS one after another and the workgroup writes one partial tile, summed by k_img_wgrad_reduce in a fixed
// order.
constexpr int WG_WAVES = 8;      // waves per workgroup of the weight-gradient kernel
template <int CIMG>
__global__ __launch_bounds__(64 * WG_WAVES) void k_img_wgrad(const float* __restrict__ small, const float* __restrict__ big,
                                                    float* __restrict__ part, int64_t P, int Cs, int H, int W,
                                                    int64_t steps_per_wave) {
    constexpr int KK = 16 * CIMG, NT = (KK + 31) / 32, U = 8;
    __shared__ float red[32][NT * 32 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, half = lane >> 5;
    const int cs0 = blockIdx.y * 32;
    const int H2 = 2 * H, W2 = 2 * W;
    const int64_t HW = (int64_t)H * W;
    // image-side value this lane gathers for each column tile (kk = 32 t + m), as an offset from the pixel (2i, 2j) of
    // channel 0 and a mask of the image edges at which it falls outside (bit 0: top row and ky = 0, 1: bottom row and
    // ky = 3, 2: left column and kx = 0, 3: right column and kx = 3; 4: kk past the last value; 5: always set, met by
    // the step's bit 5 when the position itself is past the end)
    int base[NT];
    unsigned emask[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int kk = 32 * t + m;
        const bool kv = kk < KK;
        const int kc = kv ? kk : 0;
        const int cimg = kc >> 4, ky = (kc >> 2) & 3, kx = kc & 3;
        base[t] = (cimg * H2 + ky - 1) * W2 + kx - 1;
        emask[t] = (ky == 0 ? 1u : 0u) | (ky == 3 ? 2u : 0u) | (kx == 0 ? 4u : 0u) | (kx == 3 ? 8u : 0u) | (kv ? 0u : 16u) | 32u;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const int64_t nsteps = (P + 1) / 2;
    const int64_t st0 = ((int64_t)blockIdx.x * WG_WAVES + wave) * steps_per_wave;
    const int64_t st1 = st0 + steps_per_wave < nsteps ? st0 + steps_per_wave : nsteps;
    if (st0 < st1) {        // (uniform over the wave)
        // two positions per step: (image, row, column) advance by a fixed amount with one carry each
        const int dn = (int)(2 / HW), rem = (int)(2 % HW), di = rem / W, dj = rem - di * W;
        const int imgsz = CIMG * H2 * W2, Hm1 = H - 1, Wm1 = W - 1;
        int64_t pos = 2 * st0 + half;
        const int64_t pcl = pos < P ? pos : P - 1;
        int n = (int)(pcl / HW);
        const int r0 = (int)(pcl - (int64_t)n * HW);
        int i = r0 / W, j = r0 - i * W;
        const int nlast = (int)((P - 1) / HW);
        const float* ap = small + pos * Cs + cs0 + m;                  // this lane's feature value, 2 Cs further per step
        const float* const alast = small + (P - 1) * Cs + cs0 + m;
        for (int64_t st = st0; st < st1; st += U) {
            float a[U], b[U][NT];
            unsigned bad[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = pos < P && st + u < st1;
                const int nn = n <= nlast ? n : nlast;              // the (masked) image offset stays inside the tensor
                const int rc = nn * imgsz + 2 * i * W2 + 2 * j;      // pixel (2i, 2j) of channel 0: a valid address
                const unsigned e = (i == 0 ? 1u : 0u) | (i == Hm1 ? 2u : 0u) | (j == 0 ? 4u : 0u) | (j == Wm1 ? 8u : 0u) | 16u |
                                   (valid ? 0u : 32u);
                a[u] = *(valid ? ap : alast);
                bad[u] = 0u;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool out = (emask[t] & e) != 0u;
                    b[u][t] = big[out ? rc : rc + base[t]];
                    bad[u] |= (out ? 1u : 0u) << t;
                }
                if (!valid) bad[u] |= 1u << NT;
                pos += 2;
                ap += 2 * Cs;
                j += dj;
                const int cj = j >= W ? 1 : 0;
                j -= cj * W;
                i += di + cj;
                const int ci = i >= H ? 1 : 0;
                i -= ci * H;
                n += dn + ci;
            }
            __builtin_amdgcn_sched_barrier(0);      // all requests of the eight steps are out before the first matrix instruction
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float av = ((bad[u] >> NT) & 1u) ? 0.f : a[u];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ((bad[u] >> t) & 1u) ? 0.f : b[u][t], acc[t], 0, 0, 0);
            }
        }
    }
    // accumulator layout: column (kk) = lane & 31, row (channel) = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    for (int w = 0; w < WG_WAVES; ++w) {
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
    for (int e = threadIdx.x; e < 32 * KK; e += 64 * WG_WAVES) {
        const int c = e / KK, kk = e - c * KK;
        out[e] = red[c][kk];
    }
}

// dW[(cs block, c, kk)] = sum over the workgroups' partial tiles.  Workgroup = 16 outputs x 16 groups of partials; a group
// adds every sixteenth partial in order, the sixteen sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void k_img_wgrad_reduce(const float* __restrict__ part, int nparts, int64_t total,
                                                           float* __restrict__ dW) {
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + col;
    float s = 0.f;
    if (e < total)
        for (int p = grp; p < nparts; p += 16) s += part[(int64_t)p * total + e];
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && e < total) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        dW[e] = t;
    }
}

// ---- UP ------------------------------------------------------------------------------------------------------------
// Workgroup = an 8 x 32 tile of positions of one image, one thread per position.  The tile's features (with a one-
// position halo, 32 channels at a time) are staged through LDS: global reads are whole 128-byte channel vectors, the
// per-thread reads of one channel at a time come from LDS (position stride 33 words: consecutive positions fall into
// different banks).  A position forms the 2 x 2 output pixels it centres, for every image channel: output row 2i takes
// kernel rows 1 (input row i) and 3 (row i - 1), row 2i + 1 takes kernel rows 0 (row i + 1) and 2 (row i); columns
// likewise.  The weights are uniform over the wave (scalar operands).
constexpr int UP_TR = 8, UP_TC = 32, UP_PS = 33;
template <int CIMG>
__global__ __launch_bounds__(256) void k_img_up(const float* __restrict__ small, const float* __restrict__ Wt,
                                                 const float* __restrict__ bias, float* __restrict__ big, int Cs, int H, int W,
                                                 int tiles_w) {
    __shared__ float sx[(UP_TR + 2) * (UP_TC + 2) * UP_PS];
    const int tid = threadIdx.x;
    const int tile_r = blockIdx.x / tiles_w, tile_c = blockIdx.x - tile_r * tiles_w;
    const int r0 = tile_r * UP_TR, c0 = tile_c * UP_TC;
    const int64_t n = blockIdx.y;
    const float* x = small + n * (int64_t)H * W * Cs;
    const int r = tid >> 5, c = tid & 31;
    const int i = r0 + r, j = c0 + c;
    const bool valid = i < H && j < W;
    float acc[CIMG][2][2];
#pragma unroll
    for (int co = 0; co < CIMG; ++co) {
        const float b = bias ? bias[co] : 0.f;
        acc[co][0][0] = b; acc[co][0][1] = b; acc[co][1][0] = b; acc[co][1][1] = b;
    }
    for (int cb = 0; cb < Cs; cb += 32) {
        if (cb > 0) __syncthreads();                 // every thread has finished with the previous channel block
        // stage (TR + 2) x (TC + 2) positions x 32 channels; positions outside the image hold zeros
        for (int e = tid; e < (UP_TR + 2) * (UP_TC + 2) * 8; e += 256) {
            const int p = e >> 3, q = e & 7;
            const int tr = p / (UP_TC + 2), tc = p - tr * (UP_TC + 2);
            const int gi = r0 - 1 + tr, gj = c0 - 1 + tc;
            const bool in = (unsigned)gi < (unsigned)H && (unsigned)gj < (unsigned)W;
            const f32x4 v = in ? *reinterpret_cast<const f32x4*>(x + ((int64_t)gi * W + gj) * Cs + cb + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            float* d = sx + p * UP_PS + 4 * q;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
        const float* mine = sx + (r * (UP_TC + 2) + c) * UP_PS;      // the (-1, -1) neighbour of this thread's position
        for (int ci = 0; ci < 32; ++ci) {
            float v[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) v[a][b] = mine[(a * (UP_TC + 2) + b) * UP_PS + ci];
            const float* wc = Wt + (int64_t)(cb + ci) * CIMG * 16;   // [cimg][ky][kx], uniform over the wave
#pragma unroll
            for (int co = 0; co < CIMG; ++co) {
                const float* w = wc + co * 16;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ky0 = a == 0 ? 1 : 0, v0 = a == 0 ? 1 : 2;   // (kernel row, index into v = row offset + 1)
                    const int ky1 = a == 0 ? 3 : 2, v1 = a == 0 ? 0 : 1;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int kx0 = b == 0 ? 1 : 0, u0 = b == 0 ? 1 : 2;
                        const int kx1 = b == 0 ? 3 : 2, u1 = b == 0 ? 0 : 1;
                        float t = acc[co][a][b];
                        t = fmaf(v[v0][u0], w[ky0 * 4 + kx0], t);
                        t = fmaf(v[v0][u1], w[ky0 * 4 + kx1], t);
                        t = fmaf(v[v1][u0], w[ky1 * 4 + kx0], t);
                        t = fmaf(v[v1][u1], w[ky1 * 4 + kx1], t);
                        acc[co][a][b] = t;
                    }
                }
            }
        }
    }
    if (valid) {
        const int H2 = 2 * H, W2 = 2 * W;
        float* y = big + n * (int64_t)CIMG * H2 * W2;
#pragma unroll
        for (int co = 0; co < CIMG; ++co)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                *reinterpret_cast<float2*>(y + ((int64_t)co * H2 + 2 * i + a) * W2 + 2 * j) = make_float2(acc[co][a][0], acc[co][a][1]);
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
    int64_t blocks = (nsteps + WG_WAVES * 64 - 1) / (WG_WAVES * 64);        // at least 64 steps per wave
    const int64_t cap = 2 * (int64_t)cu_count_img();        // two workgroups (16 waves) per CU, all resident in one round
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
    DRM_REQUIRE(N * Cimg * 4 * H * W < ((int64_t)1 << 31), DRM_ENOTSUP, "image tensor too large for 32-bit offsets");
    int64_t gx = (ntiles + 3) / 4;
    const int64_t cap = 4 * (int64_t)cu_count_img() / (Cs / 32 > 4 ? 4 : Cs / 32);   // about four resident workgroups per CU over all channel blocks
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
    hipStream_t s = (hipStream_t)stream;
    const int tiles_w = (W + UP_TC - 1) / UP_TC, tiles_h = (H + UP_TR - 1) / UP_TR;
    const dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)N), block(256);
    const double out = (double)N * Cimg * 4.0 * H * W;
    ProfScope prof(PC_ELEMENTWISE, s, 2.0 * out * Cs * 4.0, 4.0 * ((double)N * Cs * H * W + out + 16.0 * Cs * Cimg));
    switch (Cimg) {
        case 1: k_img_up<1><<<grid, block, 0, s>>>(small_in, weight, bias, big_out, Cs, H, W, tiles_w); break;
        case 2: k_img_up<2><<<grid, block, 0, s>>>(small_in, weight, bias, big_out, Cs, H, W, tiles_w); break;
        case 3: k_img_up<3><<<grid, block, 0, s>>>(small_in, weight, bias, big_out, Cs, H, W, tiles_w); break;
        default: k_img_up<4><<<grid, block, 0, s>>>(small_in, weight, bias, big_out, Cs, H, W, tiles_w); break;
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
    DRM_REQUIRE(N * Cimg * 4 * H * W < ((int64_t)1 << 31), DRM_ENOTSUP, "image tensor too large for 32-bit offsets");
    const int blocks = wgrad_blocks(P);
    const int64_t total = (int64_t)Cs * 16 * Cimg;
    DRM_REQUIRE(workspace_bytes >= (size_t)blocks * (size_t)total * sizeof(float), DRM_EINVAL, "workspace too small: need %zu bytes",
                (size_t)blocks * (size_t)total * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nsteps = (P + 1) / 2;
    const int64_t spw = (nsteps + (int64_t)blocks * WG_WAVES - 1) / ((int64_t)blocks * WG_WAVES);
    const dim3 grid((unsigned)blocks, (unsigned)(Cs / 32)), block(64 * WG_WAVES);
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
    k_img_wgrad_reduce<<<dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s>>>(part, blocks, total, grad_weight);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
