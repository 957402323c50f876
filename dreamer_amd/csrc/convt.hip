// ConvTranspose2d(Cin -> Cout <= 4, kernel 4, stride 2, padding 1) + bias, forward: the last decoder stage
// (lib/world_model.py:298-302, 32 channels of 32x32 -> 3 channels of 64x64).
//
// With three output channels the layer is 3.2 GFLOP against 134 MB of input and 50 MB of output at the
// training shape: a stream.  One thread owns one INPUT position (i, j) and forms the 2 x 2 output pixels it
// centres, for every output channel: row 2i takes kernel rows 1 (input row i) and 3 (row i-1), row 2i+1
// takes kernel rows 0 (row i+1) and 2 (row i); columns likewise.  Per input channel that is a 3 x 3
// neighbourhood (neighbouring lanes share cache lines) and 16 * Cout multiply-adds whose weights are
// uniform over the wave (scalar operands).  Stores are 8 bytes per lane, contiguous across lanes.
#include "common.hpp"
#include "../../include/dreamer_amd.h"

namespace drm {

template <int CO>
__global__ __launch_bounds__(256) void k_convt_k4s2_fwd(const float* __restrict__ X, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, float* __restrict__ Y, int Cin,
                                                         int H, int W) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= H * W) return;
    const int i = pos / W, j = pos - i * W;
    const int64_t n = blockIdx.y;
    const float* x = X + n * (int64_t)Cin * H * W;
    const bool up = i > 0, dn = i + 1 < H, lf = j > 0, rt = j + 1 < W;
    float acc[CO][2][2];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float b = bias ? bias[co] : 0.f;
        acc[co][0][0] = b; acc[co][0][1] = b; acc[co][1][0] = b; acc[co][1][1] = b;
    }
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xc = x + (int64_t)ci * H * W + (int64_t)i * W + j;
        // v[di + 1][dj + 1]
        float v[3][3];
        v[0][0] = (up && lf) ? xc[-W - 1] : 0.f; v[0][1] = up ? xc[-W] : 0.f; v[0][2] = (up && rt) ? xc[-W + 1] : 0.f;
        v[1][0] = lf ? xc[-1] : 0.f;             v[1][1] = xc[0];             v[1][2] = rt ? xc[1] : 0.f;
        v[2][0] = (dn && lf) ? xc[W - 1] : 0.f;  v[2][1] = dn ? xc[W] : 0.f;  v[2][2] = (dn && rt) ? xc[W + 1] : 0.f;
        const float* wc = Wt + (int64_t)ci * CO * 16;   // [co][ky][kx], uniform over the wave
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float* w = wc + co * 16;
            // output row parity a: (ky, input row offset) pairs {(1, 0), (3, -1)} for a = 0, {(0, +1), (2, 0)} for a = 1
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ky0 = a == 0 ? 1 : 0, r0 = a == 0 ? 1 : 2;   // index into v: offset + 1
                const int ky1 = a == 0 ? 3 : 2, r1 = a == 0 ? 0 : 1;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int kx0 = b == 0 ? 1 : 0, c0 = b == 0 ? 1 : 2;
                    const int kx1 = b == 0 ? 3 : 2, c1 = b == 0 ? 0 : 1;
                    float s = acc[co][a][b];
                    s = fmaf(v[r0][c0], w[ky0 * 4 + kx0], s);
                    s = fmaf(v[r0][c1], w[ky0 * 4 + kx1], s);
                    s = fmaf(v[r1][c0], w[ky1 * 4 + kx0], s);
                    s = fmaf(v[r1][c1], w[ky1 * 4 + kx1], s);
                    acc[co][a][b] = s;
                }
            }
        }
    }
    const int Ho = 2 * H, Wo = 2 * W;
    float* y = Y + n * (int64_t)CO * Ho * Wo;
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int a = 0; a < 2; ++a)
            *reinterpret_cast<float2*>(y + ((int64_t)co * Ho + 2 * i + a) * Wo + 2 * j) = make_float2(acc[co][a][0], acc[co][a][1]);
}

// Input gradient of the layer: dX[n, i, j, ci] = sum_{co, ky, kx} g[n, co, 2i - 1 + ky, 2j - 1 + kx] W[ci, co, ky, kx], with
// g (N, Cout, 2H, 2W) contiguous and dX written in channels-last memory (N, H, W, Cin) when CL, else (N, Cin, H, W).
// One thread owns one input position: 16 Cout upstream values, then Cin results of 16 Cout multiply-adds each whose
// weights are uniform over the wave.  134 MB written and 50 MB read at the training shape: a stream.
template <int CO, bool CL>
__global__ __launch_bounds__(256) void k_convt_k4s2_dgrad(const float* __restrict__ G, const float* __restrict__ Wt,
                                                           float* __restrict__ dX, int Cin, int H, int W) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= H * W) return;
    const int i = pos / W, j = pos - i * W;
    const int64_t n = blockIdx.y;
    const int Ho = 2 * H, Wo = 2 * W;
    const float* g = G + n * (int64_t)CO * Ho * Wo;
    float u[CO][4][4];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int y = 2 * i - 1 + ky;
            const bool yok = (unsigned)y < (unsigned)Ho;
            const float* row = g + ((int64_t)co * Ho + (yok ? y : 0)) * Wo + 2 * j;
            // columns 2j - 1 .. 2j + 2: the middle two are one aligned 8-byte access
            const float2 mid = yok ? *reinterpret_cast<const float2*>(row) : make_float2(0.f, 0.f);
            u[co][ky][0] = (yok && j > 0) ? row[-1] : 0.f;
            u[co][ky][1] = mid.x; u[co][ky][2] = mid.y;
            u[co][ky][3] = (yok && j + 1 < W) ? row[2] : 0.f;
        }
    float* out = CL ? dX + (n * H * W + pos) * (int64_t)Cin : dX + n * (int64_t)Cin * H * W + pos;
    for (int c4 = 0; c4 < Cin; c4 += 4) {
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* w = Wt + (int64_t)min(c4 + q, Cin - 1) * CO * 16;     // [co][ky][kx], uniform over the wave
            float s = 0.f;
#pragma unroll
            for (int co = 0; co < CO; ++co)
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) s = fmaf(u[co][ky][kx], w[co * 16 + ky * 4 + kx], s);
            r[q] = s;
        }
        if (CL) {
            *reinterpret_cast<f32x4*>(out + c4) = f32x4{r[0], r[1], r[2], r[3]};      // Cin % 4 == 0
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c4 + q < Cin) out[(int64_t)(c4 + q) * H * W] = r[q];
        }
    }
}

}  // namespace drm

extern "C" int drm_convt_k4s2_input_grad(const float* grad_y, const float* weight, float* grad_x, int64_t N, int32_t Cin,
                                         int32_t Cout, int32_t H, int32_t W, int32_t channels_last, drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(grad_y && weight && grad_x, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(N >= 0 && N < 65536 && Cin >= 1 && H >= 1 && W >= 1, DRM_EINVAL, "bad shape");
    DRM_REQUIRE(Cout >= 1 && Cout <= 4, DRM_ENOTSUP, "built for at most 4 output channels (got %d)", Cout);
    DRM_REQUIRE(!channels_last || (Cin % 4 == 0 && aligned16(grad_x)), DRM_EINVAL, "channels-last output needs a channel count that is a multiple of 4 and 16-byte alignment");
    DRM_REQUIRE((reinterpret_cast<uintptr_t>(grad_y) & 7u) == 0, DRM_EINVAL, "grad_y must be 8-byte aligned");
    if (N == 0) return DRM_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)N), block(256);
    const double up = (double)N * Cout * 4.0 * H * W;
    ProfScope prof(PC_ELEMENTWISE, s, 2.0 * up * Cin * 4.0, 4.0 * ((double)N * Cin * H * W + up + 16.0 * Cin * Cout));
#define DRM_DGRAD(CO)                                                                                                   \
    if (channels_last) k_convt_k4s2_dgrad<CO, true><<<grid, block, 0, s>>>(grad_y, weight, grad_x, Cin, H, W);          \
    else k_convt_k4s2_dgrad<CO, false><<<grid, block, 0, s>>>(grad_y, weight, grad_x, Cin, H, W);
    switch (Cout) {
        case 1: DRM_DGRAD(1) break;
        case 2: DRM_DGRAD(2) break;
        case 3: DRM_DGRAD(3) break;
        default: DRM_DGRAD(4) break;
    }
#undef DRM_DGRAD
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

extern "C" int drm_convt_k4s2_forward(const float* x, const float* weight, const float* bias, float* y, int64_t N, int32_t Cin,
                                      int32_t Cout, int32_t H, int32_t W, drm_stream_t stream) {
    using namespace drm;
    DRM_REQUIRE(x && weight && y, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(N >= 0 && N < 65536 && Cin >= 1 && H >= 1 && W >= 1, DRM_EINVAL, "bad shape");
    DRM_REQUIRE(Cout >= 1 && Cout <= 4, DRM_ENOTSUP, "built for at most 4 output channels (got %d)", Cout);
    DRM_REQUIRE((reinterpret_cast<uintptr_t>(y) & 7u) == 0, DRM_EINVAL, "output must be 8-byte aligned");
    if (N == 0) return DRM_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)N), block(256);
    const double out = (double)N * Cout * 4.0 * H * W;
    ProfScope prof(PC_ELEMENTWISE, s, 2.0 * out * Cin * 4.0, 4.0 * ((double)N * Cin * H * W + out + 16.0 * Cin * Cout));
    switch (Cout) {
        case 1: k_convt_k4s2_fwd<1><<<grid, block, 0, s>>>(x, weight, bias, y, Cin, H, W); break;
        case 2: k_convt_k4s2_fwd<2><<<grid, block, 0, s>>>(x, weight, bias, y, Cin, H, W); break;
        case 3: k_convt_k4s2_fwd<3><<<grid, block, 0, s>>>(x, weight, bias, y, Cin, H, W); break;
        default: k_convt_k4s2_fwd<4><<<grid, block, 0, s>>>(x, weight, bias, y, Cin, H, W); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}
