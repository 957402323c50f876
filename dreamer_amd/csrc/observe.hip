// Observe (posterior) recurrence over a replayed (B, T) batch.
//
// Few sequences (B <= 64, the training batch): two launches per time step.
//   k_observe_gates : draw z_{t-1} from the previous step's posterior logits, episode-boundary reset, then
//                     the GRU cell.  x_t = lin_b + sum_l lin_zT[z_l] + c a_tab[a] enters the cell only
//                     through W_ih x_t, and z is one-hot, so gi is a sum of rows of the packed table
//                     gz = lin_zT W_ih^T (plus c * ga[a], ga = a_tab W_ih^T): no matrix product.
//   k_skinny        : posterior logits = e_part + W_fc_h h_t  and  gh_raw = W_hh h_t  (both read h_t)
//   x_t itself (saved for the weight gradients) is formed for all steps at once after the loop.
// More sequences: three launches per time step.
//   k_observe_reset_x : episode-boundary reset, one-hot row gather of the GRU input x
//   k_linear16<3,GRU> : gi = W_ih x on MFMA + gate combine with the precomputed W_hh h  -> h_t
//   k_linear16<2,SAMPLE> (+ companion store problem): posterior logits = e_part + W_fc_h h_t, unimix,
//                       Gumbel-argmax draw of z_t; and gh_raw = W_hh h_t for the next step
#include "launch.hpp"

#include <cstring>
#include <mutex>
#include <vector>

namespace drm {

struct ResetArgs {
    const float* h_prev; int64_t ldh;
    const uint8_t* z_prev; int64_t ldz;
    const int64_t* a_prev; int64_t lda;
    const float* c_prev; int64_t ldc;
    const float* expo;                     // (B, L*32) reset noise of this step
    const float* lin_zT; const float* lin_b; const float* a_tab;
    int D, L, classes;
    uint8_t* z_in; int64_t ldzi;
    float* h_in; int64_t ldhi;
    float* x; int64_t ldxo;
};

// Workgroup (sequence, slice): a thread owns columns 512 y + tid, + 256 of every gridDim.y-th 512-column slice (every
// slice's workgroup repeats the small reset draw rather than wait for another one; slice 0 writes it out)
template <int LMAX>
__global__ __launch_bounds__(256) void k_observe_reset_x(const ResetArgs P) {
    __shared__ uint8_t s_z[LMAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float c = P.c_prev[(int64_t)b * P.ldc];
    const int a = (int)P.a_prev[(int64_t)b * P.lda];
    if (c != 0.f) {
        if (tid < P.L) s_z[tid] = P.z_prev[(int64_t)b * P.ldz + tid];
    } else {
        // fresh uniform one-hot: argmax softmax(1/K - log E), eight lanes per latent
        const int sub = tid & 7;
        for (int l = tid >> 3; l < P.L; l += 32) {
            const float4 ex = *reinterpret_cast<const float4*>(P.expo + ((int64_t)b * P.L + l) * 32 + sub * 4);
            const float e[4] = {ex.x, ex.y, ex.z, ex.w};
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (1.0f / 32.0f) - logf(e[j]);
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
            if (sub == 0) s_z[l] = (uint8_t)bi;
        }
    }
    __syncthreads();
    if (tid < P.L && blockIdx.y == 0) P.z_in[(int64_t)b * P.ldzi + tid] = s_z[tid];
    for (int i0 = blockIdx.y * 512; i0 < P.D; i0 += 512 * gridDim.y) {
        const int c0 = i0 + tid, c1 = i0 + tid + 256;
        const bool jn0 = c0 < P.D, jn1 = c1 < P.D;
        const float hp0 = jn0 ? P.h_prev[(int64_t)b * P.ldh + c0] : 0.f, hp1 = jn1 ? P.h_prev[(int64_t)b * P.ldh + c1] : 0.f;
        const float at0 = jn0 ? P.a_tab[(int64_t)a * P.D + c0] : 0.f, at1 = jn1 ? P.a_tab[(int64_t)a * P.D + c1] : 0.f;
        float xz0 = jn0 ? P.lin_b[c0] : 0.f, xz1 = jn1 ? P.lin_b[c1] : 0.f;
        float u0[LMAX], u1[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            const bool ok = l < P.L;
            const int64_t rowoff = ok ? (int64_t)(l * P.classes + s_z[l]) * P.D : 0;
            u0[l] = (ok && jn0) ? P.lin_zT[rowoff + c0] : 0.f;
            u1[l] = (ok && jn1) ? P.lin_zT[rowoff + c1] : 0.f;
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { xz0 += u0[l]; xz1 += u1[l]; }
        if (jn0) { P.x[(int64_t)b * P.ldxo + c0] = xz0 + c * at0; P.h_in[(int64_t)b * P.ldhi + c0] = c * hp0; }
        if (jn1) { P.x[(int64_t)b * P.ldxo + c1] = xz1 + c * at1; P.h_in[(int64_t)b * P.ldhi + c1] = c * hp1; }
    }
}

// ---- wide path (recurrent width > 512, any number of sequences) ------------------------------------
// The matrix products run on the large-row kernel (launch_big); what surrounds them is elementwise.
// GRU cell of one step over (B, D): gi = W_ih x (no bias), gh_raw = W_hh h_{t-1} (no bias, not yet scaled by the
// continue flag: W_hh (c h) = c W_hh h); writes h_t and the saved gate values r, u, n, gh_n.
struct WideGatesArgs {
    const float* gi; const float* gh_raw;      // (B, 3D) contiguous
    const float* b_ih; const float* b_hh;      // (3D)
    const float* c_prev; int64_t ldc;          // (B)
    const float* h_prev; int64_t ldh;
    float* h_out; int64_t ldho;
    float* gates_out; int64_t ldgo;            // (B, 4D)
    int B, D;
};
__global__ __launch_bounds__(256) void k_observe_gates_wide(const WideGatesArgs P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)P.B * P.D) return;
    const int D = P.D, b = (int)(idx / D), j = (int)(idx - (int64_t)b * D);
    const float c = P.c_prev[(int64_t)b * P.ldc];
    const float* gi = P.gi + (int64_t)b * 3 * D;
    const float* gh = P.gh_raw + (int64_t)b * 3 * D;
    const float ghr = gh[j] * c + P.b_hh[j], ghu = gh[D + j] * c + P.b_hh[D + j], ghn = gh[2 * D + j] * c + P.b_hh[2 * D + j];
    const float hp = P.h_prev[(int64_t)b * P.ldh + j] * c;
    const float rr = sigmoid_f((gi[j] + P.b_ih[j]) + ghr);
    const float uu = sigmoid_f((gi[D + j] + P.b_ih[D + j]) + ghu);
    const float nn = tanhf((gi[2 * D + j] + P.b_ih[2 * D + j]) + rr * ghn);
    P.h_out[(int64_t)b * P.ldho + j] = (1.0f - uu) * nn + uu * hp;
    float* go = P.gates_out + (int64_t)b * P.ldgo;
    go[j] = rr; go[D + j] = uu; go[2 * D + j] = nn; go[3 * D + j] = ghn;
}

// Posterior logits = raw (W_fc_h h_t) + e_part, then the draw of z_t (eight lanes per latent): writes the logits, the
// soft sample probabilities y = q / sum q with q_k = ((1 - eps) softmax_k + eps / K) / E_k, the index and the one-hot.
struct WideSampleArgs {
    const float* raw; int64_t ldraw;           // (B, L*32)
    const float* e_part; int64_t ldep;
    const float* expo; int64_t lde;
    float* post_out; int64_t ldpo;
    float* y_out; int64_t ldy;
    uint8_t* zidx_out; int64_t ldzo;
    float* z_dense; int64_t ldzd;              // or nullptr
    int B, L;
    float one_m_eps, eps_over_k;
};
__global__ __launch_bounds__(256) void k_observe_sample_wide(const WideSampleArgs P) {
    const int sub = threadIdx.x & 7, k0 = sub * 4;
    const int64_t total = (int64_t)P.B * P.L;
    for (int64_t it = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); it < total; it += (int64_t)gridDim.x * 32) {
        const int b = (int)(it / P.L), l = (int)(it - (int64_t)b * P.L);
        const float4 rw = *reinterpret_cast<const float4*>(P.raw + (int64_t)b * P.ldraw + l * 32 + k0);
        const float4 ep = *reinterpret_cast<const float4*>(P.e_part + (int64_t)b * P.ldep + l * 32 + k0);
        const float4 ex = *reinterpret_cast<const float4*>(P.expo + (int64_t)b * P.lde + l * 32 + k0);
        const float v[4] = {rw.x + ep.x, rw.y + ep.y, rw.z + ep.z, rw.w + ep.w};
        const float e[4] = {ex.x, ex.y, ex.z, ex.w};
        const float mx = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float pj[4], s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pj[i] = expf(v[i] - mx); s += pj[i]; }
        const float scale = P.one_m_eps / group_sum<8>(s);
        float q[4], qs = 0.f, best = -1.f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[i] = fmaf(pj[i], scale, P.eps_over_k) / e[i];
            qs += q[i];
            if (q[i] > best) { best = q[i]; bi = k0 + i; }
        }
        qs = group_sum<8>(qs);
        group_argmax<8>(best, bi);
        *reinterpret_cast<float4*>(P.post_out + (int64_t)b * P.ldpo + l * 32 + k0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(P.y_out + (int64_t)b * P.ldy + l * 32 + k0) = make_float4(q[0] / qs, q[1] / qs, q[2] / qs, q[3] / qs);
        if (sub == 0) P.zidx_out[(int64_t)b * P.ldzo + l] = (uint8_t)bi;
        if (P.z_dense)
            *reinterpret_cast<float4*>(P.z_dense + (int64_t)b * P.ldzd + l * 32 + k0) =
                make_float4(bi == k0 ? 1.f : 0.f, bi == k0 + 1 ? 1.f : 0.f, bi == k0 + 2 ? 1.f : 0.f, bi == k0 + 3 ? 1.f : 0.f);
    }
}

// GRU cell backward of one step over (B, D):
//     dh = dh_fc + g_h + c_{t+1} (dhu_next + dhw_next)      (dh_fc = d(post)_t W_fc_h, dhw = d(gh)_{t+1} W_hh)
// then d(gi), d(gh) from the saved gates, and dhu = dh * u for the step before.
struct WideGruBwdArgs {
    const float* dh_fc;                        // (B, D) contiguous
    const float* g_h; int64_t ldgh;
    const float* dhu_next; const float* dhw_next;   // (B, D) contiguous; dhu_next == nullptr at the last step
    const float* next_scale; int64_t ldns;
    const float* gates; int64_t ldg;           // (B, 4D)
    const float* h_in; int64_t ldhi;
    float* dgi; float* dgh; int64_t lddg;      // (B, 3D)
    float* dhu;                                // (B, D) contiguous (may alias dhu_next)
    int B, D;
};
__global__ __launch_bounds__(256) void k_observe_gru_bwd_wide(const WideGruBwdArgs P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)P.B * P.D) return;
    const int D = P.D, m = (int)(idx / D), n = (int)(idx - (int64_t)m * D);
    float dh = P.dh_fc[idx] + P.g_h[(int64_t)m * P.ldgh + n];
    if (P.dhu_next != nullptr) dh += P.next_scale[(int64_t)m * P.ldns] * (P.dhu_next[idx] + P.dhw_next[idx]);
    const float* gt = P.gates + (int64_t)m * P.ldg;
    const float rr = gt[n], uu = gt[D + n], nn = gt[2 * D + n], ghn = gt[3 * D + n];
    const float hin = P.h_in[(int64_t)m * P.ldhi + n];
    const float dn_pre = dh * (1.0f - uu) * (1.0f - nn * nn);
    const float du_pre = dh * (hin - nn) * uu * (1.0f - uu);
    const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
    float* gi = P.dgi + (int64_t)m * P.lddg;
    float* gh = P.dgh + (int64_t)m * P.lddg;
    gi[n] = dr_pre; gi[D + n] = du_pre; gi[2 * D + n] = dn_pre;
    gh[n] = dr_pre; gh[D + n] = du_pre; gh[2 * D + n] = dn_pre * rr;
    P.dhu[idx] = dh * uu;
}

// ---- few-sequence path ---------------------------------------------------------------------------
struct ObsGatesArgs {
    // draw of z_{t-1} (post == nullptr at t = 0: z_prev0 is given)
    const float* post; int64_t ldpost;        // (B, L*32) posterior logits of step t-1
    const float* expo_z; int64_t ldez;        // (B, L*32) Exp(1) noise of that draw
    float* y_out; int64_t ldy;                // (B, L*32) soft sample probabilities of step t-1
    uint8_t* zidx_out; int64_t ldzo;          // (B, L)
    float* z_dense; int64_t ldzd;             // (B, L*32) or nullptr
    const uint8_t* z_prev0; int64_t ldz0;
    float one_m_eps, eps_over_k;
    // episode-boundary reset of step t
    const float* c_prev; int64_t ldc;
    const float* expo_reset; int64_t lder;    // (B, L*32)
    const int64_t* a_prev; int64_t lda;
    uint8_t* z_in_out; int64_t ldzi;          // (B, L) latents fed into step t
    // GRU cell
    const float* gz; const float* ga;         // (L*32, 3D), (A, 3D)
    const float* b_ih; const float* b_hh;     // (3D); b_ih here = W_ih lin_b + b_ih (the packed gb)
    const float* gh_raw; int64_t ldgh;        // (B, 3D) = W_hh h_{t-1}
    const float* h_prev; int64_t ldh;
    float* h_out; int64_t ldho;
    float* h_in_out; int64_t ldhi;            // c * h_{t-1}
    float* gates_out; int64_t ldgo;           // (B, 4D): r, u, n, gh_n
    int D, L, do_step;                        // do_step == 0: only the draw (after the last step)
};

// grid (ceil(D / 32), B): workgroup = 32 hidden units of one sequence.  Every workgroup of a sequence
// repeats the draw (1024 logits, four per thread) rather than wait for another workgroup.
__global__ __launch_bounds__(256) void k_observe_gates(const ObsGatesArgs P) {
    __shared__ uint8_t s_z[64];
    __shared__ __attribute__((aligned(16))) float s_red[3][32][32];
    __shared__ float s_gi[3][32];
    const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    const int sub = tid & 7;
    const int D = P.D;
    const float c = P.do_step ? P.c_prev[(int64_t)b * P.ldc] : 1.f;
    // operands of the gate arithmetic (threads 0..31), requested before anything else
    const int j = slice * 32 + (tid & 31);
    const bool gate_thread = P.do_step && tid < 32 && j < D;
    float ghr = 0.f, ghu = 0.f, ghn = 0.f, bir = 0.f, biu = 0.f, bin_ = 0.f, gar = 0.f, gau = 0.f, gan = 0.f, hp = 0.f;
    if (gate_thread) {
        const float* gh = P.gh_raw + (int64_t)b * P.ldgh;
        const float* ga = P.ga + (int64_t)P.a_prev[(int64_t)b * P.lda] * 3 * D;
        ghr = gh[j] * c + P.b_hh[j]; ghu = gh[D + j] * c + P.b_hh[D + j]; ghn = gh[2 * D + j] * c + P.b_hh[2 * D + j];
        bir = P.b_ih[j]; biu = P.b_ih[D + j]; bin_ = P.b_ih[2 * D + j];
        gar = ga[j]; gau = ga[D + j]; gan = ga[2 * D + j];
        hp = P.h_prev[(int64_t)b * P.ldh + j] * c;
    }
    for (int l = tid >> 3; l < P.L; l += 32) {
        int zs;
        if (P.post != nullptr) {
            // y = softmax(log_unimix(post) - log E) = q / sum q with q_k = ((1-eps) softmax(post)_k + eps/K) / E_k
            const float4 lg = *reinterpret_cast<const float4*>(P.post + (int64_t)b * P.ldpost + l * 32 + sub * 4);
            const float4 ex = *reinterpret_cast<const float4*>(P.expo_z + (int64_t)b * P.ldez + l * 32 + sub * 4);
            const float v[4] = {lg.x, lg.y, lg.z, lg.w};
            const float e[4] = {ex.x, ex.y, ex.z, ex.w};
            const float mx = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            float pj[4], s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { pj[i] = expf(v[i] - mx); s += pj[i]; }
            const float scale = P.one_m_eps / group_sum<8>(s);
            float q[4], qs = 0.f, best = -1.f;
            int bi = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i] = fmaf(pj[i], scale, P.eps_over_k) / e[i];
                qs += q[i];
                if (q[i] > best) { best = q[i]; bi = sub * 4 + i; }
            }
            qs = group_sum<8>(qs);
            group_argmax<8>(best, bi);
            zs = bi;
            if (slice == 0) {
                const int k0 = sub * 4;
                *reinterpret_cast<float4*>(P.y_out + (int64_t)b * P.ldy + l * 32 + k0) =
                    make_float4(q[0] / qs, q[1] / qs, q[2] / qs, q[3] / qs);
                if (sub == 0) P.zidx_out[(int64_t)b * P.ldzo + l] = (uint8_t)bi;
                if (P.z_dense)
                    *reinterpret_cast<float4*>(P.z_dense + (int64_t)b * P.ldzd + l * 32 + k0) =
                        make_float4(bi == k0 ? 1.f : 0.f, bi == k0 + 1 ? 1.f : 0.f, bi == k0 + 2 ? 1.f : 0.f, bi == k0 + 3 ? 1.f : 0.f);
            }
        } else {
            zs = P.z_prev0[(int64_t)b * P.ldz0 + l];
        }
        if (P.do_step && c == 0.f) {
            // fresh uniform one-hot (lib/world_model.py:57-61): argmax softmax(1/K - log E) = the class with the smallest E
            const float4 ex = *reinterpret_cast<const float4*>(P.expo_reset + (int64_t)b * P.lder + l * 32 + sub * 4);
            const float e[4] = {ex.x, ex.y, ex.z, ex.w};
            float best = -3.0e38f;
            int bi = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (-e[i] > best) { best = -e[i]; bi = sub * 4 + i; }
            group_argmax<8>(best, bi);
            zs = bi;
        }
        if (sub == 0) {
            s_z[l] = (uint8_t)zs;
            if (slice == 0 && P.do_step) P.z_in_out[(int64_t)b * P.ldzi + l] = (uint8_t)zs;
        }
    }
    if (!P.do_step) return;
    __syncthreads();
    // gi slice = sum over latents of table rows: thread (latent slot, 4 columns), three gates
    {
        const int ls = tid >> 3, c4 = sub * 4;
        const int jj = slice * 32 + c4;
        float4 a[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) a[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jj < D) {
            for (int l = ls; l < P.L; l += 32) {
                const float* row = P.gz + (int64_t)(l * 32 + s_z[l]) * 3 * D + jj;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float4 u = *reinterpret_cast<const float4*>(row + g * D);
                    a[g].x += u.x; a[g].y += u.y; a[g].z += u.z; a[g].w += u.w;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) *reinterpret_cast<float4*>(&s_red[g][ls][c4]) = a[g];
    }
    __syncthreads();
    if (tid < 96) {
        const int g = tid >> 5, jl = tid & 31;
        float a = 0.f;
        const int live = P.L < 32 ? P.L : 32;
        for (int i = 0; i < live; ++i) a += s_red[g][i][jl];
        s_gi[g][jl] = a;
    }
    __syncthreads();
    if (!gate_thread) return;
    const int jl = tid & 31;
    const float rr = sigmoid_f(((s_gi[0][jl] + c * gar) + bir) + ghr);
    const float uu = sigmoid_f(((s_gi[1][jl] + c * gau) + biu) + ghu);
    const float nn = tanhf(((s_gi[2][jl] + c * gan) + bin_) + rr * ghn);
    P.h_out[(int64_t)b * P.ldho + j] = (1.0f - uu) * nn + uu * hp;
    P.h_in_out[(int64_t)b * P.ldhi + j] = hp;
    float* go = P.gates_out + (int64_t)b * P.ldgo;
    go[j] = rr; go[D + j] = uu; go[2 * D + j] = nn; go[3 * D + j] = ghn;
}

// x[r] = lin_b + sum_l lin_zT[l*32 + z_in[r, l]] + c[r] * a_tab[a[r]] for all r = (b, t) rows at once
// (the GRU input, needed only by the weight gradients); one workgroup per row, D <= 512.
struct XBulkArgs {
    const uint8_t* z_in; const int64_t* a_prev; const float* c_prev;
    const float* lin_zT; const float* lin_b; const float* a_tab;
    float* x; int D, L;
};
__global__ __launch_bounds__(256) void k_observe_x_bulk(const XBulkArgs P) {
    __shared__ uint8_t s_z[64];
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < P.L) s_z[tid] = P.z_in[r * P.L + tid];
    const float c = P.c_prev[r];
    const int a = (int)P.a_prev[r];
    __syncthreads();
    for (int i = tid; i < P.D; i += 256) {
        float acc = P.lin_b[i];
        for (int l0 = 0; l0 < P.L; l0 += 8) {
            float u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = (l0 + q < P.L) ? P.lin_zT[(int64_t)((l0 + q) * 32 + s_z[l0 + q]) * P.D + i] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += u[q];
        }
        P.x[r * P.D + i] = acc + c * P.a_tab[(int64_t)a * P.D + i];
    }
}

// Straight-through sample / unimix / log-softmax backward of one step, one workgroup per sequence:
//     dz = scale * dz_raw + g_z;  s1 = sum_k y dz;  dl = y (dz - s1) wmix;  s2 = sum_k dl;
//     d(post) = dl - psoft s2 + g_post          (sums over the 32 classes of a latent: 8 lanes x 4)
struct StBwdArgs {
    const float* dz_raw; int64_t lddz;       // (B, L*32) or nullptr (last step: no path through z)
    const float* scale; int64_t lds;         // (B) continue flag of step t+1
    const float* g_z; const float* y; const float* wmix; const float* psoft; const float* g_post; int64_t ldl;
    float* dpost; int64_t lddp;
    int L;
};
__global__ __launch_bounds__(256) void k_observe_st_bwd(const StBwdArgs P) {
    const int b = blockIdx.x, tid = threadIdx.x, sub = tid & 7;
    const float sc = P.dz_raw ? P.scale[(int64_t)b * P.lds] : 0.f;
    for (int l = tid >> 3; l < P.L; l += 32) {
        const int64_t o = (int64_t)b * P.ldl + l * 32 + sub * 4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 dr = P.dz_raw ? *reinterpret_cast<const float4*>(P.dz_raw + (int64_t)b * P.lddz + l * 32 + sub * 4) : z4;
        const float4 gz = *reinterpret_cast<const float4*>(P.g_z + o);
        const float4 yy = *reinterpret_cast<const float4*>(P.y + o);
        const float4 wm = *reinterpret_cast<const float4*>(P.wmix + o);
        const float4 ps = *reinterpret_cast<const float4*>(P.psoft + o);
        const float4 gp = *reinterpret_cast<const float4*>(P.g_post + o);
        const float dz[4] = {dr.x * sc + gz.x, dr.y * sc + gz.y, dr.z * sc + gz.z, dr.w * sc + gz.w};
        const float yv[4] = {yy.x, yy.y, yy.z, yy.w}, wv[4] = {wm.x, wm.y, wm.z, wm.w};
        const float pv[4] = {ps.x, ps.y, ps.z, ps.w}, gv[4] = {gp.x, gp.y, gp.z, gp.w};
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s1 += yv[i] * dz[i];
        s1 = group_sum<8>(s1);
        float dl[4], s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dl[i] = yv[i] * (dz[i] - s1) * wv[i]; s2 += dl[i]; }
        s2 = group_sum<8>(s2);
        float4 out;
        out.x = dl[0] - pv[0] * s2 + gv[0]; out.y = dl[1] - pv[1] * s2 + gv[1];
        out.z = dl[2] - pv[2] * s2 + gv[2]; out.w = dl[3] - pv[3] * s2 + gv[3];
        *reinterpret_cast<float4*>(P.dpost + (int64_t)b * P.lddp + l * 32 + sub * 4) = out;
    }
}

// defined in imagine.hip's translation unit as well; both are file-local
__global__ __launch_bounds__(256) void k_transpose_cols_o(const float* __restrict__ W, int64_t ldw, int col0, int ncols,
                                                           int nrows, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, c = c0 + tx;
        tile[i][tx] = (n < nrows && c < ncols) ? W[(int64_t)n * ldw + col0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, n = n0 + tx;
        if (c < ncols && n < nrows) out[(int64_t)c * nrows + n] = tile[tx][i];
    }
}

__global__ __launch_bounds__(256) void k_action_table_o(const float* __restrict__ lin_w, int64_t ldw, int col0, int E,
                                                         const float* __restrict__ a_emb, int A, int D,
                                                         float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int a = blockIdx.y;
    if (j >= D) return;
    const float* w = lin_w + (int64_t)j * ldw + col0;
    const float* e = a_emb + (int64_t)a * E;
    float s = 0.f;
    for (int i = 0; i < E; ++i) s = fmaf(w[i], e[i], s);
    out[(int64_t)a * D + j] = s;
}

static int check_observe_dims(const drm_dims& d) {
    DRM_REQUIRE(d.K == 32, DRM_ENOTSUP, "classes per latent must be 32 (got %d)", d.K);
    DRM_REQUIRE(d.D >= 32 && d.D % 32 == 0 && d.D <= 4096, DRM_ENOTSUP, "D=%d must be a multiple of 32, at most 4096", d.D);
    DRM_REQUIRE(d.L >= 1 && d.L <= MAX_LATENTS, DRM_ENOTSUP, "latents=%d (max %d)", d.L, MAX_LATENTS);
    DRM_REQUIRE(d.A >= 1 && d.E >= 1, DRM_EINVAL, "A=%d E=%d", d.A, d.E);
    return DRM_OK;
}

struct ObsPack { size_t lin_zT, a_tab, gz, ga, gb, total; };
static ObsPack obs_pack_layout(const drm_dims& d) {
    ObsPack p{};
    auto up4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
    p.lin_zT = 0;
    p.a_tab = (size_t)d.L * d.K * d.D;
    p.gz = up4(p.a_tab + (size_t)d.A * d.D);
    p.ga = p.gz + (size_t)d.L * d.K * 3 * d.D;
    p.gb = up4(p.ga + (size_t)d.A * 3 * d.D);
    p.total = up4(p.gb + (size_t)3 * d.D);
    return p;
}

static int observe_enqueue(const drm_observe_weights* w, const drm_observe_io* io, hipStream_t s) {
    if (int rc = check_observe_dims(w->dims)) return rc;
    prof_fresh();
    const drm_dims& d = w->dims;
    const int B = io->B, T = io->T, D = d.D, L = d.L, LK = d.L * d.K;
    DRM_REQUIRE(B >= 1 && T >= 1, DRM_EINVAL, "B=%d T=%d", B, T);
    DRM_REQUIRE(w->lin_b && w->gru_w_ih && w->gru_w_hh && w->gru_b_ih && w->gru_b_hh && w->fc_w_h, DRM_EINVAL, "null weight");
    const ObsPack pl = obs_pack_layout(d);
    DRM_REQUIRE(w->pack && w->pack_bytes >= pl.total * sizeof(float), DRM_EINVAL, "weights not packed (drm_pack_observe)");
    DRM_REQUIRE(io->h0 && io->zidx0 && io->a_prev && io->c_prev && io->expo_reset && io->expo_z && io->e_part && io->h_out &&
                    io->zidx_out && io->zidx_in_out && io->post_out && io->y_out && io->gates_out && io->x_out && io->h_in_out,
                DRM_EINVAL, "null io pointer");
    DRM_REQUIRE(io->workspace && io->workspace_bytes >= (size_t)B * 3 * D * sizeof(float) && aligned16(io->workspace),
                DRM_EINVAL, "workspace too small or misaligned");
    DRM_REQUIRE(aligned16(io->expo_reset) && aligned16(io->expo_z) && aligned16(io->e_part), DRM_EINVAL, "misaligned input");
    float* ghraw = (float*)io->workspace;
    const float l1m = (float)log1p(-(double)d.unimix_eps);
    const float lek = (float)(log((double)d.unimix_eps) - log((double)d.K));
    const int64_t TD = (int64_t)T * D, TL = (int64_t)T * L, TLK = (int64_t)T * LK;

    if (D > 512) {
        // ---- wide recurrent state: per step reset + x gather, gi = W_ih x, gates, (W_fc_h h || W_hh h), draw
        DRM_REQUIRE(io->precision == 0 || io->precision == 1, DRM_EINVAL, "precision=%d", io->precision);
        const size_t act_floats = ((size_t)B * (6 * (size_t)D + LK) + 3) & ~(size_t)3;
        DRM_REQUIRE(io->workspace_bytes >= act_floats * sizeof(float), DRM_EINVAL, "workspace too small");
        const int prec = io->precision;
        float* gi = ghraw + (size_t)B * 3 * D;
        float* praw = gi + (size_t)B * 3 * D;
        // bf16 products: the three weight matrices are rounded once per call into the rest of the workspace (when it
        // was sized by drm_observe_workspace_bytes) and every step reads those images, half the bytes
        struct ImageScope { ~ImageScope() { big_image_cache_end(); } } image_scope;
        if (prec == 1 && std::getenv("DRM_NO_WEIGHT_IMAGES") == nullptr)
            big_image_cache_begin(ghraw + act_floats, io->workspace_bytes - act_floats * sizeof(float), 1);
        {
            ProblemSpec p{io->h0, D, nullptr, nullptr, w->gru_w_hh, D, nullptr, nullptr, 0, ghraw, 3 * D, 3 * D, D};
            if (int rc = launch_big(&p, 1, B, nullptr, 0, 0, 0, s, prec)) return rc;
        }
        for (int t = 0; t < T; ++t) {
            const float* h_prev = t == 0 ? io->h0 : io->h_out + (size_t)(t - 1) * D;
            const int64_t ldh = t == 0 ? D : TD;
            float* h_t = io->h_out + (size_t)t * D;
            {
                ResetArgs a{};
                a.h_prev = h_prev; a.ldh = ldh;
                a.z_prev = t == 0 ? io->zidx0 : io->zidx_out + (size_t)(t - 1) * L; a.ldz = t == 0 ? L : TL;
                a.a_prev = io->a_prev + t; a.lda = T; a.c_prev = io->c_prev + t; a.ldc = T;
                a.expo = io->expo_reset + (size_t)t * B * LK;
                a.lin_zT = w->pack + pl.lin_zT; a.lin_b = w->lin_b; a.a_tab = w->pack + pl.a_tab;
                a.D = D; a.L = L; a.classes = d.K;
                a.z_in = io->zidx_in_out + (size_t)t * L; a.ldzi = TL;
                a.h_in = io->h_in_out + (size_t)t * D; a.ldhi = TD; a.x = io->x_out + (size_t)t * D; a.ldxo = TD;
                ProfScope prof(PC_ACTOR_HEAD, s, (double)B * D * (L + 2), 4.0 * (double)B * D * (L + 4));
                if (L <= 32) k_observe_reset_x<32><<<dim3(B, (D + 511) / 512), dim3(256), 0, s>>>(a);
                else k_observe_reset_x<64><<<dim3(B, (D + 511) / 512), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
            {
                ProblemSpec p{io->x_out + (size_t)t * D, TD, nullptr, nullptr, w->gru_w_ih, D, nullptr, nullptr, 0, gi, 3 * D, 3 * D, D};
                if (int rc = launch_big(&p, 1, B, nullptr, 0, 0, 0, s, prec)) return rc;
            }
            {
                WideGatesArgs a{gi, ghraw, w->gru_b_ih, w->gru_b_hh, io->c_prev + t, T, h_prev, ldh, h_t, TD,
                                io->gates_out + (size_t)t * 4 * D, (int64_t)T * 4 * D, B, D};
                ProfScope prof(PC_ELEMENTWISE, s, 20.0 * B * D, 4.0 * 12.0 * B * D);
                k_observe_gates_wide<<<dim3((unsigned)(((int64_t)B * D + 255) / 256)), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
            {
                ProblemSpec p[2] = {
                    {h_t, TD, nullptr, nullptr, w->fc_w_h, w->ld_fc, nullptr, nullptr, 0, praw, LK, LK, D},
                    {h_t, TD, nullptr, nullptr, w->gru_w_hh, D, nullptr, nullptr, 0, ghraw, 3 * D, 3 * D, D}};
                if (int rc = launch_big(p, t + 1 < T ? 2 : 1, B, nullptr, 0, 0, 0, s, prec)) return rc;
            }
            {
                WideSampleArgs a{praw, LK, io->e_part + (size_t)t * LK, TLK, io->expo_z + (size_t)t * B * LK, LK,
                                 io->post_out + (size_t)t * LK, TLK, io->y_out + (size_t)t * LK, TLK,
                                 io->zidx_out + (size_t)t * L, TL, io->z_out ? io->z_out + (size_t)t * LK : nullptr, TLK, B, L,
                                 1.0f - d.unimix_eps, d.unimix_eps / (float)d.K};
                const int64_t blocks = ((int64_t)B * L + 31) / 32;
                ProfScope prof(PC_SAMPLE, s, 12.0 * B * LK, (double)B * L * (4.0 * 128.0 + 128.0 + 1.0));
                k_observe_sample_wide<<<dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
        }
        return DRM_OK;
    }
    if (B <= SKINNY_MAX_ROWS) {
        {   // gh_raw for step 0 from h0
            ProblemSpec p{io->h0, D, nullptr, nullptr, w->gru_w_hh, D, nullptr, nullptr, 0, ghraw, 3 * D, 3 * D, D};
            if (int rc = launch_skinny(&p, 1, B, s)) return rc;
        }
        for (int t = 0; t <= T; ++t) {
            ObsGatesArgs a{};
            a.one_m_eps = 1.0f - d.unimix_eps; a.eps_over_k = d.unimix_eps / (float)d.K;
            a.D = D; a.L = L; a.do_step = t < T ? 1 : 0;
            if (t > 0) {
                a.post = io->post_out + (size_t)(t - 1) * LK; a.ldpost = TLK;
                a.expo_z = io->expo_z + (size_t)(t - 1) * B * LK; a.ldez = LK;
                a.y_out = io->y_out + (size_t)(t - 1) * LK; a.ldy = TLK;
                a.zidx_out = io->zidx_out + (size_t)(t - 1) * L; a.ldzo = TL;
                a.z_dense = io->z_out ? io->z_out + (size_t)(t - 1) * LK : nullptr; a.ldzd = TLK;
            } else {
                a.z_prev0 = io->zidx0; a.ldz0 = L;
            }
            if (t < T) {
                a.c_prev = io->c_prev + t; a.ldc = T;
                a.expo_reset = io->expo_reset + (size_t)t * B * LK; a.lder = LK;
                a.a_prev = io->a_prev + t; a.lda = T;
                a.z_in_out = io->zidx_in_out + (size_t)t * L; a.ldzi = TL;
                a.gz = w->pack + pl.gz; a.ga = w->pack + pl.ga; a.b_ih = w->pack + pl.gb; a.b_hh = w->gru_b_hh;
                a.gh_raw = ghraw; a.ldgh = 3 * D;
                a.h_prev = t == 0 ? io->h0 : io->h_out + (size_t)(t - 1) * D; a.ldh = t == 0 ? D : TD;
                a.h_out = io->h_out + (size_t)t * D; a.ldho = TD;
                a.h_in_out = io->h_in_out + (size_t)t * D; a.ldhi = TD;
                a.gates_out = io->gates_out + (size_t)t * 4 * D; a.ldgo = (int64_t)T * 4 * D;
            }
            {
                ProfScope prof(PC_GRU, s, (double)B * (40.0 * LK + 3.0 * D * (L + 8)),
                               4.0 * (double)B * (3.0 * LK + 3.0 * D * (L + 4)));
                k_observe_gates<<<dim3(t < T ? (D + 31) / 32 : 1, B), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
            if (t == T) break;
            float* h_t = io->h_out + (size_t)t * D;
            ProblemSpec p[2] = {
                {h_t, TD, nullptr, nullptr, w->fc_w_h, w->ld_fc, nullptr, nullptr, 0, io->post_out + (size_t)t * LK, TLK, LK, D},
                {h_t, TD, nullptr, nullptr, w->gru_w_hh, D, nullptr, nullptr, 0, ghraw, 3 * D, 3 * D, D}};
            p[0].add = io->e_part + (size_t)t * LK; p[0].ldadd = TLK;
            if (int rc = launch_skinny(p, t + 1 < T ? 2 : 1, B, s)) return rc;
        }
        XBulkArgs x{io->zidx_in_out, io->a_prev, io->c_prev, w->pack + pl.lin_zT, w->lin_b, w->pack + pl.a_tab, io->x_out, D, L};
        ProfScope prof(PC_GATHER, s, (double)B * T * D * (L + 2), 4.0 * (double)B * T * D * (L + 3));
        k_observe_x_bulk<<<dim3((unsigned)((int64_t)B * T)), dim3(256), 0, s>>>(x);
        DRM_LAUNCH_CHECK();
        return DRM_OK;
    }
    {   // gh_raw for step 0 from h0
        ProblemSpec p{io->h0, D, nullptr, nullptr, w->gru_w_hh, D, nullptr, nullptr, 0, ghraw, 3 * D, 3 * D, D};
        if (int rc = launch_store(&p, 1, B, nullptr, 0, 0, 0, s)) return rc;
    }
    for (int t = 0; t < T; ++t) {
        const float* h_prev = t == 0 ? io->h0 : io->h_out + (size_t)(t - 1) * D;
        const int64_t ldh = t == 0 ? D : TD;
        const uint8_t* z_prev = t == 0 ? io->zidx0 : io->zidx_out + (size_t)(t - 1) * L;
        const int64_t ldz = t == 0 ? L : TL;
        float* x_t = io->x_out + (size_t)t * D;
        float* hin_t = io->h_in_out + (size_t)t * D;
        uint8_t* zin_t = io->zidx_in_out + (size_t)t * L;
        {
            ResetArgs a{};
            a.h_prev = h_prev; a.ldh = ldh; a.z_prev = z_prev; a.ldz = ldz;
            a.a_prev = io->a_prev + t; a.lda = T; a.c_prev = io->c_prev + t; a.ldc = T;
            a.expo = io->expo_reset + (size_t)t * B * LK;
            a.lin_zT = w->pack + pl.lin_zT; a.lin_b = w->lin_b; a.a_tab = w->pack + pl.a_tab;
            a.D = D; a.L = L; a.classes = d.K;
            a.z_in = zin_t; a.ldzi = TL; a.h_in = hin_t; a.ldhi = TD; a.x = x_t; a.ldxo = TD;
            ProfScope prof(PC_ACTOR_HEAD, s, (double)B * D * (L + 2), 4.0 * (double)B * D * (L + 4));
            if (L <= 32) k_observe_reset_x<32><<<dim3(B, (D + 511) / 512), dim3(256), 0, s>>>(a);
            else k_observe_reset_x<64><<<dim3(B, (D + 511) / 512), dim3(256), 0, s>>>(a);
            DRM_LAUNCH_CHECK();
        }
        float* h_t = io->h_out + (size_t)t * D;
        {
            ProblemSpec p{x_t, TD, nullptr, nullptr, w->gru_w_ih, D, nullptr, nullptr, 0, h_t, TD, 3 * D, D};
            GruEpilogue G{w->gru_b_ih, ghraw, 3 * D, h_prev, ldh, h_t, TD, D, io->c_prev + t, T, w->gru_b_hh,
                          io->gates_out + (size_t)t * 4 * D, (int64_t)T * 4 * D};
            if (int rc = launch_gru(p, B, G, s)) return rc;
        }
        {
            ProblemSpec p{h_t, TD, nullptr, nullptr, w->fc_w_h, w->ld_fc, nullptr, nullptr, 0, ghraw, 0, LK, D};
            ProblemSpec gh{h_t, TD, nullptr, nullptr, w->gru_w_hh, D, nullptr, nullptr, 0, ghraw, 3 * D, 3 * D, D};
            SampleEpilogue E{};
            E.add = io->e_part + (size_t)t * LK; E.ldadd = TLK;
            E.y_out = io->y_out + (size_t)t * LK; E.ldyo = TLK;
            E.expo = io->expo_z + (size_t)t * B * LK; E.lde = LK;
            E.idx_used = io->zidx_out + (size_t)t * L; E.ldu = TL;
            E.z_dense = io->z_out ? io->z_out + (size_t)t * LK : nullptr; E.ldzd = TLK;
            E.logits_out = io->post_out + (size_t)t * LK; E.ldlo = TLK;
            E.log1m_eps = l1m; E.log_eps_over_k = lek;
            if (int rc = launch_sample(p, B, L, E, s, t + 1 < T ? &gh : nullptr)) return rc;
        }
    }
    return DRM_OK;
}

struct ObsGraph { std::vector<uint8_t> key; hipGraphExec_t exec; };
static std::mutex g_obs_mu;
static std::vector<ObsGraph> g_obs_graphs;

}  // namespace drm

using namespace drm;

extern "C" size_t drm_pack_observe_bytes(const drm_dims* dims) {
    if (!dims || check_observe_dims(*dims) != DRM_OK) return 0;
    return obs_pack_layout(*dims).total * sizeof(float);
}

extern "C" int drm_pack_observe(const drm_observe_weights* w, float* pack_out, size_t pack_bytes, drm_stream_t stream) {
    return drm_pack_observe_prec(w, pack_out, pack_bytes, 0, stream);
}

extern "C" int drm_pack_observe_prec(const drm_observe_weights* w, float* pack_out, size_t pack_bytes, int32_t precision,
                                     drm_stream_t stream) {
    DRM_REQUIRE(w && pack_out, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(precision == 0 || precision == 1, DRM_EINVAL, "precision=%d", precision);
    if (int rc = check_observe_dims(w->dims)) return rc;
    const drm_dims& d = w->dims;
    const ObsPack pl = obs_pack_layout(d);
    DRM_REQUIRE(pack_bytes >= pl.total * sizeof(float) && aligned16(pack_out), DRM_EINVAL, "pack buffer too small or misaligned");
    DRM_REQUIRE(w->lin_w && w->a_emb, DRM_EINVAL, "null weight");
    hipStream_t s = (hipStream_t)stream;
    const int LK = d.L * d.K;
    {
        ProfScope prof(PC_PACK, s, 2.0 * d.A * d.D * d.E, 8.0 * (double)LK * d.D);
        k_transpose_cols_o<<<dim3((LK + 31) / 32, (d.D + 31) / 32), dim3(256), 0, s>>>(w->lin_w, LK + d.E, 0, LK, d.D,
                                                                                      pack_out + pl.lin_zT);
        DRM_LAUNCH_CHECK();
        k_action_table_o<<<dim3((d.D + 255) / 256, d.A), dim3(256), 0, s>>>(w->lin_w, LK + d.E, LK, d.E, w->a_emb, d.A, d.D,
                                                                             pack_out + pl.a_tab);
        DRM_LAUNCH_CHECK();
    }
    // gz = lin_zT W_ih^T (one row of GRU input pre-activations per (latent, class)), ga = a_tab W_ih^T
    DRM_REQUIRE(w->gru_w_ih, DRM_EINVAL, "null weight");
    ProblemSpec gz{pack_out + pl.lin_zT, d.D, nullptr, nullptr, w->gru_w_ih, d.D, nullptr, nullptr, 0,
                   pack_out + pl.gz, 3 * d.D, 3 * d.D, d.D};
    if (int rc = launch_big(&gz, 1, LK, nullptr, 0, 0, 0, s, d.D > 512 ? precision : 0)) return rc;
    if (d.D > 512) return DRM_OK;      // the wide path forms gi = W_ih x per step; it reads gz only (backward pass)
    ProblemSpec ga{pack_out + pl.a_tab, d.D, nullptr, nullptr, w->gru_w_ih, d.D, nullptr, nullptr, 0,
                   pack_out + pl.ga, 3 * d.D, 3 * d.D, d.D};
    if (d.A <= SKINNY_MAX_ROWS) { if (int rc = launch_skinny(&ga, 1, d.A, s)) return rc; }
    else if (int rc = launch_store(&ga, 1, d.A, nullptr, 0, 0, 0, s)) return rc;
    // gb = W_ih lin_b + b_ih: the constant part of gi
    DRM_REQUIRE(w->lin_b && w->gru_b_ih && aligned16(w->lin_b), DRM_EINVAL, "null or misaligned lin_b / b_ih");
    ProblemSpec gb{w->lin_b, d.D, nullptr, nullptr, w->gru_w_ih, d.D, w->gru_b_ih, nullptr, 0, pack_out + pl.gb, 3 * d.D, 3 * d.D, d.D};
    return launch_skinny(&gb, 1, 1, s);
}

extern "C" size_t drm_observe_workspace_bytes(const drm_dims* dims, int32_t B, int32_t T) {
    if (!dims || check_observe_dims(*dims) != DRM_OK || B < 1 || T < 1) return 0;
    if (dims->D > 512) {
        // activations of one step + room for bf16 images of W_ih, W_hh, W_fc_h (precision 1)
        const size_t D = dims->D, LK = (size_t)dims->L * dims->K;
        const size_t act = ((size_t)B * (6 * D + LK) + 3) & ~(size_t)3;
        return act * sizeof(float) + (6 * D * D + LK * D + 64) * sizeof(uint16_t);
    }
    return (size_t)B * 3 * dims->D * sizeof(float);
}

extern "C" int drm_observe_forward_f32(const drm_observe_weights* w, const drm_observe_io* io, drm_stream_t stream) {
    DRM_REQUIRE(w && io, DRM_EINVAL, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!io->use_graph || prof_on() || sync_check_mode() != 0) return observe_enqueue(w, io, s);
    int dev = 0;
    DRM_HIP(hipGetDevice(&dev));
    std::vector<uint8_t> key(sizeof(*w) + sizeof(*io) + sizeof(int));
    memcpy(key.data(), w, sizeof(*w));
    memcpy(key.data() + sizeof(*w), io, sizeof(*io));
    memcpy(key.data() + sizeof(*w) + sizeof(*io), &dev, sizeof(int));
    std::lock_guard<std::mutex> lk(g_obs_mu);
    for (auto& e : g_obs_graphs)
        if (e.key == key) { DRM_HIP(hipGraphLaunch(e.exec, s)); return DRM_OK; }
    static thread_local hipStream_t cap = nullptr;
    static thread_local int cap_dev = -1;
    if (cap == nullptr || cap_dev != dev) {
        DRM_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        cap_dev = dev;
    }
    DRM_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    const int rc = observe_enqueue(w, io, cap);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(cap, &graph);
    if (rc != DRM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    DRM_HIP(ce);
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    DRM_HIP(ie);
    if (g_obs_graphs.size() >= 8) { (void)hipGraphExecDestroy(g_obs_graphs.front().exec); g_obs_graphs.erase(g_obs_graphs.begin()); }
    g_obs_graphs.push_back(ObsGraph{std::move(key), exec});
    DRM_HIP(hipGraphLaunch(exec, s));
    return DRM_OK;
}

extern "C" size_t drm_observe_bwd_workspace_bytes(const drm_dims* dims, int32_t B) {
    if (!dims || check_observe_dims(*dims) != DRM_OK || B < 1) return 0;
    // many sequences: dh*u and two alternating dh_next buffers; few: dh*u, d(gh) W_hh, and dz_raw
    const size_t a = (size_t)B * 3 * dims->D, b = (size_t)B * (2 * (size_t)dims->D + (size_t)dims->L * dims->K);
    if (dims->D > 512) {
        // activations of one step + room for bf16 images of gz, W_hh^T, W_fc_h^T, W_ih^T (precision 1)
        const size_t D = dims->D, LK = (size_t)dims->L * dims->K;
        const size_t act = ((size_t)B * (3 * D + LK) + 3) & ~(size_t)3;
        return act * sizeof(float) + (6 * D * D + 4 * LK * D + 64) * sizeof(uint16_t);
    }
    return (a > b ? a : b) * sizeof(float);
}

extern "C" int drm_observe_backward_f32(const drm_dims* dims, const drm_observe_bwd_io* io, drm_stream_t stream) {
    DRM_REQUIRE(dims && io, DRM_EINVAL, "null pointer");
    if (int rc = check_observe_dims(*dims)) return rc;
    const drm_dims& d = *dims;
    const int B = io->B, T = io->T, D = d.D, L = d.L, LK = d.L * d.K;
    DRM_REQUIRE(B >= 1 && T >= 1, DRM_EINVAL, "B=%d T=%d", B, T);
    DRM_REQUIRE(io->c_prev && io->y && io->wmix && io->psoft && io->gates && io->h_in && io->g_h && io->g_z && io->g_post &&
                    io->fc_w_hT && io->gru_w_hhT && io->lin_zT && io->d_gi && io->d_gh && io->d_x && io->d_post,
                DRM_EINVAL, "null io pointer");
    const bool x_from_w_ih = D > 512 && io->gru_w_ih != nullptr && (3 * D) % 32 == 0;
    DRM_REQUIRE(io->gru_w_ihT || x_from_w_ih, DRM_EINVAL, "null io pointer (gru_w_ihT)");
    DRM_REQUIRE(io->workspace && io->workspace_bytes >= (size_t)B * 3 * D * sizeof(float) && aligned16(io->workspace),
                DRM_EINVAL, "workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    float* dhu = (float*)io->workspace;
    if (D > 512) {
        // ---- wide recurrent state: the few-sequence decomposition below with the products on the large-row kernel
        DRM_REQUIRE(io->precision == 0 || io->precision == 1, DRM_EINVAL, "precision=%d", io->precision);
        const size_t act_floats = ((size_t)B * (3 * (size_t)D + LK) + 3) & ~(size_t)3;
        DRM_REQUIRE(io->workspace_bytes >= act_floats * sizeof(float), DRM_EINVAL, "workspace too small");
        const int prec = io->precision;
        struct ImageScope { ~ImageScope() { big_image_cache_end(); } } image_scope;
        if (prec == 1 && std::getenv("DRM_NO_WEIGHT_IMAGES") == nullptr)
            big_image_cache_begin(dhu + act_floats, io->workspace_bytes - act_floats * sizeof(float), 1);
        const ObsPack pl = obs_pack_layout(d);
        const float* gz = io->lin_zT + pl.gz;
        float* dhw = dhu + (size_t)B * D;
        float* dhfc = dhw + (size_t)B * D;
        float* dzraw = dhfc + (size_t)B * D;
        const int64_t TD_ = (int64_t)T * D, T3D_ = (int64_t)T * 3 * D, TLK_ = (int64_t)T * LK;
        for (int t = T - 1; t >= 0; --t) {
            const bool last = t == T - 1;
            if (!last) {
                ProblemSpec p[2] = {
                    {io->d_gi + (size_t)(t + 1) * 3 * D, T3D_, nullptr, nullptr, gz, 3 * D, nullptr, nullptr, 0, dzraw, LK, LK, 3 * D},
                    {io->d_gh + (size_t)(t + 1) * 3 * D, T3D_, nullptr, nullptr, io->gru_w_hhT, 3 * D, nullptr, nullptr, 0, dhw, D, D,
                     3 * D}};
                if (int rc = launch_big(p, 2, B, nullptr, 0, 0, 0, s, prec)) return rc;
            }
            {
                StBwdArgs a{};
                a.dz_raw = last ? nullptr : dzraw; a.lddz = LK; a.scale = last ? nullptr : io->c_prev + (t + 1); a.lds = T;
                a.g_z = io->g_z + (size_t)t * LK; a.y = io->y + (size_t)t * LK; a.wmix = io->wmix + (size_t)t * LK;
                a.psoft = io->psoft + (size_t)t * LK; a.g_post = io->g_post + (size_t)t * LK; a.ldl = TLK_;
                a.dpost = io->d_post + (size_t)t * LK; a.lddp = TLK_; a.L = L;
                ProfScope prof(PC_SAMPLE, s, 12.0 * B * LK, 28.0 * B * LK);
                k_observe_st_bwd<<<dim3(B), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
            {
                ProblemSpec p{io->d_post + (size_t)t * LK, TLK_, nullptr, nullptr, io->fc_w_hT, LK, nullptr, nullptr, 0, dhfc, D, D, LK};
                if (int rc = launch_big(&p, 1, B, nullptr, 0, 0, 0, s, prec)) return rc;
            }
            {
                WideGruBwdArgs a{dhfc, io->g_h + (size_t)t * D, TD_, last ? nullptr : dhu, dhw, last ? nullptr : io->c_prev + (t + 1), T,
                                 io->gates + (size_t)t * 4 * D, (int64_t)T * 4 * D, io->h_in + (size_t)t * D, TD_,
                                 io->d_gi + (size_t)t * 3 * D, io->d_gh + (size_t)t * 3 * D, T3D_, dhu, B, D};
                ProfScope prof(PC_ELEMENTWISE, s, 24.0 * B * D, 4.0 * 16.0 * B * D);
                k_observe_gru_bwd_wide<<<dim3((unsigned)(((int64_t)B * D + 255) / 256)), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
        }
        if (x_from_w_ih)       // d(x) = d(gi) W_ih over all (b, t) rows, W_ih read as it lies
            return drm_gemm_split_f32(0, prec == 1, B * T, D, 3 * D, io->d_gi, 3 * D, io->gru_w_ih, D, io->d_x, D, nullptr, 0, stream);
        ProblemSpec px{io->d_gi, 3 * D, nullptr, nullptr, io->gru_w_ihT, 3 * D, nullptr, nullptr, 0, io->d_x, D, D, 3 * D};
        return launch_big(&px, 1, B * T, nullptr, 0, 0, 0, s, prec);
    }
    if (B <= SKINNY_MAX_ROWS && 3 * D <= 1536 && LK <= 1536) {
        // ---- few sequences: three launches per step on ~256 workgroups each.  z enters the GRU only
        // through gi = W_ih x and x is linear in the one-hot z, so d(z_t) = d(gi_{t+1}) gz^T with the
        // forward pass's packed table gz (it follows lin_zT in the pack buffer); d(x) itself is needed
        // only for the weight gradients and is formed for all steps at once after the loop.
        DRM_REQUIRE(io->workspace_bytes >= (size_t)B * (2 * (size_t)D + LK) * sizeof(float), DRM_EINVAL, "workspace too small");
        const ObsPack pl = obs_pack_layout(d);
        const float* gz = io->lin_zT + pl.gz;
        float* dhw = dhu + (size_t)B * D;
        float* dzraw = dhw + (size_t)B * D;
        const int64_t TD_ = (int64_t)T * D, T3D_ = (int64_t)T * 3 * D, TLK_ = (int64_t)T * LK;
        for (int t = T - 1; t >= 0; --t) {
            const bool last = t == T - 1;
            if (!last) {
                ProblemSpec p[2] = {
                    {io->d_gi + (size_t)(t + 1) * 3 * D, T3D_, nullptr, nullptr, gz, 3 * D, nullptr, nullptr, 0, dzraw, LK, LK, 3 * D},
                    {io->d_gh + (size_t)(t + 1) * 3 * D, T3D_, nullptr, nullptr, io->gru_w_hhT, 3 * D, nullptr, nullptr, 0, dhw, D, D,
                     3 * D}};
                if (int rc = launch_skinny_deep(p, 2, B, s)) return rc;
            }
            {
                StBwdArgs a{};
                a.dz_raw = last ? nullptr : dzraw; a.lddz = LK; a.scale = last ? nullptr : io->c_prev + (t + 1); a.lds = T;
                a.g_z = io->g_z + (size_t)t * LK; a.y = io->y + (size_t)t * LK; a.wmix = io->wmix + (size_t)t * LK;
                a.psoft = io->psoft + (size_t)t * LK; a.g_post = io->g_post + (size_t)t * LK; a.ldl = TLK_;
                a.dpost = io->d_post + (size_t)t * LK; a.lddp = TLK_; a.L = L;
                ProfScope prof(PC_SAMPLE, s, 12.0 * B * LK, 28.0 * B * LK);
                k_observe_st_bwd<<<dim3(B), dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
            {
                ProblemSpec p{io->d_post + (size_t)t * LK, TLK_, nullptr, nullptr, io->fc_w_hT, LK, nullptr, nullptr, 0, dhu, D, D, LK};
                SkinnyGruBwd G{};
                G.g_h = io->g_h + (size_t)t * D; G.ldgh = TD_;
                G.dhu_next = last ? nullptr : dhu; G.dhw_next = dhw; G.next_scale = last ? nullptr : io->c_prev + (t + 1); G.ldns = T;
                G.gates = io->gates + (size_t)t * 4 * D; G.ldg = (int64_t)T * 4 * D;
                G.h_in = io->h_in + (size_t)t * D; G.ldhi = TD_;
                G.dgi = io->d_gi + (size_t)t * 3 * D; G.dgh = io->d_gh + (size_t)t * 3 * D; G.lddg = T3D_;
                G.dhu = dhu; G.D = D;
                if (int rc = launch_skinny_deep(&p, 1, B, s, &G)) return rc;
            }
        }
        // d(x) = d(gi) W_ih for all (b, t) rows at once
        ProblemSpec px{io->d_gi, 3 * D, nullptr, nullptr, io->gru_w_ihT, 3 * D, nullptr, nullptr, 0, io->d_x, D, D, 3 * D};
        return launch_big(&px, 1, B * T, nullptr, 0, 0, 0, s);
    }
    float* dhn[2] = {dhu + (size_t)B * D, dhu + (size_t)2 * B * D};
    const int64_t TD = (int64_t)T * D, T3D = (int64_t)T * 3 * D, T4D = (int64_t)T * 4 * D, TLK = (int64_t)T * LK;
    DRM_HIP(hipMemsetAsync(dhn[0], 0, (size_t)B * D * sizeof(float), s));
    auto stbwd = [&](int t, const float* dx_next, bool skip) -> int {
        // d(post)[t] from dz = (d(x)[t+1] @ lin_z) * c[t+1] + g_z[t]
        BwdEpilogue E{};
        E.rowscale = skip ? nullptr : io->c_prev + (t + 1); E.ldrs = T;
        E.g_z = io->g_z + (size_t)t * LK; E.y = io->y + (size_t)t * LK; E.wmix = io->wmix + (size_t)t * LK;
        E.psoft = io->psoft + (size_t)t * LK; E.g_post = io->g_post + (size_t)t * LK; E.ldl = TLK;
        E.dpost = io->d_post + (size_t)t * LK; E.lddp = TLK;
        E.skip_acc = skip ? 1 : 0;
        ProblemSpec p{skip ? io->h_in : dx_next, TD, nullptr, nullptr, io->lin_zT, D, nullptr, nullptr, 0, dhu, 0, LK, D};
        return launch_stbwd(p, B, L, E, s);
    };
    if (int rc = stbwd(T - 1, nullptr, true)) return rc;
    for (int t = T - 1; t >= 0; --t) {
        const int cur = (T - 1 - t) & 1;
        {   // dh = d(post)[t] @ fc_w_h + g_h[t] + dh_next; GRU cell backward
            ProblemSpec p{io->d_post + (size_t)t * LK, TLK, nullptr, nullptr, io->fc_w_hT, LK, nullptr, nullptr, 0, dhu, D, D, LK};
            BwdEpilogue E{};
            E.g_h = io->g_h + (size_t)t * D; E.ldgh = TD; E.dh_next = dhn[cur]; E.lddn = D;
            E.gates = io->gates + (size_t)t * 4 * D; E.ldg = T4D; E.h_in = io->h_in + (size_t)t * D; E.ldhi = TD;
            E.dgi = io->d_gi + (size_t)t * 3 * D; E.dgh = io->d_gh + (size_t)t * 3 * D; E.lddg = T3D;
            E.dhu = dhu; E.lddhu = D; E.D = D;
            if (int rc = launch_grubwd(p, B, E, s)) return rc;
        }
        {   // d(x)[t] = d(gi) @ W_ih;  dh_next = (dh*u + d(gh) @ W_hh) * c[t]
            ProblemSpec p[2] = {
                {io->d_gi + (size_t)t * 3 * D, T3D, nullptr, nullptr, io->gru_w_ihT, 3 * D, nullptr, nullptr, 0,
                 io->d_x + (size_t)t * D, TD, D, 3 * D},
                {io->d_gh + (size_t)t * 3 * D, T3D, nullptr, nullptr, io->gru_w_hhT, 3 * D, nullptr, nullptr, 0, dhn[cur ^ 1], D, D,
                 3 * D}};
            p[1].add = dhu; p[1].ldadd = D; p[1].rowscale = io->c_prev + t; p[1].ldrs = T;
            if (int rc = launch_store(p, 2, B, nullptr, 0, 0, 0, s)) return rc;
        }
        if (t > 0) {
            if (int rc = stbwd(t - 1, io->d_x + (size_t)t * D, false)) return rc;
        }
    }
    return DRM_OK;
}
