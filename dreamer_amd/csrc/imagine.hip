// Imagination block (actor -> action -> prior RSSM step, H times; heads; critic value;
// lambda-return) as a chain of fused-linear launches.  Every layer boundary is an all-to-all over
// the feature dimension (LayerNorm and the next matmul need whole rows), so each is a kernel
// boundary; inside a kernel the output columns are spread over the CUs.
#include "launch.hpp"
#include "rng.hpp"
#include "linear_big.hpp"

#include <cstring>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace drm {

// ------------------------------------------------------------------------------------------------
// weight packing: one-hot gather tables.
// ------------------------------------------------------------------------------------------------
// out[c][n] = W[n][col0 + c]   (c < ncols, n < nrows)
__global__ __launch_bounds__(256) void k_transpose_cols(const float* __restrict__ W, int64_t ldw, int col0, int ncols,
                                                         int nrows, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
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

// a_tab[a][j] = sum_e lin_w[j][LK + e] * a_emb[a][e]
__global__ __launch_bounds__(256) void k_action_table(const float* __restrict__ lin_w, int64_t ldw, int col0, int E,
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

// ------------------------------------------------------------------------------------------------
// block-wide sum over 256 threads (4 waves)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* s4) {
    v = group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ------------------------------------------------------------------------------------------------
// actor head + action draw + GRU input row, one workgroup per trajectory:
//   act = SiLU(LN(pre));  logits = head(act);  logp = normalise(log_unimix(logits));
//   a = argmax(softmax(logp) / E);  x = lin_b + sum_l lin_zT[l*K + z_l] + a_tab[a]
// (lib/actor.py:94-106, Categorical.sample, lib/world_model.py:64-70 with one-hot z)
// ------------------------------------------------------------------------------------------------
struct ActorHeadArgs {
    const float* pre; int64_t ldp;
    const float* ln_g; const float* ln_b;
    const float* head_w; const float* head_b;
    int M, A;
    const float* expo; int64_t lde;          // (B, A)
    const int64_t* forced; int64_t ldf;      // (B) strided or nullptr
    float log1m_eps, log_eps_over_a;
    const uint8_t* zidx; int64_t ldz; int latents, classes;
    const float* lin_zT; const float* lin_b; const float* a_tab; int D;
    const float* xz; int64_t ldxz;           // when given: lin_b + the z table rows, already summed (B, D)
    float* x; int64_t ldxo;
    int64_t* a_used; int64_t ldau;
    int64_t* a_sampled; int64_t ldas;        // nullable
    float* logp_out; int64_t ldlp;           // nullable
};

// M <= 512 and D <= 512: every thread owns at most two features / two output columns, and every
// global load that does not depend on the drawn action is issued at the top of the kernel (the
// one-hot table rows of x only need z, so they are in flight while the policy is evaluated; the
// action-table rows of all actions are prefetched and selected afterwards).
template <int AMAX, int LMAX>
__global__ __launch_bounds__(256) void k_actor_head_x(const ActorHeadArgs P) {
    __shared__ float s4[4];
    __shared__ uint8_t s_z[LMAX];
    __shared__ float s_dot[4][AMAX];
    __shared__ int s_a;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    STAMP_DECL
    STAMP();
    const float* pre = P.pre + (int64_t)b * P.ldp;
    const int i0 = tid, i1 = tid + 256;
    const bool on0 = i0 < P.M, on1 = i1 < P.M;
    const bool jn0 = i0 < P.D, jn1 = i1 < P.D;
    // ---- loads, level 1
    if (tid < P.latents && P.xz == nullptr) s_z[tid] = P.zidx[(int64_t)b * P.ldz + tid];
    const float x0 = on0 ? pre[i0] : 0.f, x1 = on1 ? pre[i1] : 0.f;
    const float g0 = on0 ? P.ln_g[i0] : 0.f, g1 = on1 ? P.ln_g[i1] : 0.f;
    const float e0 = on0 ? P.ln_b[i0] : 0.f, e1 = on1 ? P.ln_b[i1] : 0.f;
    float hw0[AMAX], hw1[AMAX], at0[AMAX], at1[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        const bool aa = a < P.A;
        hw0[a] = (aa && on0) ? P.head_w[(int64_t)a * P.M + i0] : 0.f;
        hw1[a] = (aa && on1) ? P.head_w[(int64_t)a * P.M + i1] : 0.f;
        at0[a] = (aa && jn0) ? P.a_tab[(int64_t)a * P.D + i0] : 0.f;
        at1[a] = (aa && jn1) ? P.a_tab[(int64_t)a * P.D + i1] : 0.f;
    }
    const float ex = lane < P.A ? P.expo[(int64_t)b * P.lde + lane] : 1.f;
    const float hb = lane < P.A ? P.head_b[lane] : 0.f;
    const bool have_xz = P.xz != nullptr;
    float xz0, xz1;
    if (have_xz) {
        xz0 = jn0 ? P.xz[(int64_t)b * P.ldxz + i0] : 0.f;
        xz1 = jn1 ? P.xz[(int64_t)b * P.ldxz + i1] : 0.f;
    } else {
        xz0 = jn0 ? P.lin_b[i0] : 0.f;
        xz1 = jn1 ? P.lin_b[i1] : 0.f;
    }
    // ---- loads, level 2: table rows selected by z, all in flight together
    __syncthreads();
    float u0[LMAX], u1[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        const bool ok = l < P.latents && !have_xz;
        const int64_t rowoff = ok ? (int64_t)(l * P.classes + s_z[l]) * P.D : 0;
        u0[l] = (ok && jn0) ? P.lin_zT[rowoff + i0] : 0.f;
        u1[l] = (ok && jn1) ? P.lin_zT[rowoff + i1] : 0.f;
    }
#ifdef DRM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
#endif
    // ---- LayerNorm + SiLU
    const float mean = block_sum_256(x0 + x1, s4) / (float)P.M;
    const float d0 = on0 ? x0 - mean : 0.f, d1 = on1 ? x1 - mean : 0.f;
    const float rstd = 1.0f / sqrtf(block_sum_256(d0 * d0 + d1 * d1, s4) / (float)P.M + LN_EPS);
    const float a0 = on0 ? silu_f(d0 * rstd * g0 + e0) : 0.f, a1 = on1 ? silu_f(d1 * rstd * g1 + e1) : 0.f;
    // ---- head: per-thread partial dots, wave reduce, cross-wave through LDS
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        const float d = group_sum<64>(fmaf(a0, hw0[a], a1 * hw1[a]));
        if (lane == 0) s_dot[wave][a] = d;
    }
    __syncthreads();
    if (wave == 0) {
        const bool on = lane < P.A;
        const float NEG = -3.0e38f;
        const int la = on ? lane : 0;
        const float lg = on ? ((s_dot[0][la % AMAX] + s_dot[1][la % AMAX]) + (s_dot[2][la % AMAX] + s_dot[3][la % AMAX])) + hb : NEG;
        const float mx = group_max<64>(lg);
        const float lse = logf(group_sum<64>(on ? expf(lg - mx) : 0.f));
        const float mixed = on ? logaddexp_f((lg - mx - lse) + P.log1m_eps, P.log_eps_over_a) : NEG;
        // Categorical(logits=mixed) normalises once more
        const float mx2 = group_max<64>(mixed);
        const float lse2 = mx2 + logf(group_sum<64>(on ? expf(mixed - mx2) : 0.f));
        const float logp = mixed - lse2;
        // probs = softmax(logp); multinomial draws argmax(probs / E)
        const float mx3 = group_max<64>(on ? logp : NEG);
        const float pe = on ? expf(logp - mx3) : 0.f;
        const float prob = pe / group_sum<64>(pe);
        float score = on ? prob / ex : -1.f;
        int ai = lane;
        group_argmax<64>(score, ai);
        if (on && P.logp_out) P.logp_out[(int64_t)b * P.ldlp + lane] = logp;
        if (lane == 0) {
            int used = ai;
            if (P.forced) used = (int)P.forced[(int64_t)b * P.ldf];
            s_a = used;
            P.a_used[(int64_t)b * P.ldau] = used;
            if (P.a_sampled) P.a_sampled[(int64_t)b * P.ldas] = ai;
        }
    }
    __syncthreads();
    STAMP();
    STAMP_FLUSH(2000ull);
    const int act = s_a;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { xz0 += u0[l]; xz1 += u1[l]; }
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        if (a == act) { t0 = at0[a]; t1 = at1[a]; }
    }
    if (jn0) P.x[(int64_t)b * P.ldxo + i0] = xz0 + t0;
    if (jn1) P.x[(int64_t)b * P.ldxo + i1] = xz1 + t1;
}

// The same for any width (M, D up to 4096): features and columns are walked in strides of 256.
__global__ __launch_bounds__(256) void k_actor_head_wide(const ActorHeadArgs P) {
    __shared__ float s4[4];
    __shared__ uint8_t s_z[MAX_LATENTS];
    __shared__ float s_dot[4][32];
    __shared__ int s_a;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const float* pre = P.pre + (int64_t)b * P.ldp;
    if (tid < P.latents) s_z[tid] = P.zidx[(int64_t)b * P.ldz + tid];
    float s = 0.f;
    for (int i = tid; i < P.M; i += 256) s += pre[i];
    const float mean = block_sum_256(s, s4) / (float)P.M;
    float qv = 0.f;
    for (int i = tid; i < P.M; i += 256) { const float d = pre[i] - mean; qv += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum_256(qv, s4) / (float)P.M + LN_EPS);
    float dot[32];
#pragma unroll
    for (int a = 0; a < 32; ++a) dot[a] = 0.f;
    for (int i = tid; i < P.M; i += 256) {
        const float act = silu_f((pre[i] - mean) * rstd * P.ln_g[i] + P.ln_b[i]);
#pragma unroll
        for (int a = 0; a < 32; ++a)
            if (a < P.A) dot[a] = fmaf(act, P.head_w[(int64_t)a * P.M + i], dot[a]);
    }
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const float d = group_sum<64>(dot[a]);
        if (lane == 0) s_dot[wave][a] = d;
    }
    __syncthreads();
    if (wave == 0) {
        const bool on = lane < P.A;
        const float NEG = -3.0e38f;
        const int la = on ? lane : 0;
        const float ex = on ? P.expo[(int64_t)b * P.lde + la] : 1.f;
        const float lg = on ? ((s_dot[0][la] + s_dot[1][la]) + (s_dot[2][la] + s_dot[3][la])) + P.head_b[la] : NEG;
        const float mx = group_max<64>(lg);
        const float lse = logf(group_sum<64>(on ? expf(lg - mx) : 0.f));
        const float mixed = on ? logaddexp_f((lg - mx - lse) + P.log1m_eps, P.log_eps_over_a) : NEG;
        const float mx2 = group_max<64>(mixed);
        const float lse2 = mx2 + logf(group_sum<64>(on ? expf(mixed - mx2) : 0.f));
        const float logp = mixed - lse2;
        const float mx3 = group_max<64>(on ? logp : NEG);
        const float pe = on ? expf(logp - mx3) : 0.f;
        const float prob = pe / group_sum<64>(pe);
        float score = on ? prob / ex : -1.f;
        int ai = lane;
        group_argmax<64>(score, ai);
        if (on && P.logp_out) P.logp_out[(int64_t)b * P.ldlp + lane] = logp;
        if (lane == 0) {
            int used = ai;
            if (P.forced) used = (int)P.forced[(int64_t)b * P.ldf];
            s_a = used;
            P.a_used[(int64_t)b * P.ldau] = used;
            if (P.a_sampled) P.a_sampled[(int64_t)b * P.ldas] = ai;
        }
    }
    __syncthreads();
    const int act = s_a;
    for (int jcol = tid; jcol < P.D; jcol += 256) {
        float acc = P.lin_b[jcol];
        for (int l0 = 0; l0 < P.latents; l0 += 8) {
            float u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                u[q] = (l0 + q < P.latents) ? P.lin_zT[(int64_t)((l0 + q) * P.classes + s_z[l0 + q]) * P.D + jcol] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += u[q];
        }
        P.x[(int64_t)b * P.ldxo + jcol] = acc + P.a_tab[(int64_t)act * P.D + jcol];
    }
}

// ------------------------------------------------------------------------------------------------
// few-row path: actor head + action draw + GRU cell, one workgroup per trajectory.
//   act = SiLU(LN(pre));  logits = head(act);  logp = normalise(log_unimix(logits));  a = argmax(softmax(logp) / E)
//   gi = gi_z + gi_a[a] + b_ih;  h' = GRUCell gates(gi, gh, h)
// The GRU input row is x = lin_b + (z table rows) + a_tab[a], so W_ih x splits into a part that does
// not need the action (gi_z, formed by an earlier launch while the policy MLP runs) and a row of the
// precomputed table gi_a = W_ih a_tab^T chosen by the action: the recurrent step needs no matrix
// product after the draw.  (lib/actor.py:94-106, Categorical.sample, lib/world_model.py:64-71)
// ------------------------------------------------------------------------------------------------
struct ActorGatesArgs {
    const float* pre; int64_t ldp;
    const float* ln_g; const float* ln_b;
    const float* head_w; const float* head_b;
    int M, A;
    const float* expo; int64_t lde;          // (B, A), or nullptr: drawn in the kernel from `rng`
    RngArgs rng;
    const int64_t* forced; int64_t ldf;      // (B) strided or nullptr
    float one_m_eps, eps_over_a;             // 1 - unimix_eps, unimix_eps / A
    const float* gi_z; int64_t ldgi;         // (B, 3D)
    const float* gi_a;                       // (A, 3D)
    const float* b_ih;                       // (3D)
    const float* gh; int64_t ldgh;           // (B, 3D) = W_hh h + b_hh
    const float* h_prev; int64_t ldh;
    float* h_new; int64_t ldhn; int D;
    int64_t* a_used; int64_t ldau;
    int64_t* a_sampled; int64_t ldas;        // nullable
    float* logp_out; int64_t ldlp;           // nullable
};

__device__ __forceinline__ float2 ld2(const float* p, bool on) {
    return on ? *reinterpret_cast<const float2*>(p) : make_float2(0.f, 0.f);
}

// M <= 512 (even), any even D: thread t owns features 2t, 2t+1 and hidden units 512 y + 2t, + 1 of slice y = blockIdx.y
// of the recurrent state (every slice's workgroup evaluates the policy; slice 0 writes the action and its log-probabilities).
template <int AMAX>
__global__ __launch_bounds__(256) void k_actor_gates(const ActorGatesArgs P) {
    __shared__ float s_st[4][2];
    __shared__ float s_dot[4][AMAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    STAMP_DECL
    STAMP();
    const int i0 = 2 * tid, j0 = blockIdx.y * 512 + 2 * tid;
    const bool on = i0 < P.M, jn = j0 < P.D, first = blockIdx.y == 0;
    const int D = P.D;
    // ---- loads that do not depend on the action
    const float* pre = P.pre + (int64_t)b * P.ldp;
    const float shift = pre[0];
    const float2 xv = ld2(pre + i0, on);
    const float2 gv = ld2(P.ln_g + i0, on), ev = ld2(P.ln_b + i0, on);
    float2 hw[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) hw[a] = ld2(P.head_w + (int64_t)a * P.M + i0, on && a < P.A);
    const int la = lane % AMAX;
    const bool aon = la < P.A;
    float ex = 1.f;
    if (P.expo != nullptr) {                 // (uniform)
        if (aon) ex = P.expo[(int64_t)b * P.lde + la];
    } else {
        // action `la` of row b: component la % 4 of group la / 4 of the row's (A + 3) / 4 groups
        const uint64_t row = (uint64_t)P.rng.t * (uint64_t)P.rng.rows_total + (uint64_t)(P.rng.row0 + b);
        const float4 e4 = rng_exp4(P.rng, 1u, row * (uint64_t)((P.A + 3) >> 2) + (uint64_t)(la >> 2));
        const int c = la & 3;
        ex = aon ? (c == 0 ? e4.x : c == 1 ? e4.y : c == 2 ? e4.z : e4.w) : 1.f;
    }
    const float hb = aon ? P.head_b[la] : 0.f;
    const float* giz = P.gi_z + (int64_t)b * P.ldgi;
    const float* ghp = P.gh + (int64_t)b * P.ldgh;
    const float2 gz_r = ld2(giz + j0, jn), gz_u = ld2(giz + D + j0, jn), gz_n = ld2(giz + 2 * D + j0, jn);
    const float2 gh_r = ld2(ghp + j0, jn), gh_u = ld2(ghp + D + j0, jn), gh_n = ld2(ghp + 2 * D + j0, jn);
    const float2 bi_r = ld2(P.b_ih + j0, jn), bi_u = ld2(P.b_ih + D + j0, jn), bi_n = ld2(P.b_ih + 2 * D + j0, jn);
    const float2 hp = ld2(P.h_prev + (int64_t)b * P.ldh + j0, jn);
    const int64_t forced = P.forced ? P.forced[(int64_t)b * P.ldf] : -1;
#ifdef DRM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
#endif
    // ---- LayerNorm (one pass, shifted by the row's first element) + SiLU
    float st[2];
    {
        const float d0 = on ? xv.x - shift : 0.f, d1 = on ? xv.y - shift : 0.f;
        st[0] = d0 + d1;
        st[1] = fmaf(d0, d0, d1 * d1);
    }
    all_sum64<2>(st);
    if (lane == 0) { s_st[wave][0] = st[0]; s_st[wave][1] = st[1]; }
    __syncthreads();
    const float inv_m = 1.0f / (float)P.M;
    const float md = ((s_st[0][0] + s_st[1][0]) + (s_st[2][0] + s_st[3][0])) * inv_m;
    const float var = fmaxf(((s_st[0][1] + s_st[1][1]) + (s_st[2][1] + s_st[3][1])) * inv_m - md * md, 0.f);
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    const float mean = shift + md;
    const float a0 = on ? silu_f((xv.x - mean) * rstd * gv.x + ev.x) : 0.f;
    const float a1 = on ? silu_f((xv.y - mean) * rstd * gv.y + ev.y) : 0.f;
    // ---- head: per-thread partial dots, wave butterfly, cross-wave through LDS
    float dot[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) dot[a] = fmaf(a0, hw[a].x, a1 * hw[a].y);
    const float dsum = halving_sum64<AMAX>(dot, lane);
    if (lane < AMAX) s_dot[wave][halving_index<AMAX>(lane)] = dsum;
    __syncthreads();
    // every group of AMAX lanes of every wave evaluates the policy (no broadcast of the action needed)
    const float NEG = -3.0e38f;
    const float lg = aon ? ((s_dot[0][la] + s_dot[1][la]) + (s_dot[2][la] + s_dot[3][la])) + hb : NEG;
    // P_a = (1-eps) softmax(logits)_a + eps/A is the policy (exp of the normalised log_unimix);
    // torch.multinomial draws argmax_a P_a / E_a.  The draw needs no logarithm; the log-probabilities
    // are formed after the action's table rows have been requested.
    const float mx = group_max<AMAX>(lg);
    const float pa = aon ? expf(lg - mx) : 0.f;
    const float zsum = group_sum<AMAX>(pa);                       // every lane takes part in the exchange
    const float Pa = aon ? fmaf(pa, P.one_m_eps / zsum, P.eps_over_a) : 0.f;
    float score = aon ? Pa / ex : -1.f;
    int ai = la;
    group_argmax<AMAX>(score, ai);
    const int used = forced >= 0 ? (int)forced : ai;
    STAMP();
    // ---- GRU cell with the action's rows of gi_a
    const float* ga = P.gi_a + (int64_t)used * 3 * D;
    const float2 ga_r = ld2(ga + j0, jn), ga_u = ld2(ga + D + j0, jn), ga_n = ld2(ga + 2 * D + j0, jn);
    if (tid == 0 && first) {
        P.a_used[(int64_t)b * P.ldau] = used;
        if (P.a_sampled) P.a_sampled[(int64_t)b * P.ldas] = ai;
    }
    const float ptot = group_sum<AMAX>(Pa);
    if (P.logp_out && tid < P.A && first) P.logp_out[(int64_t)b * P.ldlp + tid] = logf(Pa) - logf(ptot);
    if (!jn) return;
    const float r0 = sigmoid_f(((gz_r.x + ga_r.x) + bi_r.x) + gh_r.x), r1 = sigmoid_f(((gz_r.y + ga_r.y) + bi_r.y) + gh_r.y);
    const float u0 = sigmoid_f(((gz_u.x + ga_u.x) + bi_u.x) + gh_u.x), u1 = sigmoid_f(((gz_u.y + ga_u.y) + bi_u.y) + gh_u.y);
    const float n0 = tanhf(((gz_n.x + ga_n.x) + bi_n.x) + r0 * gh_n.x), n1 = tanhf(((gz_n.y + ga_n.y) + bi_n.y) + r1 * gh_n.y);
    float2 out;
    out.x = (1.0f - u0) * n0 + u0 * hp.x;
    out.y = (1.0f - u1) * n1 + u1 * hp.y;
    *reinterpret_cast<float2*>(P.h_new + (int64_t)b * P.ldhn + j0) = out;
    STAMP();
    STAMP_FLUSH(2000ull);
}

// ------------------------------------------------------------------------------------------------
// scalar heads: out = post(head_w . SiLU(LN(pre row)) + head_b); one wave per (row, head)
// (lib/world_model.py:215-216, :259-260 and lib/trainer.py:193-194)
// ------------------------------------------------------------------------------------------------
struct ScalarHead {
    const float* pre; int64_t ldp;
    const float* ln_g; const float* ln_b;
    const float* head_w; const float* head_b;
    float* out;        // post-processed value per row
    float* raw_out;    // raw head output per row or nullptr
    int post;          // 0 none, 1 symexp, 2 sigmoid
};
struct ScalarHeadsArgs { ScalarHead h[2]; int nh; int rows; int M; };

__device__ __forceinline__ void scalar_heads_row(const ScalarHeadsArgs& P, int row, int lane) {
    for (int hi = 0; hi < P.nh; ++hi) {
        const ScalarHead& Hd = P.h[hi];
        const float* pre = Hd.pre + (int64_t)row * Hd.ldp;
        float s = 0.f;
        for (int i = lane * 4; i < P.M; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(pre + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
        const float mean = group_sum<64>(s) / (float)P.M;
        float qv = 0.f;
        for (int i = lane * 4; i < P.M; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(pre + i);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            qv += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = 1.0f / sqrtf(group_sum<64>(qv) / (float)P.M + LN_EPS);
        float dot = 0.f;
        for (int i = lane * 4; i < P.M; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(pre + i);
            const float4 g = *reinterpret_cast<const float4*>(Hd.ln_g + i);
            const float4 e = *reinterpret_cast<const float4*>(Hd.ln_b + i);
            const float4 w = *reinterpret_cast<const float4*>(Hd.head_w + i);
            dot = fmaf(silu_f((v.x - mean) * rstd * g.x + e.x), w.x, dot);
            dot = fmaf(silu_f((v.y - mean) * rstd * g.y + e.y), w.y, dot);
            dot = fmaf(silu_f((v.z - mean) * rstd * g.z + e.z), w.z, dot);
            dot = fmaf(silu_f((v.w - mean) * rstd * g.w + e.w), w.w, dot);
        }
        dot = group_sum<64>(dot) + Hd.head_b[0];
        if (lane == 0) {
            if (Hd.raw_out) Hd.raw_out[row] = dot;
            Hd.out[row] = Hd.post == 1 ? symexp_f(dot) : (Hd.post == 2 ? sigmoid_f(dot) : dot);
        }
    }
}

// critic value: symexp(sum softmax(logits) * buckets), one wave per row (lib/critic.py:94-101);
// also copies the value of each trajectory's last step to v_last (lib/trainer.py:212-215).
struct CriticValueArgs { const float* logits; int64_t ldl; int NB; const float* buckets; int H; float* V; float* v_last; };

__device__ __forceinline__ void critic_value_row(const CriticValueArgs& Cv, int row, int lane) {
    const float* lg = Cv.logits + (int64_t)row * Cv.ldl;
    const float* buckets = Cv.buckets;
    const int NB = Cv.NB, H = Cv.H;
    float* V = Cv.V; float* v_last = Cv.v_last;
    float mx = -3.0e38f;
    for (int i = lane; i < NB; i += 64) mx = fmaxf(mx, lg[i]);
    mx = group_max<64>(mx);
    float se = 0.f, sb = 0.f;
    for (int i = lane; i < NB; i += 64) {
        const float e = expf(lg[i] - mx);
        se += e;
        sb = fmaf(e, buckets[i], sb);
    }
    se = group_sum<64>(se);
    sb = group_sum<64>(sb);
    if (lane == 0) {
        const float v = symexp_f(sb / se);
        V[row] = v;
        if (v_last && (row % H) == H - 1) v_last[row / H] = v;
    }
}

// Both row-wise head finishes in one launch: blockIdx.y = 0 -> reward / continue heads, 1 -> critic value.
__global__ __launch_bounds__(256) void k_scalar_heads(const ScalarHeadsArgs P, const CriticValueArgs Cv) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= P.rows) return;
    if (blockIdx.y == 0) scalar_heads_row(P, row, lane);
    else critic_value_row(Cv, row, lane);
}

// ------------------------------------------------------------------------------------------------
// large-row path helpers
// ------------------------------------------------------------------------------------------------
// unimix + Gumbel-argmax draw over complete logits (bias included), eight lanes per latent, writing into
// the strided (B, H, ...) outputs of the rollout
struct SampleRowsArgs {
    const float* logits; int64_t ldlg;       // (B, L*32)
    const float* expo; int64_t lde;          // or nullptr: drawn in the kernel from `rng`
    RngArgs rng;
    const uint8_t* forced; int64_t ldf;
    uint8_t* idx_used; int64_t ldu;
    uint8_t* idx_sampled; int64_t ldsm;
    float* z_dense; int64_t ldzd;
    float* logits_out; int64_t ldlo;
    int B, L;
    float log1m_eps, log_eps_over_k;
    float one_m_eps, eps_over_k;             // 1 - unimix_eps, unimix_eps / K
};

__global__ __launch_bounds__(256) void k_sample_rows(const SampleRowsArgs P) {
    const int sub = threadIdx.x & 7;
    const int64_t total = (int64_t)P.B * P.L;
    for (int64_t it = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); it < total; it += (int64_t)gridDim.x * 32) {
        const int b = (int)(it / P.L), l = (int)(it - (int64_t)b * P.L);
        const float4 lg = *reinterpret_cast<const float4*>(P.logits + (int64_t)b * P.ldlg + l * 32 + sub * 4);
        const float4 ex = P.expo != nullptr
            ? *reinterpret_cast<const float4*>(P.expo + (int64_t)b * P.lde + l * 32 + sub * 4)
            : rng_exp4(P.rng, 0u, (((uint64_t)P.rng.t * (uint64_t)P.rng.rows_total + (uint64_t)(P.rng.row0 + b)) * (uint64_t)P.L + (uint64_t)l) * 8u + (uint64_t)sub);
        // argmax_k softmax(log_unimix(logits) - log E)_k = argmax_k ((1-eps) softmax(logits)_k + eps/K) / E_k
        const float v[4] = {lg.x, lg.y, lg.z, lg.w};
        const float e[4] = {ex.x, ex.y, ex.z, ex.w};
        const float mx = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float pj[4], s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pj[j] = expf(v[j] - mx); s += pj[j]; }
        const float scale = P.one_m_eps * __builtin_amdgcn_rcpf(group_sum<8>(s));
        float best = -1.f;
        int bi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q = fmaf(pj[j], scale, P.eps_over_k) * __builtin_amdgcn_rcpf(e[j]);   // reciprocal within 1 ulp
            if (q > best) { best = q; bi = sub * 4 + j; }
        }
        group_argmax<8>(best, bi);
        int used = bi;
        if (P.forced) used = P.forced[(int64_t)b * P.ldf + l];
        if (sub == 0) {
            P.idx_used[(int64_t)b * P.ldu + l] = (uint8_t)used;
            if (P.idx_sampled) P.idx_sampled[(int64_t)b * P.ldsm + l] = (uint8_t)bi;
        }
        const int k0 = sub * 4;
        if (P.z_dense)
            *reinterpret_cast<float4*>(P.z_dense + (int64_t)b * P.ldzd + l * 32 + k0) =
                make_float4(used == k0 ? 1.f : 0.f, used == k0 + 1 ? 1.f : 0.f, used == k0 + 2 ? 1.f : 0.f, used == k0 + 3 ? 1.f : 0.f);
        if (P.logits_out) *reinterpret_cast<float4*>(P.logits_out + (int64_t)b * P.ldlo + l * 32 + k0) = lg;
    }
}

// ------------------------------------------------------------------------------------------------
// few-row path: draw z for one trajectory, then apply every one-hot table that consumes it
// ------------------------------------------------------------------------------------------------
// Workgroup (slice, b): draws all L latents of trajectory b (1024 logits: four per thread, every
// workgroup of the row repeats this small amount of work rather than wait for another one), then
// forms its slice of columns of
//     out_g[b, n] = base_g[b or 0, n] + sum_l tab_g[(l*32 + z[b, l]), n]         g = 0, 1
// which are the z part of the actor's first Linear (added to the h part computed earlier) and the
// GRU input row without its action term (lib/actor.py:94-99, lib/world_model.py:64-70 with z one-hot).
struct GatherJob {
    const float* tab; int64_t ldt;
    const float* base; int64_t ldb;     // row stride of base (0: one vector for every row)
    float* out; int64_t ldo;
    int N, cw;                          // columns, columns per slice (multiple of 4)
    float* stats = nullptr;             // or: (rows, slices) x (sum, sum of squared deviations from the slice mean) of the out values
};
struct SampleGatherArgs {
    SampleRowsArgs s;                   // logits == nullptr: no draw, z = forced
    GatherJob g[2];
    int ng;
    float one_m_eps, eps_over_k;        // 1 - unimix_eps, unimix_eps / K
};

__global__ __launch_bounds__(256) void k_sample_gather(const SampleGatherArgs P) {
    __shared__ uint8_t s_z[64];
    __shared__ __attribute__((aligned(16))) float s_red[2][1024];
    const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    const SampleRowsArgs& S = P.s;
    const int sub = tid & 7;
    STAMP_DECL
    STAMP();
    for (int l = tid >> 3; l < S.L; l += 32) {
        int used, bi = 0;
        float4 lg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (S.logits != nullptr) {
            lg = *reinterpret_cast<const float4*>(S.logits + (int64_t)b * S.ldlg + l * 32 + sub * 4);
            const float4 ex = S.expo != nullptr
                ? *reinterpret_cast<const float4*>(S.expo + (int64_t)b * S.lde + l * 32 + sub * 4)
                : rng_exp4(S.rng, 0u, (((uint64_t)S.rng.t * (uint64_t)S.rng.rows_total + (uint64_t)(S.rng.row0 + b)) * (uint64_t)S.L + (uint64_t)l) * 8u + (uint64_t)sub);
            // argmax_k softmax(log_unimix(logits) - log E)_k = argmax_k ((1-eps) softmax(logits)_k + eps/K) / E_k:
            // the draw needs no logarithm (the log-domain form of lib/world_model.py:459-461 is
            // monotone in this ratio)
            const float v[4] = {lg.x, lg.y, lg.z, lg.w};
            const float e[4] = {ex.x, ex.y, ex.z, ex.w};
            const float mx = group_max<8>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            float pj[4], s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pj[j] = expf(v[j] - mx); s += pj[j]; }
            const float scale = P.one_m_eps / group_sum<8>(s);
            float best = -1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float q = fmaf(pj[j], scale, P.eps_over_k) / e[j];
                if (q > best) { best = q; bi = sub * 4 + j; }
            }
            group_argmax<8>(best, bi);
            used = bi;
            if (S.forced) used = S.forced[(int64_t)b * S.ldf + l];
        } else {
            used = S.forced[(int64_t)b * S.ldf + l];
        }
        if (sub == 0) s_z[l] = (uint8_t)used;
        if (slice == 0 && S.logits != nullptr) {
            if (sub == 0) {
                if (S.idx_used) S.idx_used[(int64_t)b * S.ldu + l] = (uint8_t)used;
                if (S.idx_sampled) S.idx_sampled[(int64_t)b * S.ldsm + l] = (uint8_t)bi;
            }
            const int k0 = sub * 4;
            if (S.z_dense)
                *reinterpret_cast<float4*>(S.z_dense + (int64_t)b * S.ldzd + l * 32 + k0) =
                    make_float4(used == k0 ? 1.f : 0.f, used == k0 + 1 ? 1.f : 0.f, used == k0 + 2 ? 1.f : 0.f, used == k0 + 3 ? 1.f : 0.f);
            if (S.logits_out) *reinterpret_cast<float4*>(S.logits_out + (int64_t)b * S.ldlo + l * 32 + k0) = lg;
        }
    }
    if (P.ng == 0) return;
    __syncthreads();
    STAMP();
    // table rows: (cw / 4) threads cover a slice of one table row, 256 / (cw / 4) latents in flight
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        if (gi >= P.ng) break;
        const GatherJob& J = P.g[gi];
        const int tpr = J.cw >> 2;                  // threads per table-row slice
        const int lpp = 256 / tpr;                  // latents per pass
        const int ls = tid / tpr, c4 = (tid - ls * tpr) * 4;
        const int n = slice * J.cw + c4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ls < lpp && n < J.N) {
            for (int l = ls; l < S.L; l += lpp) {
                const float4 u = *reinterpret_cast<const float4*>(J.tab + (int64_t)(l * 32 + s_z[l]) * J.ldt + n);
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            }
        }
        if (ls < lpp) *reinterpret_cast<float4*>(&s_red[gi][ls * J.cw + c4]) = a;
    }
    __syncthreads();
    STAMP();
    STAMP_FLUSH(3000ull);
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        if (gi >= P.ng) break;
        const GatherJob& J = P.g[gi];
        const int lpp = 256 / (J.cw >> 2);
        const int n = slice * J.cw + tid;
        float a = 0.f;
        if (tid < J.cw && n < J.N) {
            a = J.base[(int64_t)b * J.ldb + n];
            const int live = lpp < S.L ? lpp : S.L;
            for (int i = 0; i < live; ++i) a += s_red[gi][i * J.cw + tid];
            J.out[(int64_t)b * J.ldo + n] = a;
        }
        if (J.stats != nullptr) {           // uniform: every thread takes part in the sums
            // the slice's sum, then its sum of squared deviations from the slice mean (two rounds over the four waves)
            const bool mine = tid < J.cw && n < J.N;
            float st[1] = {a};
            all_sum64<1>(st);
            __syncthreads();                // s_red[gi] has been read by every thread of this pass
            if ((tid & 63) == 0) s_red[gi][tid >> 6] = st[0];
            __syncthreads();
            const float sm = (s_red[gi][0] + s_red[gi][1]) + (s_red[gi][2] + s_red[gi][3]);
            const int cnt = min(J.cw, J.N - slice * J.cw);
            const float dv = mine ? a - sm / (float)max(cnt, 1) : 0.f;
            float sq[1] = {dv * dv};
            all_sum64<1>(sq);
            __syncthreads();
            if ((tid & 63) == 0) s_red[gi][4 + (tid >> 6)] = sq[0];
            __syncthreads();
            if (tid == 0)
                *reinterpret_cast<float2*>(J.stats + ((int64_t)b * gridDim.x + slice) * 2) =
                    make_float2(sm, (s_red[gi][4] + s_red[gi][5]) + (s_red[gi][6] + s_red[gi][7]));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out[r, n] = base[r, n] + sum_l tab[(l*32 + z[r, l]), n] for many rows: the one-hot part of a first layer over
// cat[h, z] (heads over all B*H imagined states: lib/world_model.py:215-216, 259-260, lib/critic.py value path; per
// step: the actor's first layer and the GRU input row, lib/actor.py:94-99, lib/world_model.py:64-70).  Gathering the 32 table
// rows per output row from L2 costs more than the product over h itself; here a workgroup owns a slab of 32 columns of
// one table, copies it into LDS once (L*32 rows x 128 bytes <= 128 KB) and walks its share of the rows, so the gather
// runs at LDS speed and global memory sees one read-modify-write of the output.  Optionally leaves the row statistics
// (sum, sum of squared deviations from the slab's mean: one part per slab) for a consumer that normalises while loading.
// ------------------------------------------------------------------------------------------------
struct TableAddJob {
    const float* tab; int64_t ldt;
    const float* base; int64_t ldb;     // out = base + sum (base may be out itself; ldb = 0: one vector for every row)
    float* out; int64_t ldo;
    float* stats;                       // or nullptr
    int N, slab_begin;                  // columns (multiple of 32), first blockIdx.x of this job
};
struct TableAddArgs {
    TableAddJob j[3];
    int nj;
    const uint8_t* zidx; int64_t ldz;   // (rows, L) class indices
    int rows, L;
};
constexpr int TABLE_ADD_MAX_L = 32;

__global__ __launch_bounds__(1024) void k_table_add(const TableAddArgs P) {
    __shared__ __attribute__((aligned(16))) float s_tab[TABLE_ADD_MAX_L * 32 * 32];
    const int tid = threadIdx.x;
    int job = P.nj - 1;
    while (job > 0 && (int)blockIdx.x < P.j[job].slab_begin) --job;
    const TableAddJob& J = P.j[job];
    const int slab = blockIdx.x - J.slab_begin, parts = J.N >> 5;
    const int n0 = slab * 32;
    const int c4 = (tid & 7) * 4;
    for (int r = tid >> 3; r < P.L * 32; r += 128)
        *reinterpret_cast<float4*>(&s_tab[r * 32 + c4]) = *reinterpret_cast<const float4*>(J.tab + (int64_t)r * J.ldt + n0 + c4);
    __syncthreads();
    // rows of this workgroup: a contiguous share; 128 rows per pass, 8 threads per row
    const int per = (P.rows + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(r0 + per, P.rows);
    const int nw = (P.L + 3) >> 2;      // index words per row
    auto load_z = [&](int r, uint32_t (&zw)[8]) {
        const uint8_t* zp = P.zidx + (int64_t)r * P.ldz;
        if (((reinterpret_cast<uintptr_t>(zp) | (uintptr_t)P.L) & 3) == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) zw[i] = i < nw ? reinterpret_cast<const uint32_t*>(zp)[i] : 0u;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t w = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) w |= (4 * i + b < P.L ? (uint32_t)zp[4 * i + b] : 0u) << (8 * b);
                zw[i] = w;
            }
        }
    };
    int r = r0 + (tid >> 3);
    uint32_t zw[8], zn[8];
    float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
    if (r < r1) {
        load_z(r, zw);
        cur = *reinterpret_cast<const float4*>(J.base + (int64_t)r * J.ldb + n0 + c4);
    }
    for (; r < r1; r += 128) {
        const int rn = r + 128;
        if (rn < r1) {                  // the next pass's operands travel while this pass reads LDS
            load_z(rn, zn);
            nxt = *reinterpret_cast<const float4*>(J.base + (int64_t)rn * J.ldb + n0 + c4);
        }
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int l = 4 * i + b;
                if (l < P.L) {
                    const int k = (zw[i] >> (8 * b)) & 31;
                    const float4 u = *reinterpret_cast<const float4*>(&s_tab[(l * 32 + k) * 32 + c4]);
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                }
            }
        }
        const float4 y = make_float4(cur.x + a.x, cur.y + a.y, cur.z + a.z, cur.w + a.w);
        *reinterpret_cast<float4*>(J.out + (int64_t)r * J.ldo + n0 + c4) = y;
        if (J.stats != nullptr) {       // uniform per workgroup: the slab's sum and sum of squared deviations from its mean
            const float sm = group_sum<8>((y.x + y.y) + (y.z + y.w));
            const float mp = sm * (1.0f / 32.0f);
            const float d0 = y.x - mp, d1 = y.y - mp, d2 = y.z - mp, d3 = y.w - mp;
            const float sq = group_sum<8>((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            if ((tid & 7) == 0)
                *reinterpret_cast<float2*>(J.stats + ((int64_t)r * parts + slab) * 2) = make_float2(sm, sq);
        }
        cur = nxt;
#pragma unroll
        for (int i = 0; i < 8; ++i) zw[i] = zn[i];
    }
}

// bf16 images of the weight operands of one call (precision 2: three planes, the exact split; precision 1: one plane,
// the rounded weights): made on first use inside the call, found again by (pointer, shape) afterwards.  The area lives
// in the caller's workspace.
struct SplitCache {
    uint16_t* base; size_t capacity, used;            // in bf16 elements
    int planes;
    struct Entry { const float* W; int64_t ldw; int N, K; const uint16_t* planes; } e[64];
    int n;
};
static thread_local SplitCache t_split_store{};
static thread_local SplitCache* t_split = nullptr;
void big_image_cache_begin(void* base, size_t bytes, int planes) {
    t_split_store = SplitCache{};
    t_split_store.base = reinterpret_cast<uint16_t*>(base);
    t_split_store.capacity = bytes / sizeof(uint16_t);
    t_split_store.planes = planes;
    t_split = base != nullptr && bytes > 0 ? &t_split_store : nullptr;
}
void big_image_cache_end() { t_split = nullptr; }
static const uint16_t* split_planes_of(const ProblemSpec& q, hipStream_t stream) {
    SplitCache* c = t_split;
    if (c == nullptr) return nullptr;
    for (int i = 0; i < c->n; ++i)
        if (c->e[i].W == q.W && c->e[i].ldw == q.ldw && c->e[i].N == q.N && c->e[i].K == q.K) return c->e[i].planes;
    const size_t need = ((size_t)c->planes * q.N * q.K + 7) & ~(size_t)7;      // keeps every image 16-byte aligned
    if (c->n >= 64 || c->used + need > c->capacity || q.K % 4 != 0) return nullptr;
    uint16_t* planes = c->base + c->used;
    const int64_t total4 = (int64_t)q.N * q.K / 4;
    const int64_t blocks = (total4 + 255) / 256;
    k_split_planes<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream>>>(q.W, q.ldw, q.N, q.K, planes, c->planes);
    if (hipGetLastError() != hipSuccess) return nullptr;
    c->e[c->n++] = {q.W, q.ldw, q.N, q.K, planes};
    c->used += need;
    return planes;
}

// compute units of the current device (looked up once per device)
static int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

int launch_big(ProblemSpec* specs, int np, int M, const uint8_t* zidx, int64_t ldz, int latents, int classes,
               hipStream_t stream, int precision) {
    DRM_REQUIRE(np >= 1 && np <= 4 && M >= 1, DRM_EINVAL, "np=%d M=%d", np, M);
    // few rows, plain problems: memory-bound; the K-looped few-row kernel cuts them into many more workgroups than
    // 64-row tiles would (fp32, or bf16 weight images widened to fp32 in the kernel)
    if (M <= SKINNY_MAX_ROWS && precision != 2 && std::getenv("DRM_NO_SKINNY_LOOP") == nullptr) {
        bool ok = true;
        for (int i = 0; i < np; ++i) ok = ok && specs[i].X && specs[i].W && specs[i].Y && skinny_loop_ok(specs[i]);
        if (ok) {
            const uint16_t* images[4] = {nullptr, nullptr, nullptr, nullptr};
            bool have = precision == 1 && t_split != nullptr && t_split->planes == 1;
            for (int i = 0; i < np && have; ++i) have = (images[i] = split_planes_of(specs[i], stream)) != nullptr;
            return launch_skinny_loop(specs, np, M, stream, precision == 1, have ? images : nullptr);
        }
    }
    BigLaunch L{};
    int tiles = 0;
    double fl = 0, by = 0;
    // 64-column tiles when 128-column ones would give fewer than two workgroups per CU (fp32 form)
    int tiles128 = 0;
    for (int i = 0; i < np; ++i) tiles128 += (specs[i].N + BIG_BN - 1) / BIG_BN;
    const bool narrow = precision != 1 && (int64_t)tiles128 * ((M + BIG_BM - 1) / BIG_BM) < 512 &&
                        std::getenv("DRM_NO_NARROW_TILES") == nullptr;
    const int bn = narrow ? BIG_BN64 : BIG_BN;
    for (int i = 0; i < np; ++i) {
        const ProblemSpec& q = specs[i];
        DRM_REQUIRE(q.X && q.W && q.Y && q.K >= 32 && q.K % 32 == 0 && q.N >= 1, DRM_EINVAL, "bad big problem");
        DRM_REQUIRE(aligned16(q.X) && aligned16(q.W) && q.ldx % 4 == 0 && q.ldw % 4 == 0, DRM_EINVAL, "misaligned operand");
        DRM_REQUIRE(!q.add && !q.rowscale, DRM_EINVAL, "large-row GEMM: no add / row-scale epilogue");
        DRM_REQUIRE(!q.ln_g || (precision != 1 && q.ln_b && q.stats_in && q.parts_in >= 1 && q.parts_in <= 16 && q.part_w_in >= 1 &&
                                (int64_t)q.parts_in * q.part_w_in >= q.K && (int64_t)(q.parts_in - 1) * q.part_w_in < q.K),
                    DRM_EINVAL, "large-row GEMM: LayerNorm on load needs the fp32 or the split form and the row statistics "
                    "(1..16 slices whose width covers the row)");
        DRM_REQUIRE(!q.stats_out || precision != 1, DRM_EINVAL, "large-row GEMM: row statistics come from the fp32 and split forms only");
        DRM_REQUIRE(!q.tab || zidx, DRM_EINVAL, "gather table without indices");
        BigProblem& p = L.p[i];
        p.X = q.X; p.ldx = q.ldx; p.W = q.W; p.ldw = q.ldw; p.bias = q.bias; p.tab = q.tab; p.ldt = q.ldt;
        p.Y = q.Y; p.ldy = q.ldy; p.N = q.N; p.K = q.K; p.tile_begin = tiles;
        p.ln_g = q.ln_g; p.ln_b = q.ln_b; p.stats_in = q.stats_in; p.parts_in = q.parts_in; p.stats_out = q.stats_out;
        // slices per row of the statistics: 32 columns per accumulator column block when stored from the accumulators,
        // one per column tile through LDS (problems with a table)
        const int pw_out = q.tab || precision == 2 ? bn : bn / 2;
        p.parts_out = q.stats_out ? (q.N + pw_out - 1) / pw_out : 0;
        p.pw_in = q.part_w_in;
        specs[i].parts_out = p.parts_out;
        specs[i].part_w_out = q.stats_out ? pw_out : 0;
        tiles += (q.N + bn - 1) / bn;
        const double n = q.N, k = q.K, m = M;
        fl += 2.0 * m * n * k + (q.tab ? (double)latents * m * n : 0.0);
        by += 4.0 * (m * k + n * k + n + m * n) + (q.tab ? 4.0 * latents * classes * n + (double)latents * m : 0.0);
    }
    L.np = np; L.M = M; L.zidx = zidx; L.ldz = ldz; L.latents = latents; L.classes = classes;
    bool presplit = (precision == 2 || precision == 1) && t_split != nullptr && t_split->planes == (precision == 2 ? 3 : 1);
    if (presplit) {   // before the timed scope: the images are made once and serve every later launch of the call
        for (int i = 0; i < np && presplit; ++i) presplit = (L.p[i].Ws = split_planes_of(specs[i], stream)) != nullptr;
    }
    ProfScope prof(PC_LINEAR_BIG, stream, fl, by);
    // bf16 and split forms with plenty of tiles: 128-row tiles (two row tiles per wave against the same weight fragments)
    static const int tall_min = [] { const char* e = std::getenv("DRM_SPLIT_TALL_MIN_TILES"); return e ? atoi(e) : 512; }();
    const bool tall = (precision == 1 || precision == 2) && !narrow && (int64_t)tiles * ((M + 127) / 128) >= tall_min;
    const dim3 grid(tiles, tall ? (M + 127) / 128 : (M + BIG_BM - 1) / BIG_BM);
    if (precision == 1 && presplit && tall) k_linear_big_bf16<true, 2><<<grid, dim3(256), 0, stream>>>(L);
    else if (precision == 1 && tall) k_linear_big_bf16<false, 2><<<grid, dim3(256), 0, stream>>>(L);
    else if (precision == 1 && presplit) k_linear_big_bf16<true><<<grid, dim3(256), 0, stream>>>(L);
    else if (precision == 1) k_linear_big_bf16<false><<<grid, dim3(256), 0, stream>>>(L);
    else if (precision == 2) {
        bool any_ln = false;
        for (int i = 0; i < np; ++i) any_ln = any_ln || specs[i].ln_g != nullptr;
#define DRM_SPLIT(PRE, NTV) { if (any_ln) k_linear_big_split<PRE, NTV, true><<<grid, dim3(256), 0, stream>>>(L); \
                              else k_linear_big_split<PRE, NTV, false><<<grid, dim3(256), 0, stream>>>(L); }
        if (tall && presplit) {
            if (any_ln) k_linear_big_split<true, 2, true, 2><<<grid, dim3(256), 0, stream>>>(L);
            else k_linear_big_split<true, 2, false, 2><<<grid, dim3(256), 0, stream>>>(L);
        } else if (tall) {
            if (any_ln) k_linear_big_split<false, 2, true, 2><<<grid, dim3(256), 0, stream>>>(L);
            else k_linear_big_split<false, 2, false, 2><<<grid, dim3(256), 0, stream>>>(L);
        } else if (presplit && narrow) DRM_SPLIT(true, 1) else if (presplit) DRM_SPLIT(true, 2) else if (narrow) DRM_SPLIT(false, 1) else DRM_SPLIT(false, 2)
#undef DRM_SPLIT
    } else {
        // fp32 form: a workgroup walks several tiles when no problem needs the epilogue through LDS and every K is an
        // even number (>= 4) of 32-deep steps; the grid is then what is resident at once
        L.ntx = (int)grid.x; L.nty = (int)grid.y;
        const int64_t ntiles = (int64_t)grid.x * grid.y;
        bool walk = std::getenv("DRM_NO_TILE_WALK") == nullptr;
        for (int i = 0; i < np; ++i) walk = walk && specs[i].tab == nullptr && (specs[i].K / BIG_BK) % 2 == 0 && specs[i].K / BIG_BK >= 4;
        const int64_t resident = (int64_t)cu_count() * 2;
        const unsigned g = (unsigned)(walk && ntiles > resident ? resident : ntiles);
        bool any_ln = false;
        for (int i = 0; i < np; ++i) any_ln = any_ln || specs[i].ln_g != nullptr || specs[i].stats_out != nullptr;
        if (narrow && any_ln) k_linear_big<1, true><<<dim3(g), dim3(256), 0, stream>>>(L);
        else if (narrow) k_linear_big<1, false><<<dim3(g), dim3(256), 0, stream>>>(L);
        else if (any_ln) k_linear_big<2, true><<<dim3(g), dim3(256), 0, stream>>>(L);
        else k_linear_big<2, false><<<dim3(g), dim3(256), 0, stream>>>(L);
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

static int launch_ln_silu(const float* X, int64_t ldx, const float* g, const float* e, float* Y, int64_t ldy, int M, int K,
                          hipStream_t stream) {
    DRM_REQUIRE(K <= 4096 && K % 4 == 0, DRM_ENOTSUP, "LayerNorm width %d", K);
    ProfScope prof(PC_ELEMENTWISE, stream, 12.0 * M * K, 8.0 * M * K);
    const dim3 grid((M + 3) / 4), block(256);
    if (K <= 512) k_ln_silu_rows<2><<<grid, block, 0, stream>>>(X, ldx, g, e, Y, ldy, M, K);
    else if (K <= 1024) k_ln_silu_rows<4><<<grid, block, 0, stream>>>(X, ldx, g, e, Y, ldy, M, K);
    else if (K <= 2048) k_ln_silu_rows<8><<<grid, block, 0, stream>>>(X, ldx, g, e, Y, ldy, M, K);
    else k_ln_silu_rows<16><<<grid, block, 0, stream>>>(X, ldx, g, e, Y, ldy, M, K);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

static int check_dims(const drm_dims& d) {
    DRM_REQUIRE(d.K == 32, DRM_ENOTSUP, "classes per latent must be 32 (got %d)", d.K);
    DRM_REQUIRE(d.D >= 32 && d.D % 32 == 0 && d.M >= 32 && d.M % 32 == 0 && d.D <= 4096 && d.M <= 4096, DRM_ENOTSUP,
                "D=%d and M=%d must be multiples of 32, at most 4096", d.D, d.M);
    DRM_REQUIRE(d.L >= 1 && d.L <= MAX_LATENTS, DRM_ENOTSUP, "latents=%d (max %d)", d.L, MAX_LATENTS);
    DRM_REQUIRE(d.A >= 1 && d.A <= 32, DRM_ENOTSUP, "actions=%d (max 32)", d.A);
    DRM_REQUIRE(d.NB >= 1 && d.NB <= 256, DRM_ENOTSUP, "buckets=%d (max 256)", d.NB);
    DRM_REQUIRE(d.E >= 1, DRM_EINVAL, "E=%d", d.E);
    return DRM_OK;
}

struct PackLayout {
    size_t actor_zT, critic_zT, rew_zT, cont_zT, lin_zT, a_tab, gi_a, total;   // offsets in floats
};
static PackLayout pack_layout(const drm_dims& d) {
    PackLayout p{};
    const size_t LK = (size_t)d.L * d.K;
    size_t o = 0;
    p.actor_zT = o; o += LK * d.M;
    p.critic_zT = o; o += LK * d.M;
    p.rew_zT = o; o += LK * d.M;
    p.cont_zT = o; o += LK * d.M;
    p.lin_zT = o; o += LK * d.D;
    p.a_tab = o; o += (size_t)d.A * d.D;
    o = (o + 3) & ~(size_t)3;
    p.gi_a = o; o += (size_t)d.A * 3 * d.D;
    p.total = (o + 3) & ~(size_t)3;
    return p;
}

struct WsLayout {
    size_t a_pre[2], x, gh, d_pre[2], head[3][2], critic_logits, gi, logits, act[3], a1h, xz, split, split_floats, total;   // offsets in floats
    size_t st_a[2], st_d[2], st_h[3][2];   // row statistics (32 floats per row) of the fp32 large-row path
};
constexpr int BIG_MIN_ROWS = 256;   // from this many trajectories on, the MFMA-bound large-row kernels are used
constexpr int TWO_CHAINS_MIN_ROWS = 512;    // workspaces are sized for two chains from here on
static int two_chains_min_rows() {          // from this many trajectories on, two half-batch chains of launches side by side
    static const int v = [] { const char* e = std::getenv("DRM_TWO_CHAINS_MIN_ROWS"); const int n = e ? atoi(e) : 2048; return n < 512 ? 512 : n; }();
    return v;
}
static WsLayout ws_layout(const drm_dims& d, int B, int H) {
    WsLayout w{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
    w.a_pre[0] = take((size_t)B * d.M); w.a_pre[1] = take((size_t)B * d.M);
    w.x = take((size_t)B * d.D);
    w.gh = take((size_t)B * 3 * d.D);
    w.d_pre[0] = take((size_t)B * d.M); w.d_pre[1] = take((size_t)B * d.M);
    const size_t rows = (size_t)B * H;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) w.head[i][j] = take(rows * d.M);
    w.critic_logits = take(rows * 256);
    const bool wide = d.D > 512 || d.M > 512;
    if (B >= BIG_MIN_ROWS || wide) {
        w.gi = take((size_t)B * 3 * d.D);
        w.logits = take((size_t)B * d.L * d.K);
        for (int i = 0; i < 3; ++i) w.act[i] = take(rows * d.M);
        // three bf16 images of every weight the large-row GEMMs read (precision 2): 1.5 floats per element
        const size_t S = (size_t)d.D + (size_t)d.L * d.K;
        const size_t welems = 5 * (S * d.M + 3 * (size_t)d.M * d.M) + (size_t)d.L * d.K * d.M + 256 * (size_t)d.M +
                              6 * (size_t)d.D * d.D + (size_t)d.D * S;
        w.split_floats = (welems * 3 + 1) / 2;
        w.split = take(w.split_floats);
    }
    if (B <= SKINNY_MAX_ROWS && !wide) {
        w.gi = take((size_t)B * 3 * d.D);
        w.logits = take((size_t)B * d.L * d.K);
    }
    if ((!wide && (B <= SKINNY_MAX_ROWS || B >= BIG_MIN_ROWS)) || (wide && B <= SKINNY_MAX_ROWS && d.M <= 512)) {
        w.a1h = take((size_t)B * d.M);
        w.xz = take((size_t)B * d.D);
        if (B >= BIG_MIN_ROWS) {
            for (int i = 0; i < 2; ++i) { w.st_a[i] = take((size_t)B * 32); w.st_d[i] = take((size_t)B * 32); }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) w.st_h[i][j] = take(rows * 32);
        }
    }
    w.total = o;
    return w;
}

static int check_mlp(const drm_mlp& m, int in_dim, int hidden, int out_dim, const char* name) {
    DRM_REQUIRE(m.n_layers >= 1 && m.n_layers <= DRM_MAX_MLP_LAYERS, DRM_ENOTSUP, "%s: %d layers", name, m.n_layers);
    DRM_REQUIRE(m.in_dim == in_dim && m.hidden == hidden && m.out_dim == out_dim, DRM_EINVAL,
                "%s: dims (%d,%d,%d) expected (%d,%d,%d)", name, m.in_dim, m.hidden, m.out_dim, in_dim, hidden, out_dim);
    for (int i = 0; i < m.n_layers; ++i)
        DRM_REQUIRE(m.w[i] && m.b[i] && m.ln_g[i] && m.ln_b[i], DRM_EINVAL, "%s: null layer %d", name, i);
    DRM_REQUIRE(m.head_w && m.head_b, DRM_EINVAL, "%s: null head", name);
    return DRM_OK;
}

static int check_weights(const drm_imagine_weights& w) {
    if (int rc = check_dims(w.dims)) return rc;
    const drm_dims& d = w.dims;
    const int S = d.D + d.L * d.K;
    DRM_REQUIRE(w.a_emb && w.lin_w && w.lin_b && w.gru_w_ih && w.gru_w_hh && w.gru_b_ih && w.gru_b_hh && w.buckets,
                DRM_EINVAL, "null sequence-model pointer");
    if (int rc = check_mlp(w.dyn, d.D, d.M, d.L * d.K, "dyn")) return rc;
    if (int rc = check_mlp(w.rew, S, d.M, 1, "rew")) return rc;
    if (int rc = check_mlp(w.cont, S, d.M, 1, "cont")) return rc;
    if (int rc = check_mlp(w.actor, S, d.M, d.A, "actor")) return rc;
    if (int rc = check_mlp(w.critic, S, d.M, d.NB, "critic")) return rc;
    return DRM_OK;
}

}  // namespace drm

using namespace drm;

extern "C" size_t drm_pack_imagine_bytes(const drm_dims* dims) {
    if (!dims || check_dims(*dims) != DRM_OK) return 0;
    return pack_layout(*dims).total * sizeof(float);
}

extern "C" int drm_pack_imagine(const drm_imagine_weights* w, float* pack_out, size_t pack_bytes, drm_stream_t stream) {
    DRM_REQUIRE(w && pack_out, DRM_EINVAL, "null pointer");
    if (int rc = check_weights(*w)) return rc;
    const drm_dims& d = w->dims;
    const PackLayout pl = pack_layout(d);
    DRM_REQUIRE(pack_bytes >= pl.total * sizeof(float), DRM_EINVAL, "pack buffer too small");
    DRM_REQUIRE(aligned16(pack_out), DRM_EINVAL, "pack buffer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int LK = d.L * d.K, S = d.D + LK;
    struct { const float* W; int64_t ldw; int col0; int nrows; size_t off; } jobs[5] = {
        {w->actor.w[0], S, d.D, d.M, pl.actor_zT}, {w->critic.w[0], S, d.D, d.M, pl.critic_zT},
        {w->rew.w[0], S, d.D, d.M, pl.rew_zT},     {w->cont.w[0], S, d.D, d.M, pl.cont_zT},
        {w->lin_w, LK + d.E, 0, d.D, pl.lin_zT}};
    {
        ProfScope prof(PC_PACK, s, 2.0 * d.A * d.D * d.E, 8.0 * ((double)LK * (4.0 * d.M + d.D)) + 4.0 * d.A * d.D);
        for (auto& j : jobs) {
            k_transpose_cols<<<dim3((LK + 31) / 32, (j.nrows + 31) / 32), dim3(256), 0, s>>>(j.W, j.ldw, j.col0, LK, j.nrows,
                                                                                             pack_out + j.off);
            DRM_LAUNCH_CHECK();
        }
        k_action_table<<<dim3((d.D + 255) / 256, d.A), dim3(256), 0, s>>>(w->lin_w, LK + d.E, LK, d.E, w->a_emb, d.A, d.D,
                                                                           pack_out + pl.a_tab);
        DRM_LAUNCH_CHECK();
    }
    // gi_a[a] = W_ih a_tab[a]: the action's share of the GRU input product, one row per action
    ProblemSpec g{pack_out + pl.a_tab, d.D, nullptr, nullptr, w->gru_w_ih, d.D, nullptr, nullptr, 0,
                  pack_out + pl.gi_a, 3 * d.D, 3 * d.D, d.D};
    if (d.D <= 512) return launch_skinny(&g, 1, d.A, s);
    return launch_big(&g, 1, d.A, nullptr, 0, 0, 0, s);
}

extern "C" int drm_fused_linear_bf16(const drm_linear_args* a, drm_stream_t stream) {
    DRM_REQUIRE(a, DRM_EINVAL, "null args");
    DRM_REQUIRE(a->M >= 1 && a->ln_gamma == nullptr, DRM_EINVAL, "M=%d; the bf16 product takes activated inputs", a->M);
    DRM_REQUIRE(!a->table || (a->classes >= 1 && a->latents >= 1), DRM_EINVAL, "bad gather shape");
    ProblemSpec s{a->X, a->ldx, nullptr, nullptr, a->W, a->ldw, a->bias, a->table, a->ldt, a->Y, a->ldy, a->N, a->K};
    return launch_big(&s, 1, a->M, a->zidx, a->ldz, a->latents, a->classes, (hipStream_t)stream, 1);
}

extern "C" int drm_fused_linear_split(const drm_linear_args* a, drm_stream_t stream) {
    DRM_REQUIRE(a, DRM_EINVAL, "null args");
    DRM_REQUIRE(a->M >= 1 && a->ln_gamma == nullptr, DRM_EINVAL, "M=%d; the split product takes activated inputs", a->M);
    DRM_REQUIRE(!a->table || (a->classes >= 1 && a->latents >= 1), DRM_EINVAL, "bad gather shape");
    ProblemSpec s{a->X, a->ldx, nullptr, nullptr, a->W, a->ldw, a->bias, a->table, a->ldt, a->Y, a->ldy, a->N, a->K};
    return launch_big(&s, 1, a->M, a->zidx, a->ldz, a->latents, a->classes, (hipStream_t)stream, 2);
}

extern "C" size_t drm_imagine_workspace_bytes(const drm_dims* dims, int32_t B, int32_t H) {
    if (!dims || check_dims(*dims) != DRM_OK || B < 1 || H < 1) return 0;
    size_t bytes = ws_layout(*dims, B, H).total * sizeof(float);
    if (B >= TWO_CHAINS_MIN_ROWS && B % 128 == 0) {      // two chains of B / 2 trajectories, a workspace each
        for (int n = 2; n <= 8; n *= 2) {
            if (B % (64 * n) != 0) break;
            const size_t parts = n * ((ws_layout(*dims, B / n, H).total * sizeof(float) + 255) & ~(size_t)255);
            bytes = parts > bytes ? parts : bytes;
        }
    }
    return bytes;
}

extern "C" int drm_fused_linear_f32(const drm_linear_args* a, drm_stream_t stream) {
    DRM_REQUIRE(a, DRM_EINVAL, "null args");
    DRM_REQUIRE(a->M >= 1, DRM_EINVAL, "M=%d", a->M);
    DRM_REQUIRE(!a->table || (a->classes >= 1 && a->latents >= 1), DRM_EINVAL, "bad gather shape");
    ProblemSpec s{a->X, a->ldx, a->ln_gamma, a->ln_beta, a->W, a->ldw, a->bias, a->table, a->ldt, a->Y, a->ldy, a->N, a->K};
    // many rows, or few rows over a long inner dimension (memory-bound: the K-looped few-row kernel)
    const bool few_long = a->M <= SKINNY_MAX_ROWS && a->K > 512 && a->table == nullptr;
    if ((a->M >= BIG_MIN_ROWS || few_long) && a->ln_gamma == nullptr)
        return launch_big(&s, 1, a->M, a->zidx, a->ldz, a->latents, a->classes, (hipStream_t)stream);
    return launch_store(&s, 1, a->M, a->zidx, a->ldz, a->latents, a->classes, (hipStream_t)stream);
}

// One layer for up to 4 problems that share their rows: small-row kernels apply the producer's
// LayerNorm + SiLU on load; the large-row kernels take activated inputs, so the activation is a pass
// of its own into `act[i]` first.
static int run_linear(ProblemSpec* p, int np, int M, const uint8_t* zidx, int64_t ldz, int latents, int classes, bool big,
                      float* const* act, int act_ld, hipStream_t s, int precision = 0) {
    if (!big) return launch_store(p, np, M, zidx, ldz, latents, classes, s);
    for (int i = 0; i < np; ++i) {
        if (p[i].ln_g == nullptr) continue;
        if (int rc = launch_ln_silu(p[i].X, p[i].ldx, p[i].ln_g, p[i].ln_b, act[i], act_ld, M, p[i].K, s)) return rc;
        p[i].X = act[i]; p[i].ldx = act_ld; p[i].ln_g = nullptr; p[i].ln_b = nullptr;
    }
    return launch_big(p, np, M, zidx, ldz, latents, classes, s, precision);
}

// One chain of launches for io->B trajectories.  `nb`: rows of a time step in the two noise tensors (io->B unless the
// trajectories are a slice of a larger call, whose noise is laid out (H, nb, ...)).
static int imagine_enqueue_part(const drm_imagine_weights* w, const drm_imagine_io* io, drm_stream_t stream, int nb, int row0 = 0) {
    if (int rc = check_weights(*w)) return rc;
    prof_fresh();
    const drm_dims& d = w->dims;
    const int B = io->B, H = io->H, D = d.D, M = d.M, L = d.L, LK = d.L * d.K, S = D + LK;
    DRM_REQUIRE(B >= 1 && H >= 1, DRM_EINVAL, "B=%d H=%d", B, H);
    DRM_REQUIRE((int64_t)B * H < (1 << 30), DRM_ENOTSUP, "B*H too large");
    DRM_REQUIRE(io->h0 && io->zidx0 && io->h_out && io->zidx_out && io->actions_out &&
                    io->rewards_out && io->conts_out && io->values_out && io->v_last_out && io->lam_returns_out,
                DRM_EINVAL, "null io pointer");
    DRM_REQUIRE((io->expo_a != nullptr) == (io->expo_z != nullptr), DRM_EINVAL, "noise: both tensors or neither");
    // no noise tensors: the draws take their Exp(1) numbers from the counter-based generator inside the kernels
    const bool device_rng = io->expo_a == nullptr;
    RngArgs rng{};
    rng.seed_lo = (uint32_t)io->rng_seed; rng.seed_hi = (uint32_t)(io->rng_seed >> 32);
    rng.state = io->rng_state; rng.counter = io->rng_counter; rng.rows_total = nb; rng.row0 = row0;
    auto rng_at = [&](int t) { RngArgs r = rng; r.t = (uint32_t)t; return r; };
    auto noise_a = [&](int t) -> const float* { return device_rng ? nullptr : io->expo_a + (size_t)t * nb * w->dims.A; };
    auto noise_z = [&](int t) -> const float* { return device_rng ? nullptr : io->expo_z + (size_t)t * nb * (size_t)(w->dims.L * w->dims.K); };
    const PackLayout pl = pack_layout(d);
    DRM_REQUIRE(w->pack && w->pack_bytes >= pl.total * sizeof(float), DRM_EINVAL, "weights not packed (drm_pack_imagine)");
    const WsLayout wl = ws_layout(d, B, H);
    DRM_REQUIRE(io->workspace && io->workspace_bytes >= wl.total * sizeof(float) && aligned16(io->workspace), DRM_EINVAL,
                "workspace too small or misaligned: need %zu bytes", wl.total * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)io->workspace;
    const float* pack = w->pack;
    const float l1m = (float)log1p(-(double)d.unimix_eps);
    const float lek = (float)(log((double)d.unimix_eps) - log((double)d.K));
    const float lea = (float)(log((double)d.unimix_eps) - log((double)d.A));
    DRM_REQUIRE(io->precision >= 0 && io->precision <= 2, DRM_EINVAL, "precision=%d", io->precision);
    const int prec = io->precision;
    const bool wide = D > 512 || M > 512;                // beyond the register-resident small-row kernels
    const bool big = B >= BIG_MIN_ROWS || wide;
    float* act[3] = {big ? ws + wl.act[0] : nullptr, big ? ws + wl.act[1] : nullptr, big ? ws + wl.act[2] : nullptr};

    // precision 1: one rounded bf16 image per weight, made once per call (half the weight bytes per step).  Precision 2
    // splits the fp32 weights on their way into LDS like the activations: three bf16 planes made up front are 6 bytes per
    // weight through the vector caches instead of 4, and operand bytes are what these kernels wait for (6.28 -> 6.08 ms per
    // 4096-trajectory block once the split itself became cheap); DRM_SPLIT_WEIGHT_IMAGES=1 brings the planes back.
    struct ImageScope { ~ImageScope() { big_image_cache_end(); } } image_scope;
    if ((prec == 1 || (prec == 2 && std::getenv("DRM_SPLIT_WEIGHT_IMAGES") != nullptr)) && big && std::getenv("DRM_NO_WEIGHT_IMAGES") == nullptr)
        big_image_cache_begin(ws + wl.split, wl.split_floats * sizeof(float), prec == 2 ? 3 : 1);

    const bool few = B <= SKINNY_MAX_ROWS && !wide;
    // fp32 products at many rows take the same launch sequence with the large-row kernel in place of the few-row one:
    // LayerNorm + SiLU applied by the consumer while it loads (row statistics left behind by the producer)
    const bool fused = big && !wide && (prec == 0 || prec == 2) && std::getenv("DRM_NO_FUSED_STEP") == nullptr;
    // few rows of a wide model (hidden width still within the register-resident kernel): the same launch sequence; the
    // products over the recurrent width are memory-bound and run on the K-looped few-row kernel (bf16 weight images at
    // precision 1), the LayerNorm'd ones over the hidden width on the register-resident one
    const bool few_wide = B <= SKINNY_MAX_ROWS && wide && M <= 512 && prec != 2 && std::getenv("DRM_NO_FEW_WIDE") == nullptr;
    auto lin = [&](ProblemSpec* p, int np) -> int {
        if (few) return launch_skinny(p, np, B, s);
        if (!few_wide) return launch_big(p, np, B, nullptr, 0, 0, 0, s, prec);
        for (int i = 0; i < np;) {
            const bool reg = p[i].K <= 512;
            int j = i + 1;
            while (j < np && (p[j].K <= 512) == reg) ++j;
            if (int rc = reg ? launch_skinny(p + i, j - i, B, s) : launch_big(p + i, j - i, B, nullptr, 0, 0, 0, s, prec)) return rc;
            i = j;
        }
        return DRM_OK;
    };
    float* st_a[2] = {fused ? ws + wl.st_a[0] : nullptr, fused ? ws + wl.st_a[1] : nullptr};
    float* st_d[2] = {fused ? ws + wl.st_d[0] : nullptr, fused ? ws + wl.st_d[1] : nullptr};
    int parts_a[2] = {0, 0}, parts_d[2] = {0, 0};     // slices per row of the statistics in st_a / st_d ...
    int pw_a[2] = {0, 0}, pw_d[2] = {0, 0};           // ... and their width in columns
    if (few || fused || few_wide) {
        // ---- step-wise path.  Few rows: every layer is cut into ~256 workgroups; the draw of z and the one-hot
        // tables it selects (actor layer 1, GRU input) share one launch
        const int last_a = w->actor.n_layers - 1, last_d = w->dyn.n_layers - 1;
        // many rows: the draw and the table rows are two launches, the tables' column slabs held in LDS by the second
        const bool step_table_add = fused && L <= TABLE_ADD_MAX_L && std::getenv("DRM_NO_TABLE_ADD") == nullptr;
        auto sample_gather = [&](int t, bool draw, bool gather) -> int {
            if (step_table_add) {
                if (draw) {
                    SampleRowsArgs a{};
                    a.logits = ws + wl.logits; a.ldlg = LK; a.expo = noise_z(t); a.lde = LK; a.rng = rng_at(t);
                    a.forced = io->forced_zidx ? io->forced_zidx + (size_t)t * L : nullptr; a.ldf = (int64_t)H * L;
                    a.idx_used = io->zidx_out + (size_t)t * L; a.ldu = (int64_t)H * L;
                    a.idx_sampled = io->zidx_sampled_out ? io->zidx_sampled_out + (size_t)t * L : nullptr; a.ldsm = (int64_t)H * L;
                    a.z_dense = io->z_out ? io->z_out + (size_t)t * LK : nullptr; a.ldzd = (int64_t)H * LK;
                    a.logits_out = io->prior_logits_out ? io->prior_logits_out + (size_t)t * LK : nullptr; a.ldlo = (int64_t)H * LK;
                    a.B = B; a.L = L; a.log1m_eps = l1m; a.log_eps_over_k = lek;
                    a.one_m_eps = 1.0f - d.unimix_eps; a.eps_over_k = d.unimix_eps / (float)d.K;
                    const int64_t blocks = ((int64_t)B * L + 31) / 32;
                    ProfScope prof(PC_SAMPLE, s, 12.0 * B * LK, (double)B * L * (256.0 + (a.z_dense ? 128.0 : 0.0) + 1.0));
                    k_sample_rows<<<dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s>>>(a);
                    DRM_LAUNCH_CHECK();
                }
                if (gather) {
                    TableAddArgs a{};
                    a.nj = 2; a.rows = B; a.L = L;
                    a.zidx = draw ? io->zidx_out + (size_t)t * L : io->zidx0; a.ldz = draw ? (int64_t)H * L : L;
                    a.j[0] = TableAddJob{pack + pl.actor_zT, M, ws + wl.a1h, M, ws + wl.a_pre[0], M, st_a[0], M, 0};
                    a.j[1] = TableAddJob{pack + pl.lin_zT, D, w->lin_b, 0, ws + wl.xz, D, nullptr, D, M / 32};
                    parts_a[0] = M / 32; pw_a[0] = 32;
                    const int slabs = M / 32 + D / 32;
                    int split = cu_count() / slabs > 0 ? cu_count() / slabs : 1;   // one workgroup per CU (128 KB of LDS each): a single round
                    if (split > (B + 127) / 128) split = (B + 127) / 128;
                    ProfScope prof(PC_SAMPLE, s, (double)B * L * (M + D), 4.0 * ((double)B * (3.0 * M + D) + (double)L * 32 * (M + D)) + (double)B * L);
                    k_table_add<<<dim3(slabs, split), dim3(1024), 0, s>>>(a);
                    DRM_LAUNCH_CHECK();
                }
                return DRM_OK;
            }
            SampleGatherArgs a{};
            a.s.B = B; a.s.L = L; a.s.log1m_eps = l1m; a.s.log_eps_over_k = lek;
            a.one_m_eps = 1.0f - d.unimix_eps; a.eps_over_k = d.unimix_eps / (float)d.K;
            if (draw) {
                a.s.logits = ws + wl.logits; a.s.ldlg = LK; a.s.expo = noise_z(t); a.s.lde = LK; a.s.rng = rng_at(t);
                a.s.forced = io->forced_zidx ? io->forced_zidx + (size_t)t * L : nullptr; a.s.ldf = (int64_t)H * L;
                a.s.idx_used = io->zidx_out + (size_t)t * L; a.s.ldu = (int64_t)H * L;
                a.s.idx_sampled = io->zidx_sampled_out ? io->zidx_sampled_out + (size_t)t * L : nullptr; a.s.ldsm = (int64_t)H * L;
                a.s.z_dense = io->z_out ? io->z_out + (size_t)t * LK : nullptr; a.s.ldzd = (int64_t)H * LK;
                a.s.logits_out = io->prior_logits_out ? io->prior_logits_out + (size_t)t * LK : nullptr; a.s.ldlo = (int64_t)H * LK;
            } else {
                a.s.forced = io->zidx0; a.s.ldf = L;
            }
            int slices = 1;
            if (gather) {
                // columns per slice so that B x slices is about the number of CUs
                const int nmax = M > D ? M : D;
                int cw = 4;
                while (cw < (fused ? 256 : 64) && (int64_t)B * ((nmax + cw - 1) / cw) > 320) cw *= 2;
                slices = (nmax + cw - 1) / cw;
                a.ng = 2;
                a.g[0] = GatherJob{pack + pl.actor_zT, M, ws + wl.a1h, M, ws + wl.a_pre[0], M, M, cw};
                a.g[0].stats = st_a[0];       // the consumer of a_pre[0] (actor layer 2) normalises on load
                parts_a[0] = slices; pw_a[0] = cw;
                a.g[1] = GatherJob{pack + pl.lin_zT, D, w->lin_b, 0, ws + wl.xz, D, D, cw};
            }
            ProfScope prof(PC_SAMPLE, s, (draw ? 40.0 * B * LK : 0.0) + (gather ? (double)B * L * (M + D) : 0.0),
                           (draw ? (double)B * L * (256.0 + (a.s.z_dense ? 128.0 : 0.0) + 1.0) : 0.0) +
                               (gather ? 4.0 * B * (L + 2.0) * (M + D) : 0.0));
            k_sample_gather<<<dim3(slices, B), dim3(256), 0, s>>>(a);
            DRM_LAUNCH_CHECK();
            return DRM_OK;
        };
        {
            ProblemSpec p[2] = {
                {io->h0, D, nullptr, nullptr, w->gru_w_hh, D, w->gru_b_hh, nullptr, 0, ws + wl.gh, 3 * D, 3 * D, D},
                {io->h0, D, nullptr, nullptr, w->actor.w[0], S, w->actor.b[0], nullptr, 0, ws + wl.a1h, M, M, D}};
            if (int rc = lin(p, 2)) return rc;
            if (int rc = sample_gather(0, false, true)) return rc;
        }
        for (int t = 0; t < H; ++t) {
            const float* h_t = t == 0 ? io->h0 : io->h_out + (size_t)(t - 1) * D;
            const int64_t ldh = t == 0 ? D : (int64_t)H * D;
            // policy MLP; the action-free part of the GRU input product rides along with its first launch
            const ProblemSpec giz{ws + wl.xz, D, nullptr, nullptr, w->gru_w_ih, D, nullptr, nullptr, 0, ws + wl.gi, 3 * D, 3 * D, D};
            if (last_a == 0) {
                ProblemSpec p = giz;
                if (int rc = lin(&p, 1)) return rc;
            }
            for (int i = 1; i <= last_a; ++i) {
                ProblemSpec p[2] = {{ws + wl.a_pre[(i - 1) & 1], M, w->actor.ln_g[i - 1], w->actor.ln_b[i - 1], w->actor.w[i], M,
                                     w->actor.b[i], nullptr, 0, ws + wl.a_pre[i & 1], M, M, M},
                                    giz};
                p[0].stats_in = st_a[(i - 1) & 1]; p[0].parts_in = parts_a[(i - 1) & 1]; p[0].part_w_in = pw_a[(i - 1) & 1];
                p[0].stats_out = i < last_a ? st_a[i & 1] : nullptr;      // the last layer's consumer normalises by itself
                if (int rc = lin(p, i == 1 ? 2 : 1)) return rc;
                parts_a[i & 1] = p[0].parts_out; pw_a[i & 1] = p[0].part_w_out;
            }
            float* h_new = io->h_out + (size_t)t * D;
            {
                ActorGatesArgs a{};
                a.pre = ws + wl.a_pre[last_a & 1]; a.ldp = M;
                a.ln_g = w->actor.ln_g[last_a]; a.ln_b = w->actor.ln_b[last_a];
                a.head_w = w->actor.head_w; a.head_b = w->actor.head_b; a.M = M; a.A = d.A;
                a.expo = noise_a(t); a.lde = d.A; a.rng = rng_at(t);
                a.forced = io->forced_actions ? io->forced_actions + t : nullptr; a.ldf = H;
                a.one_m_eps = 1.0f - d.unimix_eps; a.eps_over_a = d.unimix_eps / (float)d.A;
                a.gi_z = ws + wl.gi; a.ldgi = 3 * D; a.gi_a = pack + pl.gi_a; a.b_ih = w->gru_b_ih;
                a.gh = ws + wl.gh; a.ldgh = 3 * D; a.h_prev = h_t; a.ldh = ldh; a.h_new = h_new; a.ldhn = (int64_t)H * D; a.D = D;
                a.a_used = io->actions_out + t; a.ldau = H;
                a.a_sampled = io->actions_sampled_out ? io->actions_sampled_out + t : nullptr; a.ldas = H;
                a.logp_out = io->actor_logp_out ? io->actor_logp_out + (size_t)t * d.A : nullptr; a.ldlp = (int64_t)H * d.A;
                ProfScope prof(PC_ACTOR_HEAD, s, 2.0 * B * M * d.A + 14.0 * B * D,
                               4.0 * ((double)B * M + (double)d.A * M + 2.0 * M + (double)B * D * 11.0 + 3.0 * D));
                const dim3 grid(B, (D + 511) / 512);
                if (d.A <= 8) k_actor_gates<8><<<grid, dim3(256), 0, s>>>(a);
                else k_actor_gates<32><<<grid, dim3(256), 0, s>>>(a);
                DRM_LAUNCH_CHECK();
            }
            {
                const bool more = t + 1 < H;
                ProblemSpec p[3] = {
                    {h_new, (int64_t)H * D, nullptr, nullptr, w->dyn.w[0], D, w->dyn.b[0], nullptr, 0, ws + wl.d_pre[0], M, M, D},
                    {h_new, (int64_t)H * D, nullptr, nullptr, w->gru_w_hh, D, w->gru_b_hh, nullptr, 0, ws + wl.gh, 3 * D, 3 * D, D},
                    {h_new, (int64_t)H * D, nullptr, nullptr, w->actor.w[0], S, w->actor.b[0], nullptr, 0, ws + wl.a1h, M, M, D}};
                p[0].stats_out = st_d[0];
                if (int rc = lin(p, more ? 3 : 1)) return rc;
                parts_d[0] = p[0].parts_out; pw_d[0] = p[0].part_w_out;
            }
            for (int i = 1; i <= last_d; ++i) {
                ProblemSpec p{ws + wl.d_pre[(i - 1) & 1], M, w->dyn.ln_g[i - 1], w->dyn.ln_b[i - 1], w->dyn.w[i], M, w->dyn.b[i],
                              nullptr, 0, ws + wl.d_pre[i & 1], M, M, M};
                p.stats_in = st_d[(i - 1) & 1]; p.parts_in = parts_d[(i - 1) & 1]; p.part_w_in = pw_d[(i - 1) & 1];
                p.stats_out = st_d[i & 1];
                if (int rc = lin(&p, 1)) return rc;
                parts_d[i & 1] = p.parts_out; pw_d[i & 1] = p.part_w_out;
            }
            {
                ProblemSpec p{ws + wl.d_pre[last_d & 1], M, w->dyn.ln_g[last_d], w->dyn.ln_b[last_d], w->dyn.head_w, M,
                              w->dyn.head_b, nullptr, 0, ws + wl.logits, LK, LK, M};
                p.stats_in = st_d[last_d & 1]; p.parts_in = parts_d[last_d & 1]; p.part_w_in = pw_d[last_d & 1];
                if (int rc = lin(&p, 1)) return rc;
            }
            if (int rc = sample_gather(t, true, t + 1 < H)) return rc;
        }
    } else {
    DRM_REQUIRE(!device_rng, DRM_ENOTSUP, "in-kernel noise is built into the step-wise launch sequence only (few rows, or many rows at "
                "widths up to 512 with fp32 / split products): pass expo_a / expo_z for this shape");
    // gh for step 0
    {
        ProblemSpec p{io->h0, D, nullptr, nullptr, w->gru_w_hh, D, w->gru_b_hh, nullptr, 0, ws + wl.gh, 3 * D, 3 * D, D};
        if (int rc = run_linear(&p, 1, B, nullptr, 0, 0, 0, big, act, M, s, prec)) return rc;
    }
    for (int t = 0; t < H; ++t) {
        const float* h_t = t == 0 ? io->h0 : io->h_out + (size_t)(t - 1) * D;
        const int64_t ldh = t == 0 ? D : (int64_t)H * D;
        const uint8_t* z_t = t == 0 ? io->zidx0 : io->zidx_out + (size_t)(t - 1) * L;
        const int64_t ldz = t == 0 ? L : (int64_t)H * L;
        // actor MLP
        for (int i = 0; i < w->actor.n_layers; ++i) {
            ProblemSpec p{};
            if (i == 0) p = ProblemSpec{h_t, ldh, nullptr, nullptr, w->actor.w[0], S, w->actor.b[0], pack + pl.actor_zT, M,
                                        ws + wl.a_pre[0], M, M, D};
            else p = ProblemSpec{ws + wl.a_pre[(i - 1) & 1], M, w->actor.ln_g[i - 1], w->actor.ln_b[i - 1], w->actor.w[i], M,
                                 w->actor.b[i], nullptr, 0, ws + wl.a_pre[i & 1], M, M, M};
            if (int rc = run_linear(&p, 1, B, z_t, ldz, L, d.K, big, act, M, s, prec)) return rc;
        }
        // actor head, action draw, GRU input row
        {
            const int last = w->actor.n_layers - 1;
            ActorHeadArgs a{};
            a.pre = ws + wl.a_pre[last & 1]; a.ldp = M;
            a.ln_g = w->actor.ln_g[last]; a.ln_b = w->actor.ln_b[last];
            a.head_w = w->actor.head_w; a.head_b = w->actor.head_b; a.M = M; a.A = d.A;
            a.expo = io->expo_a + (size_t)t * nb * d.A; a.lde = d.A;
            a.forced = io->forced_actions ? io->forced_actions + t : nullptr; a.ldf = H;
            a.log1m_eps = l1m; a.log_eps_over_a = lea;
            a.zidx = z_t; a.ldz = ldz; a.latents = L; a.classes = d.K;
            a.lin_zT = pack + pl.lin_zT; a.lin_b = w->lin_b; a.a_tab = pack + pl.a_tab; a.D = D;
            a.x = ws + wl.x; a.ldxo = D;
            a.a_used = io->actions_out + t; a.ldau = H;
            a.a_sampled = io->actions_sampled_out ? io->actions_sampled_out + t : nullptr; a.ldas = H;
            a.logp_out = io->actor_logp_out ? io->actor_logp_out + (size_t)t * d.A : nullptr; a.ldlp = (int64_t)H * d.A;
            ProfScope prof(PC_ACTOR_HEAD, s, 2.0 * B * M * d.A + (double)B * D * (L + 1),
                           4.0 * ((double)B * M + (double)d.A * M + (double)B * D * (L + 2)));
            if (wide) k_actor_head_wide<<<dim3(B), dim3(256), 0, s>>>(a);
            else if (d.A <= 8 && L <= 32) k_actor_head_x<8, 32><<<dim3(B), dim3(256), 0, s>>>(a);
            else if (L <= 32) k_actor_head_x<32, 32><<<dim3(B), dim3(256), 0, s>>>(a);
            else k_actor_head_x<32, 64><<<dim3(B), dim3(256), 0, s>>>(a);
            DRM_LAUNCH_CHECK();
        }
        // GRU: gi = W_ih x, gates with the precomputed gh -> h_{t+1} written into h_out[:, t]
        float* h_new = io->h_out + (size_t)t * D;
        if (!big) {
            ProblemSpec p{ws + wl.x, D, nullptr, nullptr, w->gru_w_ih, D, nullptr, nullptr, 0, h_new, (int64_t)H * D, 3 * D, D};
            GruEpilogue G{w->gru_b_ih, ws + wl.gh, 3 * D, h_t, ldh, h_new, (int64_t)H * D, D, nullptr, 0, nullptr, nullptr, 0};
            if (int rc = launch_gru(p, B, G, s)) return rc;
        } else {
            ProblemSpec p{ws + wl.x, D, nullptr, nullptr, w->gru_w_ih, D, nullptr, nullptr, 0, ws + wl.gi, 3 * D, 3 * D, D};
            if (int rc = launch_big(&p, 1, B, nullptr, 0, 0, 0, s, prec)) return rc;
            ProfScope prof(PC_ELEMENTWISE, s, 20.0 * B * D, 4.0 * 8.0 * B * D);
            k_gru_gates<<<dim3((unsigned)(((int64_t)B * D + 255) / 256)), dim3(256), 0, s>>>(
                ws + wl.gi, 3 * D, w->gru_b_ih, ws + wl.gh, 3 * D, h_t, ldh, h_new, (int64_t)H * D, B, D);
            DRM_LAUNCH_CHECK();
        }
        // dyn layer 1 and (for the next step) gh = W_hh h_{t+1} + b_hh share their input
        {
            ProblemSpec p[2] = {
                {h_new, (int64_t)H * D, nullptr, nullptr, w->dyn.w[0], D, w->dyn.b[0], nullptr, 0, ws + wl.d_pre[0], M, M, D},
                {h_new, (int64_t)H * D, nullptr, nullptr, w->gru_w_hh, D, w->gru_b_hh, nullptr, 0, ws + wl.gh, 3 * D, 3 * D, D}};
            if (int rc = run_linear(p, t + 1 < H ? 2 : 1, B, nullptr, 0, 0, 0, big, act, M, s, prec)) return rc;
        }
        for (int i = 1; i < w->dyn.n_layers; ++i) {
            ProblemSpec p{ws + wl.d_pre[(i - 1) & 1], M, w->dyn.ln_g[i - 1], w->dyn.ln_b[i - 1], w->dyn.w[i], M, w->dyn.b[i],
                          nullptr, 0, ws + wl.d_pre[i & 1], M, M, M};
            if (int rc = run_linear(&p, 1, B, nullptr, 0, 0, 0, big, act, M, s, prec)) return rc;
        }
        // dyn head + unimix + Gumbel-argmax draw of z_{t+1}
        if (big) {
            const int last = w->dyn.n_layers - 1;
            ProblemSpec p{ws + wl.d_pre[last & 1], M, w->dyn.ln_g[last], w->dyn.ln_b[last], w->dyn.head_w, M, w->dyn.head_b,
                          nullptr, 0, ws + wl.logits, LK, LK, M};
            if (int rc = run_linear(&p, 1, B, nullptr, 0, 0, 0, true, act, M, s, prec)) return rc;
            SampleRowsArgs a{};
            a.logits = ws + wl.logits; a.ldlg = LK; a.expo = io->expo_z + (size_t)t * nb * LK; a.lde = LK;
            a.forced = io->forced_zidx ? io->forced_zidx + (size_t)t * L : nullptr; a.ldf = (int64_t)H * L;
            a.idx_used = io->zidx_out + (size_t)t * L; a.ldu = (int64_t)H * L;
            a.idx_sampled = io->zidx_sampled_out ? io->zidx_sampled_out + (size_t)t * L : nullptr; a.ldsm = (int64_t)H * L;
            a.z_dense = io->z_out ? io->z_out + (size_t)t * LK : nullptr; a.ldzd = (int64_t)H * LK;
            a.logits_out = io->prior_logits_out ? io->prior_logits_out + (size_t)t * LK : nullptr; a.ldlo = (int64_t)H * LK;
            a.B = B; a.L = L; a.log1m_eps = l1m; a.log_eps_over_k = lek;
            a.one_m_eps = 1.0f - d.unimix_eps; a.eps_over_k = d.unimix_eps / (float)d.K;
            const int64_t blocks = ((int64_t)B * L + 31) / 32;
            ProfScope prof(PC_SAMPLE, s, 12.0 * B * LK, (double)B * L * (256.0 + (a.z_dense ? 128.0 : 0.0) + 1.0));
            k_sample_rows<<<dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s>>>(a);
            DRM_LAUNCH_CHECK();
        } else {
            const int last = w->dyn.n_layers - 1;
            ProblemSpec p{ws + wl.d_pre[last & 1], M, w->dyn.ln_g[last], w->dyn.ln_b[last], w->dyn.head_w, M, w->dyn.head_b,
                          nullptr, 0, nullptr, 0, LK, M};
            p.Y = ws;  // unused by the sampling epilogue
            SampleEpilogue E{};
            E.expo = io->expo_z + (size_t)t * nb * LK; E.lde = LK;
            E.forced = io->forced_zidx ? io->forced_zidx + (size_t)t * L : nullptr; E.ldf = (int64_t)H * L;
            E.idx_used = io->zidx_out + (size_t)t * L; E.ldu = (int64_t)H * L;
            E.idx_sampled = io->zidx_sampled_out ? io->zidx_sampled_out + (size_t)t * L : nullptr; E.lds_ = (int64_t)H * L;
            E.z_dense = io->z_out ? io->z_out + (size_t)t * LK : nullptr; E.ldzd = (int64_t)H * LK;
            E.logits_out = io->prior_logits_out ? io->prior_logits_out + (size_t)t * LK : nullptr; E.ldlo = (int64_t)H * LK;
            E.log1m_eps = l1m; E.log_eps_over_k = lek;
            if (int rc = launch_sample(p, B, L, E, s)) return rc;
        }
    }
    }

    // ---- heads over all B*H imagined states ----
    const int rows = B * H;
    const drm_mlp* heads[3] = {&w->rew, &w->cont, &w->critic};
    const size_t tabs[3] = {pl.rew_zT, pl.cont_zT, pl.critic_zT};
    int depth_max = 0;
    for (int i = 0; i < 3; ++i) depth_max = heads[i]->n_layers > depth_max ? heads[i]->n_layers : depth_max;
    float* st_h[3][2];
    int parts_h[3][2] = {{0, 0}, {0, 0}, {0, 0}}, pw_h[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) st_h[i][j] = fused ? ws + wl.st_h[i][j] : nullptr;
    // many rows of a default-width model: the one-hot part of the first layers is added by a kernel of its own
    const bool table_add = big && !wide && L <= TABLE_ADD_MAX_L && rows >= BIG_MIN_ROWS && std::getenv("DRM_NO_TABLE_ADD") == nullptr;
    for (int dep = 0; dep <= depth_max; ++dep) {
        ProblemSpec p[3];
        int who[3];
        int np = 0;
        for (int i = 0; i < 3; ++i) {
            const drm_mlp& m = *heads[i];
            // a later matrix product normalises this output while loading it (the scalar finish kernel normalises by itself)
            const bool feeds_product = dep + 1 < m.n_layers || (i == 2 && dep + 1 == m.n_layers);
            if (dep == 0) {
                p[np] = ProblemSpec{io->h_out, D, nullptr, nullptr, m.w[0], S, m.b[0], table_add ? nullptr : pack + tabs[i], M,
                                    ws + wl.head[i][0], M, M, D};
            } else if (dep < m.n_layers) {
                p[np] = ProblemSpec{ws + wl.head[i][(dep - 1) & 1], M, m.ln_g[dep - 1], m.ln_b[dep - 1], m.w[dep], M, m.b[dep],
                                    nullptr, 0, ws + wl.head[i][dep & 1], M, M, M};
            } else if (dep == m.n_layers && i == 2) {  // critic bucket logits
                p[np] = ProblemSpec{ws + wl.head[2][(dep - 1) & 1], M, m.ln_g[dep - 1], m.ln_b[dep - 1], m.head_w, M, m.head_b,
                                    nullptr, 0, ws + wl.critic_logits, 256, d.NB, M};
            } else {
                continue;
            }
            if (dep > 0) {
                p[np].stats_in = st_h[i][(dep - 1) & 1]; p[np].parts_in = parts_h[i][(dep - 1) & 1];
                p[np].part_w_in = pw_h[i][(dep - 1) & 1];
            }
            if (dep < m.n_layers && feeds_product && !(dep == 0 && table_add)) p[np].stats_out = st_h[i][dep & 1];
            who[np++] = i;
        }
        if (np == 0) continue;
        if (fused) {
            if (int rc = launch_big(p, np, rows, io->zidx_out, L, L, d.K, s, prec)) return rc;
            for (int q = 0; q < np; ++q)
                if (p[q].stats_out) { parts_h[who[q]][dep & 1] = p[q].parts_out; pw_h[who[q]][dep & 1] = p[q].part_w_out; }
        } else {
            for (int q = 0; q < np; ++q) { p[q].stats_in = nullptr; p[q].parts_in = 0; p[q].stats_out = nullptr; }
            if (int rc = run_linear(p, np, rows, io->zidx_out, L, L, d.K, big, act, M, s, prec)) return rc;
        }
        if (dep == 0 && table_add) {
            // the z part of the first layers: one pass over the three outputs with the tables' column slabs held in LDS
            TableAddArgs a{};
            a.nj = np; a.zidx = io->zidx_out; a.ldz = L; a.rows = rows; a.L = L;
            for (int q = 0; q < np; ++q) {
                const int i = who[q];
                const bool want_stats = fused && (1 < heads[i]->n_layers || i == 2);
                a.j[q] = TableAddJob{pack + tabs[i], M, ws + wl.head[i][0], M, ws + wl.head[i][0], M,
                                     want_stats ? st_h[i][0] : nullptr, M, q * (M / 32)};
                if (want_stats) { parts_h[i][0] = M / 32; pw_h[i][0] = 32; }
            }
            const int slabs = np * (M / 32);
            int split = cu_count() / slabs > 0 ? cu_count() / slabs : 1;   // one workgroup per CU (128 KB of LDS each): a single round
            if (split > (rows + 127) / 128) split = (rows + 127) / 128;
            ProfScope prof(PC_SAMPLE, s, (double)rows * L * M * np, 4.0 * np * (2.0 * rows * M + (double)L * 32 * M) + (double)rows * L);
            k_table_add<<<dim3(slabs, split), dim3(1024), 0, s>>>(a);
            DRM_LAUNCH_CHECK();
        }
    }
    {
        ScalarHeadsArgs a{};
        a.nh = 2; a.rows = rows; a.M = M;
        for (int i = 0; i < 2; ++i) {
            const drm_mlp& m = *heads[i];
            const int last = m.n_layers - 1;
            a.h[i] = ScalarHead{ws + wl.head[i][last & 1], M, m.ln_g[last], m.ln_b[last], m.head_w, m.head_b,
                                i == 0 ? io->rewards_out : io->conts_out, nullptr, i == 0 ? 1 : 2};
        }
        const CriticValueArgs cv{ws + wl.critic_logits, 256, d.NB, w->buckets, H, io->values_out, io->v_last_out};
        ProfScope prof(PC_SCALAR_HEADS, s, 4.0 * rows * M + 4.0 * rows * d.NB,
                       4.0 * (2.0 * rows * M + 6.0 * M + 2.0 * rows) + 4.0 * ((double)rows * d.NB + d.NB + rows));
        k_scalar_heads<<<dim3((rows + 3) / 4, 2), dim3(256), 0, s>>>(a, cv);
        DRM_LAUNCH_CHECK();
    }
    return drm_lambda_return(io->rewards_out, io->conts_out, io->values_out, io->v_last_out, io->lam_returns_out, B, H,
                             io->gamma, io->lam, stream);
}

// Many trajectories are independent: from 2048 on the call runs as two chains of launches, one half of the trajectories
// each, on two streams of the library's own between a fork and a join on the caller's stream (inside a recording the two
// chains become two branches of the graph).  The launch boundaries, prologues and row kernels of one chain then sit under
// the matrix products of the other; every kernel and every result is what the single chain gives.
namespace drm {
constexpr int MAX_CHAINS = 8;
struct TwoChains { hipStream_t s[MAX_CHAINS]; hipEvent_t fork, join[MAX_CHAINS]; int dev; };
static thread_local TwoChains t_chains{{}, nullptr, {}, -1};
static int chain_count(int B) {
    const char* e = std::getenv("DRM_CHAINS");
    int n = e ? atoi(e) : 2;
    if (n < 2) n = 2;
    if (n > MAX_CHAINS) n = MAX_CHAINS;
    while (n > 2 && (B % (64 * n) != 0 || B / n < 512)) n /= 2;
    return n;
}
static bool two_chains_ok(int B) {
    return B >= two_chains_min_rows() && B % 128 == 0 && !prof_on() && std::getenv("DRM_NO_TWO_CHAINS") == nullptr;
}
}  // namespace drm

static int imagine_enqueue(const drm_imagine_weights* w, const drm_imagine_io* io, drm_stream_t stream) {
    if (!two_chains_ok(io->B)) return imagine_enqueue_part(w, io, stream, io->B);
    if (int rc = check_weights(*w)) return rc;
    const drm_dims& d = w->dims;
    const int nch = chain_count(io->B);
    const int B = io->B, H = io->H, half = B / nch, LK = d.L * d.K;
    DRM_REQUIRE(H >= 1 && io->workspace && aligned16(io->workspace), DRM_EINVAL, "bad io");
    const size_t part_bytes = (ws_layout(d, half, H).total * sizeof(float) + 255) & ~(size_t)255;
    DRM_REQUIRE(io->workspace_bytes >= nch * part_bytes, DRM_EINVAL, "workspace too small: need %zu bytes", nch * part_bytes);
    int dev = 0;
    DRM_HIP(hipGetDevice(&dev));
    TwoChains& c = t_chains;
    if (c.dev != dev) {
        for (int i = 0; i < MAX_CHAINS; ++i) {
            DRM_HIP(hipStreamCreateWithFlags(&c.s[i], hipStreamNonBlocking));
            DRM_HIP(hipEventCreateWithFlags(&c.join[i], hipEventDisableTiming));
        }
        DRM_HIP(hipEventCreateWithFlags(&c.fork, hipEventDisableTiming));
        c.dev = dev;
    }
    hipStream_t s = (hipStream_t)stream;
    DRM_HIP(hipEventRecord(c.fork, s));
    int rc = DRM_OK;
    for (int i = 0; i < nch && rc == DRM_OK; ++i) {
        DRM_HIP(hipStreamWaitEvent(c.s[i], c.fork, 0));
        const size_t b0 = (size_t)i * half;
        drm_imagine_io p = *io;
        p.B = half;
        p.h0 = io->h0 + b0 * d.D; p.zidx0 = io->zidx0 + b0 * d.L;
        if (io->expo_a) { p.expo_a = io->expo_a + b0 * d.A; p.expo_z = io->expo_z + b0 * LK; }
        if (io->forced_actions) p.forced_actions = io->forced_actions + b0 * H;
        if (io->forced_zidx) p.forced_zidx = io->forced_zidx + b0 * H * d.L;
        p.h_out = io->h_out + b0 * H * d.D; p.zidx_out = io->zidx_out + b0 * H * d.L;
        p.actions_out = io->actions_out + b0 * H;
        p.rewards_out = io->rewards_out + b0 * H; p.conts_out = io->conts_out + b0 * H; p.values_out = io->values_out + b0 * H;
        p.v_last_out = io->v_last_out + b0; p.lam_returns_out = io->lam_returns_out + b0 * H;
        if (io->z_out) p.z_out = io->z_out + b0 * H * LK;
        if (io->prior_logits_out) p.prior_logits_out = io->prior_logits_out + b0 * H * LK;
        if (io->actor_logp_out) p.actor_logp_out = io->actor_logp_out + b0 * H * d.A;
        if (io->actions_sampled_out) p.actions_sampled_out = io->actions_sampled_out + b0 * H;
        if (io->zidx_sampled_out) p.zidx_sampled_out = io->zidx_sampled_out + b0 * H * d.L;
        p.workspace = (char*)io->workspace + i * part_bytes; p.workspace_bytes = part_bytes;
        rc = imagine_enqueue_part(w, &p, (drm_stream_t)c.s[i], B, (int)b0);
        // joined even after a failure: a recording must not end with a dangling branch
        if (hipEventRecord(c.join[i], c.s[i]) == hipSuccess) (void)hipStreamWaitEvent(s, c.join[i], 0);
    }
    return rc;
}

// ---- optional replay of the whole block as one hipGraph -----------------------------------------
namespace drm {
struct GraphEntry { std::vector<uint8_t> key; hipGraphExec_t exec; };
static std::mutex g_graph_mu;
static std::vector<GraphEntry> g_graphs;
constexpr size_t MAX_GRAPHS = 16;
}  // namespace drm

extern "C" int drm_graph_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto& e : g_graphs) (void)hipGraphExecDestroy(e.exec);
    g_graphs.clear();
    return DRM_OK;
}

extern "C" int drm_imagine_block_f32(const drm_imagine_weights* w, const drm_imagine_io* io, drm_stream_t stream) {
    DRM_REQUIRE(w && io, DRM_EINVAL, "null pointer");
    if (!io->use_graph || prof_on() || sync_check_mode() != 0) return imagine_enqueue(w, io, stream);
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    DRM_HIP(hipGetDevice(&dev));
    std::vector<uint8_t> key(sizeof(*w) + sizeof(*io) + sizeof(int));
    memcpy(key.data(), w, sizeof(*w));
    memcpy(key.data() + sizeof(*w), io, sizeof(*io));
    memcpy(key.data() + sizeof(*w) + sizeof(*io), &dev, sizeof(int));
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto& e : g_graphs)
        if (e.key == key) { DRM_HIP(hipGraphLaunch(e.exec, s)); return DRM_OK; }
    // first call with these pointers and shapes: record the launches on a stream of the library's
    // own (the caller's may be the default stream, which cannot be captured), then replay
    static thread_local hipStream_t cap = nullptr;
    static thread_local int cap_dev = -1;
    if (cap == nullptr || cap_dev != dev) {
        DRM_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        cap_dev = dev;
    }
    DRM_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    const int rc = imagine_enqueue(w, io, (drm_stream_t)cap);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(cap, &graph);
    if (rc != DRM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    DRM_HIP(ce);
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    DRM_HIP(ie);
    if (g_graphs.size() >= MAX_GRAPHS) { (void)hipGraphExecDestroy(g_graphs.front().exec); g_graphs.erase(g_graphs.begin()); }
    g_graphs.push_back(GraphEntry{std::move(key), exec});
    DRM_HIP(hipGraphLaunch(exec, s));
    return DRM_OK;
}

// ---- the same block for callers that must not allocate ------------------------------------------
namespace drm {
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)DRM_IMAGINE_NOALLOC_MAX_WORKSPACE;
// Workspace of drm_imagine_block_noalloc_f32: part of the library's device image, so every GPU the library
// runs on holds its own copy at its own address from the moment the code object is loaded there.
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
constexpr int SCRATCH_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[SCRATCH_MAX_DEVICES] = {};   // device address of `scratch` per device ordinal

// Device address of the current device's copy of `scratch` (looked up once per device, with that device current:
// in host code the array's name is only a handle for this lookup, never an address a kernel may use).
static int scratch_of_current_device(void** out) {
    int dev = 0;
    DRM_HIP(hipGetDevice(&dev));
    DRM_REQUIRE(dev >= 0 && dev < SCRATCH_MAX_DEVICES, DRM_ENOTSUP, "device ordinal %d (at most %d devices)", dev, SCRATCH_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        DRM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        DRM_REQUIRE(p != nullptr && aligned16(p), DRM_EHIP, "no usable address for the static workspace on device %d", dev);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return DRM_OK;
}
}  // namespace drm

extern "C" int drm_imagine_block_noalloc_f32(const drm_imagine_weights* w, const drm_imagine_io* io, drm_stream_t stream) {
    DRM_REQUIRE(w && io, DRM_EINVAL, "null pointer");
    DRM_REQUIRE(io->use_graph == 0, DRM_EINVAL, "use_graph must be 0: instantiating a graph allocates");
    DRM_REQUIRE(!prof_on(), DRM_EINVAL, "per-kernel timing is enabled (drm_prof_enable): it creates events");
    if (int rc = check_dims(w->dims)) return rc;
    DRM_REQUIRE(io->B >= 1 && io->H >= 1 && (int64_t)io->B * io->H < (1 << 30), DRM_EINVAL, "B=%d H=%d", io->B, io->H);
    const size_t need = ws_layout(w->dims, io->B, io->H).total * sizeof(float);
    DRM_REQUIRE(need <= STATIC_SCRATCH_BYTES, DRM_ENOTSUP,
                "B=%d x H=%d at these widths needs a workspace of %zu bytes; the static workspace of this entry point holds "
                "%zu (use drm_imagine_block_f32 with a caller workspace)", io->B, io->H, need, STATIC_SCRATCH_BYTES);
    void* ws = nullptr;
    if (int rc = scratch_of_current_device(&ws)) return rc;
    drm_imagine_io local = *io;
    local.workspace = ws;
    local.workspace_bytes = STATIC_SCRATCH_BYTES;
    // one chain of launches on the caller's stream, whatever the number of trajectories: the side-by-side chains of
    // drm_imagine_block_f32 create streams and events on a thread's first call and need a workspace per chain
    return imagine_enqueue_part(w, &local, stream, local.B);
}

#ifdef DRM_STAMPS
extern "C" int drm_debug_set_stamp_buffer(void* p) {
    DRM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(drm::g_stamp_buf), &p, sizeof(p)));
    return DRM_OK;
}
#endif
