// Host-side launch helpers for the small-row fused linear kernels (shared by the imagination and
// observe drivers).
#pragma once
#include "linear16.hpp"
#include "linear_skinny.hpp"

namespace drm {

// ------------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------------
struct ProblemSpec {
    const float* X; int64_t ldx; const float* ln_g; const float* ln_b;
    const float* W; int64_t ldw; const float* bias; const float* tab; int64_t ldt;
    float* Y; int64_t ldy; int N, K;
    const float* add = nullptr; int64_t ldadd = 0;          // Y += add[m][n]
    const float* rowscale = nullptr; int64_t ldrs = 0;      // Y *= rowscale[m]
    // large-row fp32 launches only: per slice of the row (sum, sum of squared deviations from the slice's mean).
    // stats_in: those of the X rows, which LayerNorm (ln_g / ln_b) needs, parts_in slices of part_w_in columns;
    // stats_out: where to leave those of the Y rows (launch_big fills parts_out / part_w_out with the number of slices
    // per row it wrote and their width)
    const float* stats_in = nullptr; int parts_in = 0;
    float* stats_out = nullptr; int parts_out = 0;
    int part_w_in = 0, part_w_out = 0;
};

static inline int check_problem(const ProblemSpec& s) {
    DRM_REQUIRE(s.X && s.W && s.Y, DRM_EINVAL, "null pointer in linear problem");
    DRM_REQUIRE(s.K >= 32 && s.K % 32 == 0 && s.K <= 4096, DRM_ENOTSUP, "K=%d must be a multiple of 32 in [32, 4096]", s.K);
    DRM_REQUIRE(!s.ln_g || s.K <= 512, DRM_ENOTSUP, "LayerNorm prologue needs K <= 512 (got %d)", s.K);
    DRM_REQUIRE(s.N >= 1, DRM_EINVAL, "N=%d", s.N);
    DRM_REQUIRE(aligned16(s.X) && aligned16(s.W) && s.ldx % 4 == 0 && s.ldw % 4 == 0, DRM_EINVAL,
                "X/W must be 16-byte aligned with row strides a multiple of 4 floats");
    DRM_REQUIRE(!s.ln_g || (s.ln_b && aligned16(s.ln_g) && aligned16(s.ln_b)), DRM_EINVAL, "bad LayerNorm pointers");
    return DRM_OK;
}

constexpr int MID_MIN_ROWS = 96;

static inline int launch_store(const ProblemSpec* specs, int np, int M, const uint8_t* zidx, int64_t ldz, int latents,
                        int classes, hipStream_t stream) {
    DRM_REQUIRE(np >= 1 && np <= MAX_PROBLEMS && M >= 1, DRM_EINVAL, "np=%d M=%d", np, M);
    LinLaunch L{};
    int tiles = 0, kall = 0;
    for (int i = 0; i < np; ++i) kall = specs[i].K > kall ? specs[i].K : kall;
    // around a hundred rows and more: 32 x 32 output tiles per workgroup (one resident round, half the L2 reads)
    const bool mid = M >= MID_MIN_ROWS && kall <= 512;
    const int tw = mid ? 32 : 16;
    for (int i = 0; i < np; ++i) {
        if (int rc = check_problem(specs[i])) return rc;
        DRM_REQUIRE(!specs[i].tab || zidx, DRM_EINVAL, "gather table without indices");
        LinProblem& p = L.p[i];
        p.X = specs[i].X; p.ldx = specs[i].ldx; p.ln_g = specs[i].ln_g; p.ln_b = specs[i].ln_b;
        p.W = specs[i].W; p.ldw = specs[i].ldw; p.bias = specs[i].bias; p.tab = specs[i].tab; p.ldt = specs[i].ldt;
        p.Y = specs[i].Y; p.ldy = specs[i].ldy; p.N = specs[i].N; p.K = specs[i].K; p.tile_begin = tiles;
        p.add = specs[i].add; p.ldadd = specs[i].ldadd; p.rowscale = specs[i].rowscale; p.ldrs = specs[i].ldrs;
        tiles += (specs[i].N + tw - 1) / tw;
    }
    L.np = np; L.M = M; L.zidx = zidx; L.ldz = ldz; L.latents = latents; L.classes = classes;
    double fl = 0, by = 0;
    for (int i = 0; i < np; ++i) {
        const double n = specs[i].N, k = specs[i].K, m = M;
        fl += 2.0 * m * n * k + (specs[i].tab ? (double)latents * m * n : 0.0);
        by += 4.0 * (m * k + n * k + n + m * n) + (specs[i].tab ? 4.0 * latents * classes * n + (double)latents * m : 0.0);
    }
    DRM_REQUIRE(latents <= MAX_LATENTS, DRM_ENOTSUP, "latents=%d (max %d)", latents, MAX_LATENTS);
    int kmax = 0;
    for (int i = 0; i < np; ++i) kmax = specs[i].K > kmax ? specs[i].K : kmax;
    ProfScope prof(PC_LINEAR, stream, fl, by);
    if (mid) {   // 48- or 32-row workgroups, whichever pads the row count less
        if ((M + 47) / 48 * 48 <= (M + 31) / 32 * 32) k_linear16m<3, 2><<<dim3(tiles, (M + 47) / 48), dim3(256), 0, stream>>>(L);
        else k_linear16m<2, 2><<<dim3(tiles, (M + 31) / 32), dim3(256), 0, stream>>>(L);
        DRM_LAUNCH_CHECK();
        return DRM_OK;
    }
    const dim3 grid(tiles, (M + 15) / 16), block(256);
    switch (chunks_for(kmax)) {
        case 1: k_linear16<1, EPI_STORE, 1><<<grid, block, 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, BwdEpilogue{}); break;
        case 2: k_linear16<1, EPI_STORE, 2><<<grid, block, 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, BwdEpilogue{}); break;
        default: k_linear16<1, EPI_STORE, 4><<<grid, block, 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, BwdEpilogue{}); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

static inline int launch_gru(const ProblemSpec& spec, int M, const GruEpilogue& G, hipStream_t stream) {
    if (int rc = check_problem(spec)) return rc;
    DRM_REQUIRE(spec.N == 3 * G.D && G.D % 16 == 0, DRM_ENOTSUP, "GRU width %d", G.D);
    LinLaunch L{};
    LinProblem& p = L.p[0];
    p.X = spec.X; p.ldx = spec.ldx; p.W = spec.W; p.ldw = spec.ldw; p.N = spec.N; p.K = spec.K; p.tile_begin = 0;
    L.np = 1; L.M = M;
    ProfScope prof(PC_GRU, stream, 2.0 * M * spec.N * spec.K + 12.0 * M * G.D,
                   4.0 * ((double)M * spec.K + (double)spec.N * spec.K + spec.N + 5.0 * M * G.D + M * G.D));
    const dim3 grid(G.D / 16, (M + 15) / 16), block(256);
    switch (chunks_for(spec.K)) {
        case 1: k_linear16<3, EPI_GRU, 1><<<grid, block, 0, stream>>>(L, G, SampleEpilogue{}, BwdEpilogue{}); break;
        case 2: k_linear16<3, EPI_GRU, 2><<<grid, block, 0, stream>>>(L, G, SampleEpilogue{}, BwdEpilogue{}); break;
        default: k_linear16<3, EPI_GRU, 4><<<grid, block, 0, stream>>>(L, G, SampleEpilogue{}, BwdEpilogue{}); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

// `extra` (optional): a second GEMM with the plain store epilogue that shares the launch (its tiles are
// 32 columns wide); used for gh = W_hh h, which consumes the same freshly written h as the logits.
static inline int launch_sample(const ProblemSpec& spec, int M, int latents, const SampleEpilogue& S, hipStream_t stream,
                                const ProblemSpec* extra = nullptr) {
    if (int rc = check_problem(spec)) return rc;
    if (extra) { if (int rc = check_problem(*extra)) return rc; }
    DRM_REQUIRE(spec.N == latents * 32, DRM_EINVAL, "sampling head must have N = 32*latents");
    LinLaunch L{};
    LinProblem& p = L.p[0];
    p.X = spec.X; p.ldx = spec.ldx; p.ln_g = spec.ln_g; p.ln_b = spec.ln_b; p.W = spec.W; p.ldw = spec.ldw;
    p.bias = spec.bias; p.N = spec.N; p.K = spec.K; p.tile_begin = 0;
    L.np = 1; L.M = M;
    int tiles = latents, kmax = spec.K;
    double xfl = 0, xby = 0;
    if (extra) {
        LinProblem& e = L.p[1];
        e.X = extra->X; e.ldx = extra->ldx; e.ln_g = extra->ln_g; e.ln_b = extra->ln_b; e.W = extra->W; e.ldw = extra->ldw;
        e.bias = extra->bias; e.Y = extra->Y; e.ldy = extra->ldy; e.N = extra->N; e.K = extra->K; e.tile_begin = latents;
        L.np = 2;
        tiles += (extra->N + 31) / 32;
        kmax = extra->K > kmax ? extra->K : kmax;
        xfl = 2.0 * M * extra->N * extra->K;
        xby = 4.0 * ((double)M * extra->K + (double)extra->N * extra->K + extra->N + (double)M * extra->N);
    }
    ProfScope prof(PC_DYN_SAMPLE, stream, 2.0 * M * spec.N * spec.K + xfl,
                   4.0 * ((double)M * spec.K + (double)spec.N * spec.K + spec.N + (double)M * spec.N) + (double)M * latents + xby);
    const dim3 grid(tiles, (M + 15) / 16), block(256);
    switch (chunks_for(kmax)) {
        case 1: k_linear16<2, EPI_SAMPLE, 1><<<grid, block, 0, stream>>>(L, GruEpilogue{}, S, BwdEpilogue{}); break;
        case 2: k_linear16<2, EPI_SAMPLE, 2><<<grid, block, 0, stream>>>(L, GruEpilogue{}, S, BwdEpilogue{}); break;
        default: k_linear16<2, EPI_SAMPLE, 4><<<grid, block, 0, stream>>>(L, GruEpilogue{}, S, BwdEpilogue{}); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}


// d(h) GEMM of the reverse observe step with the GRU cell's elementwise backward as its epilogue
static inline int launch_grubwd(const ProblemSpec& spec, int M, const BwdEpilogue& E, hipStream_t stream) {
    if (int rc = check_problem(spec)) return rc;
    DRM_REQUIRE(spec.N == E.D && E.D % 16 == 0, DRM_ENOTSUP, "GRU width %d", E.D);
    LinLaunch L{};
    LinProblem& p = L.p[0];
    p.X = spec.X; p.ldx = spec.ldx; p.W = spec.W; p.ldw = spec.ldw; p.N = spec.N; p.K = spec.K; p.tile_begin = 0;
    L.np = 1; L.M = M;
    ProfScope prof(PC_GRU, stream, 2.0 * M * spec.N * spec.K + 30.0 * M * E.D,
                   4.0 * ((double)M * spec.K + (double)spec.N * spec.K + 14.0 * M * E.D));
    k_linear16<1, EPI_GRUBWD, 4><<<dim3(E.D / 16, (M + 15) / 16), dim3(256), 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, E);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

// d(z) GEMM of the reverse observe step with the straight-through sample backward as its epilogue
static inline int launch_stbwd(const ProblemSpec& spec, int M, int latents, const BwdEpilogue& E, hipStream_t stream) {
    if (int rc = check_problem(spec)) return rc;
    DRM_REQUIRE(spec.N == latents * 32, DRM_EINVAL, "N must be 32*latents");
    LinLaunch L{};
    LinProblem& p = L.p[0];
    p.X = spec.X; p.ldx = spec.ldx; p.W = spec.W; p.ldw = spec.ldw; p.N = spec.N; p.K = spec.K; p.tile_begin = 0;
    L.np = 1; L.M = M;
    ProfScope prof(PC_DYN_SAMPLE, stream, 2.0 * M * spec.N * spec.K + 12.0 * M * spec.N,
                   4.0 * ((double)M * spec.K + (double)spec.N * spec.K + 6.0 * M * spec.N));
    const dim3 grid(latents, (M + 15) / 16), block(256);
    switch (chunks_for(spec.K)) {
        case 1: k_linear16<2, EPI_STBWD, 1><<<grid, block, 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, E); break;
        case 2: k_linear16<2, EPI_STBWD, 2><<<grid, block, 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, E); break;
        default: k_linear16<2, EPI_STBWD, 4><<<grid, block, 0, stream>>>(L, GruEpilogue{}, SampleEpilogue{}, E); break;
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

// large-row MFMA GEMM (defined next to its kernel in imagine.hip)
int launch_big(ProblemSpec* specs, int np, int M, const uint8_t* zidx, int64_t ldz, int latents, int classes,
               hipStream_t stream, int precision = 0);
// bf16 images of the weights the large-row launches of the calling thread read, until ..._end(): `base` (16-byte
// aligned, `bytes` long) receives them on first use; planes = 3 serves precision 2, planes = 1 precision 1
void big_image_cache_begin(void* base, size_t bytes, int planes);
void big_image_cache_end();

// ---- few-row path ---------------------------------------------------------------------------------
// Problems of one launch share their rows and the (RW, CP) shape; LayerNorm is per problem.
constexpr int SKINNY_MAX_ROWS = 64;    // above this many rows the MFMA tiles are the better cut

static inline bool skinny_ok(const ProblemSpec& s) {
    return s.K <= 512 && s.tab == nullptr && s.rowscale == nullptr;
}

template <int RW>
static inline void skinny_dispatch(int cp, dim3 grid, hipStream_t stream, const SkinnyLaunch& L) {
    if (cp == 8) k_skinny<RW, 8><<<grid, dim3(256), 0, stream>>>(L);
    else if (cp == 4) k_skinny<RW, 4><<<grid, dim3(256), 0, stream>>>(L);
    else k_skinny<RW, 2><<<grid, dim3(256), 0, stream>>>(L);
}

// rows per workgroup (4 * RW): fewer rows mean less input to fetch and activate per workgroup, more
// columns (cold weight bytes) each; LayerNorm problems lean to fewer rows
static inline int skinny_rw(int M, bool has_ln) {
    if (M <= 4) return 1;
    if (M <= 8 || has_ln) return 2;
    return 4;
}

static inline int launch_skinny(const ProblemSpec* specs, int np, int M, hipStream_t stream) {
    DRM_REQUIRE(np >= 1 && np <= SKINNY_MAX_PROBLEMS && M >= 1, DRM_EINVAL, "np=%d M=%d", np, M);
    SkinnyLaunch L{};
    bool has_ln = false;
    int ntot = 0;
    double fl = 0, by = 0;
    for (int i = 0; i < np; ++i) {
        if (int rc = check_problem(specs[i])) return rc;
        DRM_REQUIRE(skinny_ok(specs[i]), DRM_ENOTSUP, "problem %d does not fit the few-row kernel", i);
        has_ln = has_ln || specs[i].ln_g != nullptr;
        ntot += specs[i].N;
        const double n = specs[i].N, k = specs[i].K, m = M;
        fl += 2.0 * m * n * k;
        by += 4.0 * (m * k + n * k + n + m * n);
    }
    const int rw = skinny_rw(M, has_ln);
    const int tiles_m = (M + 4 * rw - 1) / (4 * rw);
    int cp = 2;
    while (cp < 8 && (int64_t)tiles_m * ((ntot + cp - 1) / cp) > 384) cp *= 2;
    int wgs = 0;
    for (int i = 0; i < np; ++i) {
        SkinnyProblem& p = L.p[i];
        p.X = specs[i].X; p.ldx = specs[i].ldx; p.ln_g = specs[i].ln_g; p.ln_b = specs[i].ln_b;
        p.W = specs[i].W; p.ldw = specs[i].ldw; p.bias = specs[i].bias; p.Y = specs[i].Y; p.ldy = specs[i].ldy;
        p.N = specs[i].N; p.K = specs[i].K; p.wg_begin = wgs; p.add = specs[i].add; p.ldadd = specs[i].ldadd;
        wgs += (specs[i].N + cp - 1) / cp;
    }
    L.np = np; L.M = M;
    ProfScope prof(PC_LINEAR, stream, fl, by);
    const dim3 grid(wgs, tiles_m);
    if (rw == 1) skinny_dispatch<1>(cp, grid, stream, L);
    else if (rw == 2) skinny_dispatch<2>(cp, grid, stream, L);
    else skinny_dispatch<4>(cp, grid, stream, L);
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

// any K, at most 64 rows, plain store epilogue (+ bias): the memory-bound products of wide models (matrix-core kernel).  `images[i]`: a bf16
// image [N][K] of problem i's weights, or nullptr (then every problem reads fp32 weights).  `bf16`: the bf16 form of the
// product (operands rounded to bf16, fp32 accumulation).
static inline bool skinny_loop_ok(const ProblemSpec& s) {
    return s.ln_g == nullptr && s.tab == nullptr && s.rowscale == nullptr && s.add == nullptr && s.stats_out == nullptr &&
           s.K >= 32 && s.K % 4 == 0 && aligned16(s.X) && aligned16(s.W) && s.ldx % 4 == 0 && s.ldw % 4 == 0;
}
static inline int launch_skinny_loop(const ProblemSpec* specs, int np, int M, hipStream_t stream, bool bf16, const uint16_t* const* images) {
    DRM_REQUIRE(np >= 1 && np <= SKINNY_MAX_PROBLEMS && M >= 1 && M <= SKINNY_MAX_ROWS, DRM_EINVAL, "np=%d M=%d", np, M);
    SkinnyLoopLaunch L{};
    double fl = 0, by = 0;
    bool wh = bf16 && images != nullptr;
    for (int i = 0; i < np; ++i) {
        DRM_REQUIRE(specs[i].X && specs[i].W && specs[i].Y && skinny_loop_ok(specs[i]), DRM_EINVAL, "problem %d does not fit the K-looped few-row kernel", i);
        wh = wh && images[i] != nullptr;
        const double n = specs[i].N, k = specs[i].K, m = M;
        fl += 2.0 * m * n * k;
        by += 4.0 * (m * k + n + m * n) + (images && images[i] ? 2.0 : 4.0) * n * k;
    }
    // 16 RT rows x 16 columns per workgroup
    const int rt = M <= 16 ? 1 : M <= 32 ? 2 : 4;
    for (int i = 0; i < np && wh; ++i) wh = specs[i].K % 8 == 0;
    int wgs = 0;
    for (int i = 0; i < np; ++i) {
        SkinnyLoopProblem& p = L.p[i];
        p.X = specs[i].X; p.ldx = specs[i].ldx; p.W = specs[i].W; p.ldw = specs[i].ldw; p.Wh = wh ? images[i] : nullptr;
        p.bias = specs[i].bias; p.Y = specs[i].Y; p.ldy = specs[i].ldy; p.N = specs[i].N; p.K = specs[i].K; p.wg_begin = wgs;
        wgs += (specs[i].N + 15) / 16;
    }
    L.np = np; L.M = M;
    ProfScope prof(PC_LINEAR, stream, fl, by);
    const dim3 grid(wgs);
#define DRM_SKM(RTV)                                                                     \
    {                                                                                    \
        if (wh) k_skinny_mfma<RTV, 1, 2><<<grid, dim3(256), 0, stream>>>(L);             \
        else if (bf16) k_skinny_mfma<RTV, 1, 1><<<grid, dim3(256), 0, stream>>>(L);      \
        else k_skinny_mfma<RTV, 1, 0><<<grid, dim3(256), 0, stream>>>(L);                \
    }
    if (rt == 1) DRM_SKM(1) else if (rt == 2) DRM_SKM(2) else DRM_SKM(4)
#undef DRM_SKM
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

// K up to 1536 without LayerNorm; `G` != nullptr: one problem with the GRU-backward epilogue (N = G->D)
template <int RW, int EPI>
static inline void skinny_deep_dispatch(int cp, int kc, dim3 grid, hipStream_t stream, const SkinnyLaunch& L, const SkinnyGruBwd& G) {
#define DRM_DEEP(CPV, KCV) k_skinny_deep<RW, CPV, KCV, EPI><<<grid, dim3(256), 0, stream>>>(L, G)
    if (cp == 2) { if (kc == 1) DRM_DEEP(2, 1); else if (kc == 2) DRM_DEEP(2, 2); else DRM_DEEP(2, 3); }
    else { if (kc == 1) DRM_DEEP(4, 1); else if (kc == 2) DRM_DEEP(4, 2); else DRM_DEEP(4, 3); }
#undef DRM_DEEP
}

static inline bool skinny_deep_ok(const ProblemSpec& s) {
    return s.K <= 1536 && s.ln_g == nullptr && s.tab == nullptr && s.rowscale == nullptr;
}

static inline int launch_skinny_deep(const ProblemSpec* specs, int np, int M, hipStream_t stream, const SkinnyGruBwd* G = nullptr) {
    DRM_REQUIRE(np >= 1 && np <= SKINNY_MAX_PROBLEMS && M >= 1 && (G == nullptr || np == 1), DRM_EINVAL, "np=%d M=%d", np, M);
    SkinnyLaunch L{};
    int ntot = 0, kmax = 0;
    double fl = 0, by = 0;
    for (int i = 0; i < np; ++i) {
        if (int rc = check_problem(specs[i])) return rc;
        DRM_REQUIRE(skinny_deep_ok(specs[i]), DRM_ENOTSUP, "problem %d does not fit the few-row kernel", i);
        ntot += specs[i].N;
        kmax = specs[i].K > kmax ? specs[i].K : kmax;
        const double n = specs[i].N, k = specs[i].K, m = M;
        fl += 2.0 * m * n * k;
        by += 4.0 * (m * k + n * k + n + m * n);
    }
    DRM_REQUIRE(G == nullptr || specs[0].N == G->D, DRM_EINVAL, "GRU backward epilogue needs N = D");
    const int rw = skinny_rw(M, false);
    const int tiles_m = (M + 4 * rw - 1) / (4 * rw);
    const int cp = (int64_t)tiles_m * ((ntot + 1) / 2) > 384 ? 4 : 2;
    const int kc = (kmax + 511) / 512;
    int wgs = 0;
    for (int i = 0; i < np; ++i) {
        SkinnyProblem& p = L.p[i];
        p.X = specs[i].X; p.ldx = specs[i].ldx; p.W = specs[i].W; p.ldw = specs[i].ldw; p.bias = specs[i].bias;
        p.Y = specs[i].Y; p.ldy = specs[i].ldy; p.N = specs[i].N; p.K = specs[i].K; p.wg_begin = wgs;
        p.add = specs[i].add; p.ldadd = specs[i].ldadd;
        wgs += (specs[i].N + cp - 1) / cp;
    }
    L.np = np; L.M = M;
    ProfScope prof(G ? PC_GRU : PC_LINEAR, stream, fl + (G ? 30.0 * M * G->D : 0.0), by + (G ? 56.0 * M * G->D : 0.0));
    const dim3 grid(wgs, tiles_m);
    const SkinnyGruBwd g0{};
    if (G != nullptr) {
        if (rw == 1) skinny_deep_dispatch<1, 1>(cp, kc, grid, stream, L, *G);
        else if (rw == 2) skinny_deep_dispatch<2, 1>(cp, kc, grid, stream, L, *G);
        else skinny_deep_dispatch<4, 1>(cp, kc, grid, stream, L, *G);
    } else {
        if (rw == 1) skinny_deep_dispatch<1, 0>(cp, kc, grid, stream, L, g0);
        else if (rw == 2) skinny_deep_dispatch<2, 0>(cp, kc, grid, stream, L, g0);
        else skinny_deep_dispatch<4, 0>(cp, kc, grid, stream, L, g0);
    }
    DRM_LAUNCH_CHECK();
    return DRM_OK;
}

}  // namespace drm
