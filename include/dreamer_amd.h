/*
 * dreamer_amd.h — C ABI of libdreamer_amd.so: gfx950 (MI355X) HIP kernels for the Dreamer
 * hot path (RSSM imagine/observe steps, straight-through categorical sampling, lambda-return
 * scan, replay sequence gather).
 *
 * The reference (ProfessorNova/Dreamer) has no FFI layer: its boundary is the Python
 * class/method surface of lib/world_model.py, lib/actor.py, lib/critic.py,
 * lib/replay_buffer.py and lib/trainer.py.  Each entry point below names the reference lines
 * whose arithmetic it replaces; the Python package `dreamer_amd` calls these through ctypes
 * from inside methods with the reference's names and signatures.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise;
 *   - all tensors are dense row-major fp32 unless said otherwise; indices are int64 or uint8
 *     as stated; "rows" always means independent trajectories (batch rows);
 *   - every call enqueues on the given hipStream_t (passed as void*) and returns without
 *     synchronising; nothing is allocated or freed: scratch comes in as a workspace whose size
 *     the matching *_bytes() query returns;
 *   - return value: 0 on success, a negative DRM_E* code otherwise (never throws);
 *     drm_last_error() returns a thread-local message for the last failure;
 *   - noise enters as explicit Exp(1) tensors ("expo"), in the order the reference consumes its
 *     generator (F.gumbel_softmax: g = -log(E); torch.multinomial: argmax(p / E)).
 */
#ifndef DREAMER_AMD_H
#define DREAMER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRM_OK 0
#define DRM_EINVAL (-22)   /* bad shape / null pointer / misaligned pointer / workspace too small */
#define DRM_ENOTSUP (-95)  /* dimension outside what the kernels are built for */
#define DRM_EHIP (-5)      /* a HIP runtime call failed */
#define DRM_EARCH (-19)    /* the current device is not gfx950 */

#define DRM_MAX_MLP_LAYERS 4

typedef void* drm_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------ */
int drm_version(void);
const char* drm_last_error(void);
/* 0 if the current HIP device is gfx950, DRM_EARCH otherwise. */
int drm_arch_ok(void);

/* ---- per-kernel timing (measurement only; no counterpart in the reference) -------------------
 * While enabled, every kernel launch made by this library is bracketed by two HIP events on
 * the stream it is launched on, and its algorithmic work (flops; bytes with every operand read
 * once and every result written once) is recorded under its kernel class.  drm_prof_enable(1)
 * clears earlier records; drm_prof_summary() waits for the recorded events. */
#define DRM_PROF_CLASSES 12
int drm_prof_enable(int enable);
const char* drm_prof_class_name(int cls);
int drm_prof_summary(int cls, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* ---- lambda-return scan: lib/trainer.py:217-224 --------------------------------------------
 * ret[b,t] = R[b,t] + gamma*(1 - C[b,t])*((1-lam)*V[b,t] + lam*ret[b,t+1]),  ret[b,H] = V_last[b].
 * R, C, V, out: (B,H) with t contiguous; V_last: (B).  gamma and lam are the Python floats of
 * lib/config.py:36-37; they are rounded to fp32 the way torch rounds Python scalars that meet an
 * fp32 tensor ((float)gamma, (float)(1.0 - lam), (float)lam), every product and sum is rounded
 * separately (no fused multiply-add), so the result is bit-identical to the eager fp32 loop. */
int drm_lambda_return(const float* R, const float* C, const float* V, const float* V_last,
                      float* out, int64_t B, int32_t H, double gamma, double lam, drm_stream_t stream);

/* ---- replay sequence gather: lib/replay_buffer.py:64-78 (+ lib/trainer.py:163) --------------
 * The ring storage lives in device memory.  For b < B, t < T: src = (starts[b] + t) % capacity;
 * obs_out[b,t,:] = obs[src,:] (frame_bytes each), likewise actions (int64), rewards (f32),
 * continues (1 byte bool).  obs_f32_out, when not NULL, also receives float(obs)/255.0f
 * (lib/trainer.py:163).  frame_bytes must be a multiple of 16 and all obs pointers 16-byte
 * aligned.  starts is a device pointer to B int64 values, each in [0, capacity). */
typedef struct drm_ring {
    const uint8_t* observations; /* (capacity, frame_bytes) */
    const int64_t* actions;      /* (capacity) */
    const float* rewards;        /* (capacity) */
    const uint8_t* continues;    /* (capacity) bool bytes */
    int64_t capacity;
    int64_t frame_bytes;
} drm_ring;
int drm_seq_gather(const drm_ring* ring, const int64_t* starts, int32_t B, int32_t T,
                   uint8_t* obs_out, int64_t* actions_out, float* rewards_out, uint8_t* continues_out,
                   float* obs_f32_out, drm_stream_t stream);

/* ---- straight-through categorical sampling: lib/world_model.py:459-461, lib/utils.py:20-31 --
 * rows = B*L independent categoricals over K = 32 classes.
 *   lp = log_unimix(logits, eps)   (eps < 0: lp = logits, the form lib/world_model.py:59,:412 use)
 *   y  = softmax(lp - log(expo));  idx = argmax (first max on ties);  z = one_hot(idx).
 * Any of z_out (rows,K) f32, idx_out (rows) u8, y_out (rows,K) f32 may be NULL. */
int drm_sample_categorical_st(const float* logits, const float* expo, int64_t rows, int32_t K,
                              float unimix_eps, float* z_out, uint8_t* idx_out, float* y_out,
                              drm_stream_t stream);

/* argmax over the last dim of a dense (rows,K) one-hot tensor -> uint8 indices. */
int drm_onehot_to_index(const float* z, int64_t rows, int32_t K, uint8_t* idx_out, drm_stream_t stream);

/* KL terms of the world-model loss (lib/world_model.py:463-475, :501-528) over `rows` x L categoricals of
 * K = 32 classes: row value = sum_l max(KL(unimix(post_l) || unimix(prior_l)), free_nats). The dynamics
 * and the representation loss share the value (dyn_out, rep_out: `rows` floats each, both written) and
 * differ in the gradient: backward takes the upstream gradient of each use (g_dyn flows to the prior
 * logits only, g_rep to the posterior logits only; either may be NULL) and writes d_post, d_prior. */
int drm_kl_pair_forward(const float* post, const float* prior, int64_t rows, int32_t L, int32_t K, float unimix_eps,
                        float free_nats, float* dyn_out, float* rep_out, drm_stream_t stream);
int drm_kl_pair_backward(const float* post, const float* prior, int64_t rows, int32_t L, int32_t K, float unimix_eps,
                         float free_nats, const float* g_dyn, const float* g_rep, float* d_post, float* d_prior,
                         drm_stream_t stream);

/* Image term of the prediction loss (lib/world_model.py:483-487): out[r] = 0.5 * mean_e (x_hat[r,e] - x[r,e])^2
 * over frames of E floats (E % 4 == 0); backward: d_x_hat[r,e] = g[r] * (x_hat - x) / E. */
int drm_mse_rows_forward(const float* x_hat, const float* x, int64_t rows, int32_t E, float* out, drm_stream_t stream);
int drm_mse_rows_backward(const float* x_hat, const float* x, int64_t rows, int32_t E, const float* g, float* d_x_hat,
                          drm_stream_t stream);

/* Critic loss rows (lib/critic.py:140-161 with the two-hot target of lib/critic.py:103-129): for each of `rows`
 * states, row_loss = -sum_k y_k log_softmax(logits)_k + reg * sum_k kl_div(log_softmax(logits), softmax(slow_logits))_k,
 * y = two-hot encoding of clamp(symlog(returns), bucket_min, bucket_max) over NB <= 256 equally spaced buckets.
 * logits, slow_logits: (rows, NB); returns, row_loss, g_rows: (rows).  reg <= 0: no regulariser (slow_logits unused).
 * backward: d_logits = g_rows * ((softmax - y) + reg * (softmax - softmax(slow_logits))); no gradient to the others. */
int drm_critic_loss_forward(const float* logits, const float* slow_logits, const float* returns, int64_t rows, int32_t NB,
                            double bucket_min, double bucket_max, double reg, float* row_loss, drm_stream_t stream);
int drm_critic_loss_backward(const float* logits, const float* slow_logits, const float* returns, const float* g_rows,
                             int64_t rows, int32_t NB, double bucket_min, double bucket_max, double reg, float* d_logits,
                             drm_stream_t stream);

/* Critic.update_slow (lib/critic.py:131-138) over `count` tensors in one launch (per 32 tensors):
 * dst[i] <- dst[i] * decay, then dst[i] <- dst[i] + src[i] * (1 - decay), each step rounded to fp32.
 * src, dst, numel are HOST arrays of `count` device pointers / element counts. */
int drm_ema_multi(int32_t count, const float* const* src, float* const* dst, const int64_t* numel, double decay,
                  drm_stream_t stream);

/* EMAPercentileScale.update_get_S (lib/actor.py:16-33) in one launch: the q_lo / q_hi quantiles of x (n <= 16384
 * values, torch.quantile's linear interpolation), hi = max(hi, lo + 1e-8), when `training` the EMA update of the
 * device scalars p_lo, p_hi (p <- p * decay + (1 - decay) * q), and scale_out[0] = max(p_hi - p_lo, limit). */
int drm_percentile_scale(const float* x, int32_t n, double q_lo, double q_hi, double decay, double limit, int32_t training,
                         float* p_lo, float* p_hi, float* scale_out, drm_stream_t stream);

/* nn.ConvTranspose2d(Cin, Cout <= 4, kernel 4, stride 2, padding 1) + bias, forward: the last decoder stage
 * (lib/world_model.py:298-302).  x (N, Cin, H, W), weight (Cin, Cout, 4, 4) (torch's layout), bias (Cout) or
 * NULL, y (N, Cout, 2H, 2W); all contiguous fp32. */
int drm_convt_k4s2_forward(const float* x, const float* weight, const float* bias, float* y, int64_t N, int32_t Cin,
                           int32_t Cout, int32_t H, int32_t W, drm_stream_t stream);
/* Input gradient of that layer (what autograd's backward of lib/world_model.py:298-302 forms for the stage before):
 * grad_y (N, Cout, 2H, 2W) contiguous -> grad_x (N, Cin, H, W), contiguous or (channels_last != 0, Cin % 4 == 0) in
 * channels-last memory. */
int drm_convt_k4s2_input_grad(const float* grad_y, const float* weight, float* grad_x, int64_t N, int32_t Cin, int32_t Cout,
                              int32_t H, int32_t W, int32_t channels_last, drm_stream_t stream);

/* The image-side ends of the conv towers: a kernel-4 stride-2 padding-1 layer between an image `big` (N, Cimg <= 4, 2H, 2W;
 * contiguous NCHW) and a feature map `small` (N, H, W, Cs) in channels-last memory (Cs a multiple of 32), with `weight`
 * (Cs, Cimg, 4, 4) -- the layout of both nn.Conv2d(Cimg, Cs).weight and nn.ConvTranspose2d(Cs, Cimg).weight.
 *   drm_conv_img_down : small = conv2d(big, weight), no bias      (first encoder layer forward, lib/world_model.py:94-97,
 *                       :131; its bias is added by the LayerNorm kernel that follows)
 *   drm_conv_img_up   : big = conv_transpose2d(small, weight) + bias (bias (Cimg) or NULL; last decoder layer forward,
 *                       lib/world_model.py:298-302, :318)
 *   drm_conv_img_wgrad: grad_weight[cs][cimg][ky][kx] = sum_{n,i,j} small[n,i,j,cs] * big[n,cimg,2i-1+ky,2j-1+kx] -- the
 *                       weight gradient autograd forms for either layer (small = the conv's output gradient and big its
 *                       input, or small = the transposed conv's input and big its output gradient).  `workspace`: device
 *                       memory of drm_conv_img_wgrad_workspace_bytes(...) bytes (partial sums, added in a fixed order). */
int drm_conv_img_down(const float* big, const float* weight, float* small_out, int64_t N, int32_t Cs, int32_t Cimg, int32_t H,
                      int32_t W, drm_stream_t stream);
int drm_conv_img_up(const float* small_in, const float* weight, const float* bias, float* big_out, int64_t N, int32_t Cs,
                    int32_t Cimg, int32_t H, int32_t W, drm_stream_t stream);
size_t drm_conv_img_wgrad_workspace_bytes(int64_t N, int32_t Cs, int32_t Cimg, int32_t H, int32_t W);
int drm_conv_img_wgrad(const float* small_in, const float* big, float* grad_weight, void* workspace, size_t workspace_bytes,
                       int64_t N, int32_t Cs, int32_t Cimg, int32_t H, int32_t W, drm_stream_t stream);

/* ---- stride-2 4x4 convolutions of the towers as implicit matrix products, channels-last activations -------------
 * nn.Conv2d(cb -> cs, kernel 4, stride 2, padding 1) (lib/world_model.py:94-108) and nn.ConvTranspose2d(cs -> cb, kernel 4,
 * stride 2, padding 1) (lib/world_model.py:290-302), without bias (the bias is applied by the fused LayerNorm + SiLU that
 * follows).  `big` is (NI, 2 Hs, 2 Ws, cb), `small` is (NI, Hs, Ws, cs), both dense channels-last; `kernel` is
 * [cs][4][4][cb]: the weight of either module, (cs, cb, 4, 4) in torch's shape, held in channels-last memory.
 *   down : small = conv2d(big)                  conv forward; the input gradient of the transposed convolution
 *   up   : big = conv_transpose2d(small)        transposed-conv forward; the input gradient of the convolution
 *   wgrad: kernel gradient from one tensor of each resolution (weight gradient of either module)
 * cs, cb multiples of 4; down needs cb % 32 == 0, up needs cs % 32 == 0.  wgrad sums over all NI Hs Ws pixels: the sum
 * is split over workgroups into `workspace` (drm_conv4s2_wgrad_workspace_bytes) and added up in a fixed order. */
int drm_conv4s2_down(const float* big, const float* kernel, float* small_out, int64_t NI, int32_t Hs, int32_t Ws, int32_t cs,
                     int32_t cb, drm_stream_t stream);
int drm_conv4s2_up(const float* small, const float* kernel, float* big_out, int64_t NI, int32_t Hs, int32_t Ws, int32_t cs,
                   int32_t cb, drm_stream_t stream);
size_t drm_conv4s2_wgrad_workspace_bytes(int64_t NI, int32_t Hs, int32_t Ws, int32_t cs, int32_t cb);
int drm_conv4s2_wgrad(const float* small, const float* big, float* kernel_grad_out, int64_t NI, int32_t Hs, int32_t Ws,
                      int32_t cs, int32_t cb, void* workspace, size_t workspace_bytes, drm_stream_t stream);
/* The same three products with a precision argument: 0 = fp32 results (what the calls above give: both operands split
 * exactly into three bf16 parts on the bf16 matrix pipe, or v_mfma_f32_32x32x2_f32 under DRM_NO_SPLIT_PRODUCTS=1);
 * 1 = operands rounded to bf16, fp32 accumulation (the towers of a model whose update runs with bf16 matrix inputs). */
int drm_conv4s2_down_prec(const float* big, const float* kernel, float* small_out, int64_t NI, int32_t Hs, int32_t Ws,
                          int32_t cs, int32_t cb, int32_t precision, drm_stream_t stream);
int drm_conv4s2_up_prec(const float* small, const float* kernel, float* big_out, int64_t NI, int32_t Hs, int32_t Ws,
                        int32_t cs, int32_t cb, int32_t precision, drm_stream_t stream);
int drm_conv4s2_wgrad_prec(const float* small, const float* big, float* kernel_grad_out, int64_t NI, int32_t Hs, int32_t Ws,
                           int32_t cs, int32_t cb, void* workspace, size_t workspace_bytes, int32_t precision,
                           drm_stream_t stream);

/* Optimiser step on flat buffers: global-norm clip (torch.nn.utils.clip_grad_norm_, coefficient
 * min(1, max_norm / (norm + 1e-6)); max_norm <= 0: no clip) followed by Adam without weight decay, the
 * arithmetic of torch's fused Adam (lib/trainer.py:95-103 creates the optimisers, :170-173, :231-235,
 * :244-247 clip and step them).  `params`, `grads`, `exp_avg`, `exp_avg_sq`: n floats each, 16-byte aligned.
 * `step`: one device float holding the number of steps taken so far (incremented by the call).
 * `workspace`: drm_adam_workspace_bytes() bytes.  `norm_out` (optional): receives the gradient norm.
 * Gradients are left unscaled.  Two launches, no host synchronisation: recordable into a graph. */
size_t drm_adam_workspace_bytes(void);
int drm_adam_flat_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                       double beta1, double beta2, double eps, double max_norm, float* step, float* workspace,
                       float* norm_out, drm_stream_t stream);

/* Two device-to-device copies in one launch: staging of the start state (`start.h`, z indices of
 * lib/trainer.py:178) into the fixed buffers a graph-replayed drm_imagine_block_f32 reads. Regions must
 * not overlap. */
int drm_copy_pair(const void* a_src, void* a_dst, size_t a_bytes, const void* b_src, void* b_dst, size_t b_bytes,
                  drm_stream_t stream);
/* The same, and `*rng_state = rng_counter` from the same launch (the call counter a graph-replayed block with in-kernel
 * noise reads, see drm_imagine_io). */
int drm_copy_pair_rng(const void* a_src, void* a_dst, size_t a_bytes, const void* b_src, void* b_dst, size_t b_bytes,
                      uint64_t* rng_state, uint64_t rng_counter, drm_stream_t stream);

/* ---- fused linear block (building block of every step below) --------------------------------
 * Y[m, n] = bias[n] + sum_k f(X)[m,k] * W[n,k] + sum_l table[(l*classes + zidx[m,l]), n]
 *   f = identity, or SiLU(LayerNorm(X row; ln_gamma, ln_beta, eps 1e-5)) when ln_gamma != NULL
 *   (the Linear -> LayerNorm -> SiLU blocks of lib/world_model.py:153-160 etc., with the
 *   normalisation applied by the consumer of the raw pre-activations);
 *   the table term (optional) is the one-hot part of a Linear over cat[h, z_flat]
 *   (lib/world_model.py:213-215) written as a row gather.
 * W is (N,K) row-major with row stride ldw (torch Linear layout).  K % 32 == 0, K <= 4096; K <= 512 with the
 * LayerNorm prologue.  Without prologue and table, at most 64 rows over K > 512 take any K that is a multiple of 4
 * (a memory-bound kernel on 16-row matrix-core tiles: the products of a wide model at the default batch). */
typedef struct drm_linear_args {
    const float* X; int64_t ldx;
    const float* ln_gamma; const float* ln_beta;
    const float* W; int64_t ldw;
    const float* bias;
    const float* table; int64_t ldt;   /* (latents*classes, ldt) or NULL */
    const uint8_t* zidx; int64_t ldz;  /* (M, latents) */
    int32_t latents, classes;
    float* Y; int64_t ldy;
    int32_t M, N, K;
} drm_linear_args;
int drm_fused_linear_f32(const drm_linear_args* args, drm_stream_t stream);
/* The same product with bf16 matrix inputs and fp32 accumulation (operands stay fp32 in memory and are
 * rounded on load; v_mfma_f32_32x32x16_bf16).  No LayerNorm prologue (ln_gamma must be NULL); any M.
 * Results agree with the fp32 form to about 1e-2 relative (the bf16 tolerance of the rollout). */
int drm_fused_linear_bf16(const drm_linear_args* args, drm_stream_t stream);
/* The same product at fp32 accuracy on the bf16 matrix pipe: every fp32 operand is split exactly into three
 * bf16 parts on load and the six leading partial products are accumulated in fp32 (the dropped ones are
 * below the rounding of an fp32 product).  Results agree with drm_fused_linear_f32 to fp32 rounding
 * (same nn.Linear semantics, lib/world_model.py:153-160).  No LayerNorm prologue; any M. */
int drm_fused_linear_split(const drm_linear_args* args, drm_stream_t stream);

/* ---- general fp32 product for the batched (B*T)-row Linear layers of the world-model update and their gradients
 * (nn.Linear forward, lib/world_model.py:153-160, :203-217, :283-288 etc., and what autograd computes for it) ----
 *   C (M, N) = op(A) . op(B) (+ bias[n]),  fp32 MFMA, fp32 accumulation
 *   trans_a = 0: A is (M, K) with k contiguous;   trans_a = 1: A is (K, M) with m contiguous
 *   trans_b = 0: B is (N, K) with k contiguous (torch Linear weight);   trans_b = 1: B is (K, N) with n contiguous
 * forward Y = X W^T: (0, 0);  input gradient dX = dY W: (0, 1);  weight gradient dW = dY^T X: (1, 1).
 * lda, ldb multiples of 4, A and B 16-byte aligned, K % 4 == 0 for a k-contiguous operand.  Small outputs split K
 * over workgroups into `workspace` (drm_gemm_workspace_bytes, may be 0) and are summed in a fixed order. */
size_t drm_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K);
int drm_gemm_f32(int32_t trans_a, int32_t trans_b, int32_t M, int32_t N, int32_t K, const float* A, int64_t lda,
                 const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc, void* workspace,
                 size_t workspace_bytes, drm_stream_t stream);

/* The two gradient products of a large nn.Linear (lib/world_model.py:153-160, :283-288: what autograd computes for it)
 * whose operands do not have the reduction index contiguous in memory, on the bf16 matrix pipe with both operands split
 * exactly into three bf16 parts (fp32 results, as drm_fused_linear_split), without transposed copies of the operands:
 *   trans_a = 1:  C (M, N) = A^T B,  A (K, M), B (K, N)   -- weight gradient dW = dY^T X
 *   trans_a = 0:  C (M, N) = A B,    A (M, K), B (K, N)   -- input gradient dX = dY W     (K % 32 == 0)
 * N, lda, ldb, ldc multiples of 4 (M too for trans_a = 1), 16-byte aligned.  With few output tiles the reduction of the
 * trans_a = 1 form is split over workgroups into `workspace` (drm_gemm_split_workspace_bytes, may be 0; then ldc = N)
 * and summed in a fixed order.  bf16_inputs = 1: the operands rounded to bf16 instead (one matrix instruction per product
 * where the exact form has six; fp32 accumulation), the form the update side of a model run with bf16 matrix inputs takes. */
size_t drm_gemm_split_workspace_bytes(int32_t trans_a, int32_t M, int32_t N, int32_t K);
int drm_gemm_split_f32(int32_t trans_a, int32_t bf16_inputs, int32_t M, int32_t N, int32_t K, const float* A, int64_t lda,
                       const float* B, int64_t ldb, float* C, int64_t ldc, void* workspace, size_t workspace_bytes,
                       drm_stream_t stream);

/* Column sums of a row-major fp32 matrix, out[c] = sum_r x[r, c], rows added in a fixed order: the bias gradient autograd
 * forms for nn.Linear (lib/world_model.py:153-160 etc.: grad_bias = grad_output.sum(0)) and the sum over the slices of
 * the LayerNorm parameter-gradient partials (drm_ln_silu_backward).  cols and ldx multiples of 4, 16-byte aligned. */
int drm_colsum_f32(const float* x, int64_t ldx, int32_t rows, int32_t cols, float* out, drm_stream_t stream);

/* ---- imagination block: lib/trainer.py:176-224 ----------------------------------------------
 * actor -> sample action -> WorldModel.step(prior) for H steps, then reward / continue heads,
 * critic value and the lambda-return, for B independent trajectories.  The decoder call of
 * lib/world_model.py:447 is not executed (its output is discarded by lib/trainer.py:192 and it
 * consumes no random numbers). */
typedef struct drm_mlp {
    int32_t n_layers;  /* Linear+LayerNorm+SiLU blocks */
    int32_t in_dim, hidden, out_dim;
    const float* w[DRM_MAX_MLP_LAYERS];    /* (hidden, in) torch layout */
    const float* b[DRM_MAX_MLP_LAYERS];
    const float* ln_g[DRM_MAX_MLP_LAYERS];
    const float* ln_b[DRM_MAX_MLP_LAYERS];
    const float* head_w;  /* (out_dim, hidden) */
    const float* head_b;
} drm_mlp;

typedef struct drm_dims {
    int32_t D;   /* recurrent width (hidden_size): multiple of 32, <= 4096 (the observe scan too: beyond 512 its products run on the large-row kernel) */
    int32_t L;   /* latents */
    int32_t K;   /* classes per latent: must be 32 */
    int32_t M;   /* mlp_hidden_units: multiple of 32, <= 4096 */
    int32_t A;   /* actions (<= 32) */
    int32_t E;   /* action embedding width (64) */
    int32_t NB;  /* critic buckets (<= 256) */
    float unimix_eps;
} drm_dims;

typedef struct drm_imagine_weights {
    drm_dims dims;
    const float* a_emb;   /* seq.a_emb.weight (A,E) */
    const float* lin_w;   /* seq.linear.weight (D, L*K+E) */
    const float* lin_b;
    const float* gru_w_ih; /* (3D, D), gates r,z,n */
    const float* gru_w_hh;
    const float* gru_b_ih;
    const float* gru_b_hh;
    drm_mlp dyn;    /* D -> M x3 -> L*K */
    drm_mlp rew;    /* D+L*K -> M x3 -> 1 */
    drm_mlp cont;   /* D+L*K -> M x3 -> 1 */
    drm_mlp actor;  /* D+L*K -> M xn -> A */
    drm_mlp critic; /* D+L*K -> M xn -> NB */
    const float* buckets; /* (NB) */
    const float* pack;    /* written by drm_pack_imagine */
    size_t pack_bytes;
} drm_imagine_weights;

/* Bytes of the packed-table buffer for these dims (one-hot gather tables: the z-columns of the
 * first Linear of actor/critic/rew/cont and of seq.linear, transposed, plus the action table
 * seq.linear[:, L*K:] @ a_emb^T).  Must be re-packed after every optimiser step. */
size_t drm_pack_imagine_bytes(const drm_dims* dims);
int drm_pack_imagine(const drm_imagine_weights* w, float* pack_out, size_t pack_bytes, drm_stream_t stream);

typedef struct drm_imagine_io {
    int32_t B, H;
    const float* h0;               /* (B,D) start recurrent state */
    const uint8_t* zidx0;          /* (B,L) start latent class indices */
    const float* expo_a;           /* (H,B,A) Exp(1) for the action draw of each step, or NULL (then expo_z too): the */
    const float* expo_z;           /* (H,B,L,K) Exp(1) for the latent draw of each step    draws use rng_* below */
    const int64_t* forced_actions; /* (B,H) or NULL: use these instead of the sampled actions */
    const uint8_t* forced_zidx;    /* (B,H,L) or NULL: use these instead of the sampled latents */
    float* h_out;                  /* (B,H,D) */
    uint8_t* zidx_out;             /* (B,H,L) latents fed forward (forced ones if given) */
    uint8_t* zidx_sampled_out;     /* (B,H,L) or NULL: what the sampler drew */
    float* z_out;                  /* (B,H,L,K) dense one-hot or NULL */
    int64_t* actions_out;          /* (B,H) actions fed forward */
    int64_t* actions_sampled_out;  /* (B,H) or NULL */
    float* actor_logp_out;         /* (B,H,A) normalised mixed log-probs or NULL */
    float* prior_logits_out;       /* (B,H,L,K) or NULL */
    float* rewards_out;            /* (B,H) symexp(r_hat) */
    float* conts_out;              /* (B,H) sigmoid(c_hat) */
    float* values_out;             /* (B,H) critic value */
    float* v_last_out;             /* (B) */
    float* lam_returns_out;        /* (B,H) */
    double gamma, lam;
    void* workspace; size_t workspace_bytes;
    int32_t use_graph;             /* != 0: record the launches of this call into a hipGraph the first
                                      time this exact (weights, io) argument pair is seen and replay the
                                      graph afterwards (callers keep every buffer at a fixed address) */
    int32_t precision;             /* 0: fp32 everywhere.  1: the matrix products of the large-row path
                                      (B >= 256 or widths > 512) take bf16 inputs with fp32 accumulation;
                                      LayerNorm, gates, sampling and the lambda-return stay fp32.
                                      2: those products at fp32 accuracy from an exact three-way bf16 split
                                      of both operands (see drm_fused_linear_split); at widths up to 512 the
                                      faster of the two fp32-accurate forms from 256 trajectories on (the Python
                                      engine's default there) */
    /* Noise drawn inside the kernels (expo_a == expo_z == NULL; lib/world_model.py:459-461 and lib/trainer.py:190-191
     * draw it through torch's generator): Exp(1) values from Philox-4x32-10 keyed by rng_seed and addressed by
     * (call counter, kind of draw, step, trajectory, element).  The call counter is read from the device word
     * *rng_state when rng_state != NULL (a replayed graph then draws fresh numbers whenever the caller has stored a new
     * counter there, e.g. with drm_copy_pair_rng), else it is rng_counter.  Same counter and seed: same draws.  Built
     * into the step-wise launch sequence (B <= 64 at widths <= 512, or B >= 256 at widths <= 512 with precision 0 / 2);
     * other shapes return DRM_ENOTSUP without noise tensors. */
    const uint64_t* rng_state;
    uint64_t rng_seed, rng_counter;
} drm_imagine_io;

size_t drm_imagine_workspace_bytes(const drm_dims* dims, int32_t B, int32_t H);
int drm_imagine_block_f32(const drm_imagine_weights* w, const drm_imagine_io* io, drm_stream_t stream);
/* Destroys the recorded graphs (at most 16 are kept, oldest evicted first). */
int drm_graph_cache_clear(void);

/* The same block for callers that cannot allocate device memory or pass a workspace: code running under a
 * HIP graph capture of the caller's own, tight serving loops, harnesses that forbid hipMalloc.  No counterpart
 * in the reference.  Same arguments, kernels and results as drm_imagine_block_f32, except:
 *   - io->workspace / io->workspace_bytes are ignored: the workspace is a file-scope device array of
 *     DRM_IMAGINE_NOALLOC_MAX_WORKSPACE bytes inside the library (each GPU the library runs on holds its own
 *     copy until the process ends).  A call whose drm_imagine_workspace_bytes(dims, B, H) exceeds that is
 *     rejected with DRM_ENOTSUP (at the default widths, H = 15: up to 72 trajectories);
 *   - io->use_graph must be 0 and per-kernel timing must be off (both create runtime objects); the call is always one
 *     chain of launches on the caller's stream (the side-by-side chains of drm_imagine_block_f32 at many trajectories
 *     create streams and events), and with recurrent widths beyond 512 and a reduced precision the bf16 weight images
 *     of the call live in the same array (inside the size that is checked);
 *   - it calls no hipMalloc / hipFree or any other allocating routine; the array's device address is looked up
 *     once per device on the first call made with that device current;
 *   - every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two streams, two
 *     threads) race.  Nothing in the array needs to start at zero; each region is written before it is read
 *     within a call, so the contents left by an earlier call are never seen. */
#define DRM_IMAGINE_NOALLOC_MAX_WORKSPACE (16u << 20)
int drm_imagine_block_noalloc_f32(const drm_imagine_weights* w, const drm_imagine_io* io, drm_stream_t stream);

/* ---- observe (posterior) recurrence: the time loop of lib/world_model.py:563-589 ---------------
 * For t = 0..T-1 and B replayed sequences (c = c_prev[b,t] in {0,1}):
 *     z_in  = c ? z_{t-1} : one_hot(argmax softmax(1/K - log expo_reset[t,b]))   (lib/world_model.py:57-61)
 *     h_in  = c * h_{t-1};   x = lin_b + sum_l lin_w[:, l*K + z_in[l]] + c * lin_w[:, L*K:] a_emb[a_prev[b,t]]
 *     h_t   = GRUCell(x, h_in)                                                       (lib/world_model.py:64-71)
 *     post  = e_part[b,t] + fc_w_h h_t      (e_part = enc.fc over the hoisted conv features, bias included)
 *     z_t   = straight-through Gumbel-argmax of log_unimix(post) with expo_z[t,b]    (lib/world_model.py:459-461)
 * Everything the backward pass needs is written out.  The decoder, prior MLP, heads and losses do
 * not depend on the recurrence order and are evaluated by the caller over all B*T states. */
typedef struct drm_observe_weights {
    drm_dims dims;           /* D, L, K, A, E, unimix_eps used */
    const float* a_emb;      /* (A,E) */
    const float* lin_w;      /* seq.linear.weight (D, L*K+E) */
    const float* lin_b;
    const float* gru_w_ih; const float* gru_w_hh; const float* gru_b_ih; const float* gru_b_hh;
    const float* fc_w_h;     /* enc.fc.0.weight[:, conv_out:]  (L*K, D) with row stride ld_fc */
    int64_t ld_fc;
    const float* pack;       /* written by drm_pack_observe */
    size_t pack_bytes;
} drm_observe_weights;

size_t drm_pack_observe_bytes(const drm_dims* dims);
int drm_pack_observe(const drm_observe_weights* w, float* pack_out, size_t pack_bytes, drm_stream_t stream);
/* The same with the matrix product that fills the (latent, class) table of GRU input rows in the precision of the scan that
 * will read it (drm_observe_io.precision: 0 fp32, 1 bf16 matrix inputs with fp32 accumulation). */
int drm_pack_observe_prec(const drm_observe_weights* w, float* pack_out, size_t pack_bytes, int32_t precision, drm_stream_t stream);

typedef struct drm_observe_io {
    int32_t B, T;
    const float* h0;          /* (B,D) */
    const uint8_t* zidx0;     /* (B,L) */
    const int64_t* a_prev;    /* (B,T) */
    const float* c_prev;      /* (B,T) values 0 or 1 */
    const float* expo_reset;  /* (T,B,L,K) */
    const float* expo_z;      /* (T,B,L,K) */
    const float* e_part;      /* (B,T,L*K) */
    float* h_out;             /* (B,T,D) */
    uint8_t* zidx_out;        /* (B,T,L) sampled z_t */
    uint8_t* zidx_in_out;     /* (B,T,L) z fed into step t after the reset */
    float* z_out;             /* (B,T,L,K) dense one-hot of z_t */
    float* post_out;          /* (B,T,L,K) posterior logits */
    float* y_out;             /* (B,T,L,K) soft sample probabilities */
    float* gates_out;         /* (B,T,4D): r, u, n, gh_n */
    float* x_out;             /* (B,T,D) GRU input */
    float* h_in_out;          /* (B,T,D) c * h_{t-1} */
    void* workspace; size_t workspace_bytes;
    int32_t use_graph;
    int32_t precision;        /* recurrent widths > 512 only: 0 = fp32 products, 1 = operands rounded to bf16 on their way
                                 into the matrix cores (fp32 accumulation; gates, draw and reset stay fp32) */
} drm_observe_io;

size_t drm_observe_workspace_bytes(const drm_dims* dims, int32_t B, int32_t T);
int drm_observe_forward_f32(const drm_observe_weights* w, const drm_observe_io* io, drm_stream_t stream);

/* Reverse-time pass of the observe recurrence (what autograd's backward through the loop of
 * lib/world_model.py:563-589 computes).  Inputs: the tensors saved by drm_observe_forward_f32,
 * the upstream gradients wrt h (B,T,D), wrt the straight-through z (B,T,L,K) and wrt the
 * posterior logits (B,T,L,K), and transposed copies of the recurrent weights.  Outputs: per-step
 * d(gi), d(gh) (B,T,3D), d(x) (B,T,D) and d(post) (B,T,L*K); the weight gradients are products of
 * these with the saved activations and are formed by the caller as large batched GEMMs.
 *   wmix = exp(log_softmax(post) + log(1-eps) - log_unimix(post)),  psoft = softmax(post). */
typedef struct drm_observe_bwd_io {
    int32_t B, T;
    const float* c_prev;      /* (B,T) */
    const float* y;           /* (B,T,L*K) */
    const float* wmix;        /* (B,T,L*K) */
    const float* psoft;       /* (B,T,L*K) */
    const float* gates;       /* (B,T,4D) */
    const float* h_in;        /* (B,T,D) */
    const float* g_h;         /* (B,T,D) upstream */
    const float* g_z;         /* (B,T,L*K) upstream */
    const float* g_post;      /* (B,T,L*K) upstream */
    const float* fc_w_hT;     /* (D, L*K)  = enc.fc.0.weight[:, conv_out:]^T */
    const float* gru_w_ihT;   /* (D, 3D)   = weight_ih^T */
    const float* gru_w_hhT;   /* (D, 3D)   = weight_hh^T */
    const float* lin_zT;      /* (L*K, D)  = seq.linear.weight[:, :L*K]^T: the START of the buffer written by
                                 drm_pack_observe (the other packed tables are read at their offsets behind it) */
    float* d_gi;              /* (B,T,3D) */
    float* d_gh;              /* (B,T,3D) */
    float* d_x;               /* (B,T,D) */
    float* d_post;            /* (B,T,L*K) */
    void* workspace; size_t workspace_bytes;   /* >= drm_observe_bwd_workspace_bytes */
    int32_t precision;        /* as in drm_observe_io (recurrent widths > 512 only) */
    const float* gru_w_ih;    /* optional, recurrent widths > 512 with 3D % 32 == 0: weight_ih (3D, D) as it lies.  When set,
                                 d_x = d_gi W_ih is formed from it (drm_gemm_split_f32) and gru_w_ihT may be NULL: no
                                 transposed copy of the 3D x D weight per call */
} drm_observe_bwd_io;

size_t drm_observe_bwd_workspace_bytes(const drm_dims* dims, int32_t B);
int drm_observe_backward_f32(const drm_dims* dims, const drm_observe_bwd_io* io, drm_stream_t stream);

/* ---- LayerNorm over a whole sample fused with SiLU: the stage nonlinearity of the conv towers,
 * nn.LayerNorm([C,H,W]) followed by nn.SiLU (lib/world_model.py:94-108, :290-302) --------------
 * x, y, dy, dx: (N, E) contiguous with E = C*H*W (a multiple of 4, at most 32768); gamma, beta: (E).
 *   forward : y = SiLU((x - mean_n) * rstd_n * gamma + beta), biased variance, rstd = 1/sqrt(var + eps);
 *             mean, rstd (N) are saved for the backward pass.
 *   backward: dx (N, E), and param_partials (slices, 2, E) whose sum over the first axis is
 *             [d gamma; d beta] (slices = drm_ln_silu_slices(N, E); no atomics are used).
 * channel_bias (optional, C = E / hw floats, hw = H*W a multiple of 4): the bias of the convolution that
 * produced x (nn.Conv2d / nn.ConvTranspose2d bias of the same stage) is added on the way in, i.e. the
 * block computes SiLU(LayerNorm(x + b[c])) and the convolution runs without its bias pass.  Backward then
 * needs sample_sums (drm_ln_silu_sums_floats(N, E) floats, scratch: 2N, and for many long samples the per-wave sums
 * of the pass that walks the samples) and writes param_partials (slices, 3, E); the third row summed
 * over slices and over the hw positions of a channel is d b[c].  sample_sums without a channel bias
 * also selects the three-row form: the third row summed over slices is sum_n dx[n, e], the bias
 * gradient of a Linear in front of nn.LayerNorm(E) + nn.SiLU (the MLP blocks of lib/world_model.py:153-160).
 * channels_last != 0: x, y, dy, dx hold each sample as (H*W positions) x (C = E / hw channels), the memory
 * format the convolution backend serves fastest; gamma, beta and every row of param_partials stay in the
 * module's (C, H, W) order. C must be a multiple of 4. */
int drm_ln_silu_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int64_t N, int32_t E, float eps, const float* channel_bias, int32_t hw, int32_t channels_last,
                        drm_stream_t stream);
int32_t drm_ln_silu_slices(int64_t N, int32_t E);
size_t drm_ln_silu_sums_floats(int64_t N, int32_t E);
int drm_ln_silu_backward(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, float* dx, float* param_partials, int64_t N, int32_t E,
                         const float* channel_bias, int32_t hw, int32_t channels_last, float* sample_sums,
                         drm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DREAMER_AMD_H */
