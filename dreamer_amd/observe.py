"""Posterior (observe) recurrence over a replayed sequence.

Only this part of ``WorldModel.loss`` is sequential in time (lib/world_model.py:563-589):

    reset:   h <- h * c_prev;  z <- c_prev ? z : fresh uniform one-hot;  emb <- emb(a_prev) * c_prev
    x      = seq.linear(cat[z, emb])                    (no norm / activation: lib/world_model.py:69-71)
    h      = GRUCell(x, h)
    post   = e_part[t] + enc.fc.weight[:, conv:] @ h     (e_part = hoisted conv features @ W + bias)
    z      = straight-through Gumbel-argmax of log_unimix(post)

``observe_scan`` dispatches on the device of its inputs: HIP tensors go to the fused kernels
(``ObserveScanFn``, a torch.autograd.Function: native forward over all T steps, explicit reverse-time
backward), host tensors to the eager loop (the reference supports ``device="cpu"`` as well).

Recurrent widths up to 512 run on the register-resident few-row / small-tile kernels; wider states (up to
4096) take the same decomposition with every matrix product on the large-row MFMA kernel.  For those,
``wm.observe_precision = "bf16"`` rounds the operands of the products to bf16 on their way into the matrix
cores (fp32 accumulation; gates, draws and resets stay fp32): results within 1e-2 of the fp32 path.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn.functional as F

from .utils import log_unimix


def _scan_eager(wm, e_part, a_prev, c_prev, expo_reset, expo_z, state):
    from .world_model import WorldModelState, hard_sample
    B, T = a_prev.shape
    L, K = wm.num_latents, wm.classes_per_latent
    w_h = wm.enc.fc[0].weight[:, wm.enc.conv_out_size:]
    h, z = state.h, state.z
    hs, zs, posts = [], [], []
    for t in range(T):
        h = wm.seq(h, z, a_prev[:, t], c_prev[:, t:t + 1], expo_reset[t])
        post = (e_part[:, t] + F.linear(h, w_h)).view(B, L, K)
        z, _, _ = hard_sample(log_unimix(post, wm.unimix_eps), expo_z[t])
        hs.append(h); zs.append(z); posts.append(post)
    return torch.stack(hs, 1), torch.stack(zs, 1), torch.stack(posts, 1), WorldModelState(h=h, z=z)


def W_GRADIENT_STAGING() -> bool:
    from . import world_model as W
    return W._GRADIENT_STAGING


def _wgrad_bf16_ok(dy, D):
    """Shapes the bf16 large-row kernel takes for dy.T @ x: a long sum over the rows, many output rows."""
    rows, n = dy.shape
    return D > 512 and rows >= 256 and rows % 32 == 0 and n >= 256


def _wgrad_bf16(dy, x):
    """dy.T @ x with bf16 matrix inputs and fp32 accumulation: the operands read as they lie by the staging kernel (tiles
    transposed on their way into LDS); otherwise both transposed once so that the sum over the rows runs along contiguous
    memory, then one launch of the large-row kernel (output rows = columns of dy)."""
    from . import _native as N
    from . import world_model as W
    if W._GRADIENT_STAGING and N.gemm_split_ok(dy, x, True):
        return N.gemm_split(dy, x, True, bf16=True)
    return N.fused_linear(dy.t().contiguous(), x.t().contiguous(), bf16=True)


class _ObserveEngine:
    """Packed tables and workspace of one world model for the native scan."""

    def __init__(self, wm):
        from . import _native as N
        self.N = N
        p = wm.seq.linear.weight
        self.device = p.device
        self.dims = N.drm_dims(D=wm.hidden_size, L=wm.num_latents, K=wm.classes_per_latent, M=wm.hidden_size,
                               A=wm.action_size, E=wm.seq.a_emb.embedding_dim, NB=1, unimix_eps=wm.unimix_eps)
        nbytes = N.lib().drm_pack_observe_bytes(C.byref(self.dims))
        if nbytes == 0:
            raise N.NativeError(f"unsupported dimensions: {N.lib().drm_last_error().decode()}")
        self.pack = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        self.ws = None
        self.precision = 0

    def weights(self, a_emb, lin_w, lin_b, w_ih, w_hh, b_ih, b_hh, fc_w_h):
        N = self.N
        w = N.drm_observe_weights()
        w.dims = self.dims
        w.a_emb, w.lin_w, w.lin_b = a_emb.data_ptr(), lin_w.data_ptr(), lin_b.data_ptr()
        w.gru_w_ih, w.gru_w_hh, w.gru_b_ih, w.gru_b_hh = w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
        w.fc_w_h, w.ld_fc = fc_w_h.data_ptr(), fc_w_h.stride(0)
        w.pack, w.pack_bytes = self.pack.data_ptr(), self.pack.numel() * 4
        return w


class ObserveScanFn(torch.autograd.Function):
    """Native forward over all T steps; reverse-time backward from the saved per-step tensors."""

    @staticmethod
    def forward(ctx, eng, e_part, a_emb, lin_w, lin_b, w_ih, w_hh, b_ih, b_hh, fc_w_h, h0, zidx0, a_prev, c_prev,
                expo_reset, expo_z):
        N = eng.N
        B, T = a_prev.shape
        d = eng.dims
        D, L, K = d.D, d.L, d.K
        dev = e_part.device
        e_part = e_part.contiguous()
        w = eng.weights(a_emb, lin_w, lin_b, w_ih, w_hh, b_ih, b_hh, fc_w_h)
        N.check(N.lib().drm_pack_observe_prec(C.byref(w), eng.pack.data_ptr(), eng.pack.numel() * 4, eng.precision,
                                              N.stream_ptr(dev)), "drm_pack_observe_prec")
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        h = torch.empty(B, T, D, **f32); zidx = torch.empty(B, T, L, **u8); zidx_in = torch.empty(B, T, L, **u8)
        z = torch.empty(B, T, L, K, **f32); post = torch.empty(B, T, L, K, **f32); y = torch.empty(B, T, L, K, **f32)
        gates = torch.empty(B, T, 4 * D, **f32); x = torch.empty(B, T, D, **f32); h_in = torch.empty(B, T, D, **f32)
        ws_bytes = N.lib().drm_observe_workspace_bytes(C.byref(d), B, T)
        if eng.ws is None or eng.ws.numel() * 4 < ws_bytes:
            eng.ws = torch.empty(ws_bytes // 4 + 4, **f32)
        io = N.drm_observe_io()
        io.B, io.T = B, T
        h0c, c_prev, a_prev = h0.contiguous(), c_prev.contiguous(), a_prev.contiguous()
        expo_reset, expo_z = expo_reset.contiguous(), expo_z.contiguous()
        io.h0, io.zidx0, io.a_prev, io.c_prev = h0c.data_ptr(), zidx0.data_ptr(), a_prev.data_ptr(), c_prev.data_ptr()
        io.expo_reset, io.expo_z, io.e_part = expo_reset.data_ptr(), expo_z.data_ptr(), e_part.data_ptr()
        io.h_out, io.zidx_out, io.zidx_in_out, io.z_out = h.data_ptr(), zidx.data_ptr(), zidx_in.data_ptr(), z.data_ptr()
        io.post_out, io.y_out, io.gates_out = post.data_ptr(), y.data_ptr(), gates.data_ptr()
        io.x_out, io.h_in_out = x.data_ptr(), h_in.data_ptr()
        io.workspace, io.workspace_bytes = eng.ws.data_ptr(), eng.ws.numel() * 4
        io.use_graph = 0
        io.precision = eng.precision
        N.check(N.lib().drm_observe_forward_f32(C.byref(w), C.byref(io), N.stream_ptr(dev)), "drm_observe_forward_f32")
        ctx.save_for_backward(a_emb, lin_w, w_ih, w_hh, fc_w_h, a_prev, c_prev, h, zidx_in, post, y, gates, x, h_in)
        ctx.eps = float(d.unimix_eps)
        ctx.eng = eng
        ctx.mark_non_differentiable(zidx)
        return h, z, post, zidx

    @staticmethod
    def backward(ctx, g_h, g_z, g_post, _g_zidx):
        a_emb, lin_w, w_ih, w_hh, fc_w_h, a_prev, c_prev, h, zidx_in, post, y, gates, x, h_in = ctx.saved_tensors
        eng = ctx.eng
        N = eng.N
        B, T, D = h.shape
        L, K = post.shape[2], post.shape[3]
        LK = L * K
        eps = ctx.eps
        dev = h.device
        log1m, floor = math.log1p(-eps), math.log(eps) - math.log(K)
        g_h = torch.zeros_like(h) if g_h is None else g_h.contiguous()
        g_z = torch.zeros_like(post) if g_z is None else g_z.contiguous()
        g_post = torch.zeros_like(post) if g_post is None else g_post.contiguous()
        # time-parallel pieces of the unimix / log-softmax backward
        ls = F.log_softmax(post, dim=-1)
        w_mix = torch.exp(ls + log1m - torch.logaddexp(ls + log1m, torch.full((), floor, dtype=ls.dtype, device=ls.device)))
        p_soft = ls.exp()
        # wide models form d(x) from weight_ih as it lies: no transposed copy of the (3D, D) weight
        x_from_w_ih = D > 512 and (3 * D) % 32 == 0 and w_ih.is_contiguous() and W_GRADIENT_STAGING()
        fcT, hhT = fc_w_h.t().contiguous(), w_hh.t().contiguous()
        ihT = None if x_from_w_ih else w_ih.t().contiguous()
        d_gi = torch.empty(B, T, 3 * D, device=dev, dtype=h.dtype)
        d_gh = torch.empty_like(d_gi)
        d_x = torch.empty_like(x)
        d_post = torch.empty_like(post)
        ws_bytes = N.lib().drm_observe_bwd_workspace_bytes(C.byref(eng.dims), B)
        ws = torch.empty(ws_bytes // 4 + 4, device=dev, dtype=torch.float32)
        io = N.drm_observe_bwd_io()
        io.B, io.T = B, T
        io.c_prev, io.y, io.wmix, io.psoft = c_prev.data_ptr(), y.data_ptr(), w_mix.data_ptr(), p_soft.data_ptr()
        io.gates, io.h_in = gates.data_ptr(), h_in.data_ptr()
        io.g_h, io.g_z, io.g_post = g_h.data_ptr(), g_z.data_ptr(), g_post.data_ptr()
        io.fc_w_hT, io.gru_w_ihT, io.gru_w_hhT, io.lin_zT = fcT.data_ptr(), N.ptr(ihT), hhT.data_ptr(), eng.pack.data_ptr()
        io.gru_w_ih = w_ih.data_ptr() if x_from_w_ih else None
        io.d_gi, io.d_gh, io.d_x, io.d_post = d_gi.data_ptr(), d_gh.data_ptr(), d_x.data_ptr(), d_post.data_ptr()
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        io.precision = eng.precision
        N.check(N.lib().drm_observe_backward_f32(C.byref(eng.dims), C.byref(io), N.stream_ptr(dev)), "drm_observe_backward_f32")
        # weight gradients: one batched product each over all B*T rows
        gi2, gh2, dx2 = d_gi.reshape(B * T, -1), d_gh.reshape(B * T, -1), d_x.reshape(B * T, D)
        c2 = c_prev.reshape(B * T, 1)
        a2 = a_prev.reshape(B * T)
        from .world_model import _mm_dgrad, _mm_wgrad
        z_in = F.one_hot(zidx_in.reshape(B * T, L).long(), K).to(h.dtype).reshape(B * T, LK)
        a_vec = a_emb[a2] * c2
        # the two products over the recurrent width: with bf16 matrix inputs when the scan ran that way
        # (otherwise, when large enough, in the split form: fp32 accuracy on the bf16 matrix pipe)
        from . import world_model as W
        if eng.precision == 1 and _wgrad_bf16_ok(gi2, D):
            wide_wgrad = _wgrad_bf16
        else:
            wide_wgrad = lambda dy_, x_: _mm_wgrad(dy_, x_, "f32" if W._PRODUCT_FORM == "f32" else "split")
        g_w_ih = wide_wgrad(gi2, x.reshape(B * T, D))
        g_w_hh = wide_wgrad(gh2, h_in.reshape(B * T, D))
        g_lin_w = torch.cat([_mm_wgrad(dx2, z_in), _mm_wgrad(dx2, a_vec)], dim=1)
        g_a_emb = torch.zeros_like(a_emb).index_add_(0, a2, _mm_dgrad(dx2, lin_w[:, LK:]) * c2)
        g_fc = _mm_wgrad(d_post.reshape(B * T, LK), h.reshape(B * T, D))
        return (None, d_post.reshape(B, T, LK), g_a_emb, g_lin_w, N.sum_rows(dx2), g_w_ih, g_w_hh, N.sum_rows(gi2), N.sum_rows(gh2), g_fc,
                None, None, None, None, None, None)


def observe_scan(wm, e_part, a_prev, c_prev, expo_reset, expo_z, state, engine_attr: str = "_observe_engine"):
    """Returns (h (B,T,D), z (B,T,L,K), post_logits (B,T,L,K), last WorldModelState).

    ``engine_attr`` names the attribute of ``wm`` that holds the packed tables and workspace; a caller
    whose launches are recorded into a graph uses one of its own, so that no other caller ever
    reallocates the buffers the recording points to."""
    if not e_part.is_cuda:
        return _scan_eager(wm, e_part, a_prev, c_prev, expo_reset, expo_z, state)
    from . import _native as N
    from .world_model import WorldModelState
    eng = getattr(wm, engine_attr, None)
    if eng is None:
        eng = _ObserveEngine(wm)
        object.__setattr__(wm, engine_attr, eng)
    prec = getattr(wm, "observe_precision", "f32")
    if prec not in ("f32", "bf16"):
        raise ValueError(f"observe_precision must be 'f32' or 'bf16' (got {prec!r})")
    eng.precision = 1 if (prec == "bf16" and wm.hidden_size > 512) else 0
    B, T = a_prev.shape
    L, K = wm.num_latents, wm.classes_per_latent
    g = wm.seq.gru
    fc_w_h = wm.enc.fc[0].weight[:, wm.enc.conv_out_size:]
    zidx0 = N.onehot_to_index(state.z.detach().contiguous().float())
    h, z, post, zidx = ObserveScanFn.apply(eng, e_part, wm.seq.a_emb.weight, wm.seq.linear.weight, wm.seq.linear.bias,
                                           g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh, fc_w_h, state.h.detach(), zidx0,
                                           a_prev, c_prev, expo_reset, expo_z)
    return h, z, post, WorldModelState(h=h[:, -1], z=z[:, -1])
