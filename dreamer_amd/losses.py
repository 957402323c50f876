"""Fused loss terms of the world-model update on the GPU (lib/world_model.py:463-528).

``kl_pair``: both KL terms between the unimix'ed posterior and prior (dynamics: gradient to the prior
only; representation: to the posterior only) from one forward and one backward kernel.
``mse_rows``: 0.5 * mean squared reconstruction error per frame.
Host tensors (and class counts other than 32) take the eager formulas of ``WorldModel``.
"""
from __future__ import annotations

import torch

from . import _native as N


class _KLPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, post, prior, eps, free_nats):
        post, prior = post.contiguous(), prior.contiguous()
        rows, L, K = post.shape
        out = torch.empty(2, rows, dtype=torch.float32, device=post.device)
        N.check(N.lib().drm_kl_pair_forward(post.data_ptr(), prior.data_ptr(), rows, L, K, eps, free_nats, out[0].data_ptr(),
                                            out[1].data_ptr(), N.stream_ptr(post.device)), "drm_kl_pair_forward")
        ctx.save_for_backward(post, prior)
        ctx.eps, ctx.free_nats = eps, free_nats
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_dyn, g_rep):
        post, prior = ctx.saved_tensors
        rows, L, K = post.shape
        d_post, d_prior = torch.empty_like(post), torch.empty_like(prior)
        g_dyn = None if g_dyn is None else g_dyn.contiguous()
        g_rep = None if g_rep is None else g_rep.contiguous()
        N.check(N.lib().drm_kl_pair_backward(post.data_ptr(), prior.data_ptr(), rows, L, K, ctx.eps, ctx.free_nats, N.ptr(g_dyn),
                                             N.ptr(g_rep), d_post.data_ptr(), d_prior.data_ptr(), N.stream_ptr(post.device)),
                "drm_kl_pair_backward")
        return d_post, d_prior, None, None


def kl_pair(post_logits: torch.Tensor, prior_logits: torch.Tensor, unimix_eps: float, free_nats: float):
    """(dynamics rows, representation rows), each (rows,): sum over latents of max(KL, free_nats)."""
    return _KLPairFn.apply(post_logits.float(), prior_logits.float(), float(unimix_eps), float(free_nats))


class _MseRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_hat, x):
        x_hat, x = x_hat.contiguous(), x.contiguous()
        rows = x_hat.shape[0]
        E = x_hat.numel() // max(rows, 1)
        out = torch.empty(rows, dtype=torch.float32, device=x_hat.device)
        N.check(N.lib().drm_mse_rows_forward(x_hat.data_ptr(), x.data_ptr(), rows, E, out.data_ptr(), N.stream_ptr(x_hat.device)),
                "drm_mse_rows_forward")
        ctx.save_for_backward(x_hat, x)
        return out

    @staticmethod
    def backward(ctx, g):
        x_hat, x = ctx.saved_tensors
        rows = x_hat.shape[0]
        d = torch.empty_like(x_hat)
        N.check(N.lib().drm_mse_rows_backward(x_hat.data_ptr(), x.data_ptr(), rows, x_hat.numel() // max(rows, 1),
                                              g.contiguous().data_ptr(), d.data_ptr(), N.stream_ptr(x_hat.device)),
                "drm_mse_rows_backward")
        return d, None


def mse_rows(x_hat: torch.Tensor, x_true: torch.Tensor) -> torch.Tensor:
    """0.5 * mean over all but the first dimension of (x_hat - x_true)^2; gradient to ``x_hat`` only."""
    return _MseRowsFn.apply(x_hat.float(), x_true.detach().float())


class _CriticLossRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, slow_logits, returns, lo, hi, reg):
        rows, nb = logits.shape
        out = torch.empty(rows, dtype=torch.float32, device=logits.device)
        N.check(N.lib().drm_critic_loss_forward(logits.data_ptr(), N.ptr(slow_logits), returns.data_ptr(), rows, nb, lo, hi, reg,
                                                out.data_ptr(), N.stream_ptr(logits.device)), "drm_critic_loss_forward")
        ctx.save_for_backward(logits, slow_logits, returns)
        ctx.conf = (lo, hi, reg)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, slow_logits, returns = ctx.saved_tensors
        lo, hi, reg = ctx.conf
        rows, nb = logits.shape
        d = torch.empty_like(logits)
        N.check(N.lib().drm_critic_loss_backward(logits.data_ptr(), N.ptr(slow_logits), returns.data_ptr(), g.contiguous().data_ptr(),
                                                 rows, nb, lo, hi, reg, d.data_ptr(), N.stream_ptr(logits.device)),
                "drm_critic_loss_backward")
        return d, None, None, None, None, None


def critic_loss_rows(logits: torch.Tensor, slow_logits, returns: torch.Tensor, bucket_min: float, bucket_max: float,
                     reg: float) -> torch.Tensor:
    """Per-state critic loss (two-hot cross-entropy + reg * KL towards the slow critic), (rows,); gradient to
    ``logits`` only."""
    nb = logits.shape[-1]
    slow2 = None if slow_logits is None else slow_logits.detach().reshape(-1, nb).float().contiguous()
    return _CriticLossRowsFn.apply(logits.reshape(-1, nb).float().contiguous(), slow2,
                                   returns.detach().reshape(-1).float().contiguous(), float(bucket_min), float(bucket_max), float(reg))


def fused_ok(t: torch.Tensor, classes: int = 32) -> bool:
    return t.is_cuda and t.dtype == torch.float32 and classes == 32
