"""Posterior (observe) recurrence over a replayed sequence.

Only this part of ``WorldModel.loss`` is sequential in time (lib/world_model.py:563-589):

    reset:   h <- h * c_prev;  z <- c_prev ? z : fresh uniform one-hot;  emb <- emb(a_prev) * c_prev
    x      = seq.linear(cat[z, emb])                    (no norm / activation: lib/world_model.py:69-71)
    h      = GRUCell(x, h)
    post   = e_part[t] + enc.fc.weight[:, conv:] @ h     (e_part = hoisted conv features @ W + bias)
    z      = straight-through Gumbel-argmax of log_unimix(post)

``observe_scan`` dispatches on the device of its inputs: HIP tensors go to the fused kernels,
host tensors to the eager loop below (the reference supports ``device="cpu"`` as well).
"""
from __future__ import annotations

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


def observe_scan(wm, e_part, a_prev, c_prev, expo_reset, expo_z, state):
    """Returns (h (B,T,D), z (B,T,L,K), post_logits (B,T,L,K), last WorldModelState)."""
    return _scan_eager(wm, e_part, a_prev, c_prev, expo_reset, expo_z, state)
