"""Critic with the reference's surface (lib/critic.py): two-hot symlog bucket classifier over
cat[h, z_flat] with an EMA "slow" copy used as a regulariser."""
from __future__ import annotations

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from .actor import state_features
from .utils import symexp, symlog
from .world_model import WorldModelState, _Linear, _norm_act_mlp


class Critic(nn.Module):
    def __init__(self, state_size: int, mlp_hidden_units: int = 512, mlp_layers: int = 3, num_buckets: int = 255,
                 bucket_min: float = -20.0, bucket_max: float = 20.0, ema_decay: float = 0.98,
                 ema_regularizer: float = 1.0):
        super().__init__()
        self.num_buckets, self.bucket_min, self.bucket_max = num_buckets, bucket_min, bucket_max
        self.bucket_delta = (bucket_max - bucket_min) / (num_buckets - 1)
        self.register_buffer("buckets", torch.linspace(bucket_min, bucket_max, num_buckets))
        self.mlp = _norm_act_mlp(state_size, mlp_hidden_units, mlp_layers)
        self.head = _Linear(mlp_hidden_units, num_buckets)   # nn.Linear; its 240-row evaluations on the library's product
        nn.init.zeros_(self.head.weight)
        nn.init.zeros_(self.head.bias)
        self.ema_decay, self.ema_regularizer = ema_decay, ema_regularizer
        # frozen EMA copy; registered as a submodule, so its tensors appear under "slow." in the
        # state dict and in parameters() exactly as in the reference (lib/critic.py:51-53)
        self.slow = copy.deepcopy(self)
        self.slow.requires_grad_(False)

    _state_vec = staticmethod(state_features)

    def forward(self, models_states: WorldModelState) -> torch.Tensor:
        x = state_features(models_states)
        return self.head(self.mlp(x.reshape(-1, x.size(-1)))).reshape(*x.shape[:-1], self.num_buckets)

    def dist(self, models_states: WorldModelState) -> torch.distributions.Categorical:
        return torch.distributions.Categorical(logits=self.forward(models_states))

    def value(self, model_states: WorldModelState) -> torch.Tensor:
        probs = self(model_states).softmax(dim=-1)
        return symexp((probs * self.buckets).sum(dim=-1))

    def _twohot(self, x_symlog: torch.Tensor) -> torch.Tensor:
        x = x_symlog.clamp(min=self.bucket_min, max=self.bucket_max)
        pos = (x - self.bucket_min) / self.bucket_delta
        lo = torch.floor(pos).long()
        hi = (lo + 1).clamp(max=self.num_buckets - 1)
        w_hi = (pos - lo.float()).clamp(0.0, 1.0)
        target = x.new_zeros((x.numel(), self.num_buckets))
        target.scatter_(1, lo.reshape(-1, 1), (1.0 - w_hi).reshape(-1, 1))
        target.scatter_add_(1, hi.reshape(-1, 1), w_hi.reshape(-1, 1))
        return target.view(*x.shape, self.num_buckets)

    @torch.no_grad()
    def update_slow(self):
        # parameters() yields this module's own tensors first and the slow copy's after them; the
        # zip stops at the slow copy's length, pairing each tensor with its EMA (lib/critic.py:136)
        pairs = list(zip(self.parameters(), self.slow.parameters()))
        if pairs and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and q.is_contiguous() for p, q in pairs):
            from . import _native as N
            N.ema_multi([p.data for p, _ in pairs], [q.data for _, q in pairs], self.ema_decay)   # one launch
            return
        for p, p_slow in pairs:
            p_slow.data.mul_(self.ema_decay)
            p_slow.data.add_(p.data * (1.0 - self.ema_decay))

    def loss(self, model_states: WorldModelState, returns: torch.Tensor) -> torch.Tensor:
        logits = self(model_states)
        if logits.is_cuda and logits.dtype == torch.float32 and self.num_buckets <= 256:
            # two-hot target, cross-entropy and the regulariser towards the slow critic from one row kernel each way
            from . import losses as FL
            slow_logits = None
            if self.ema_regularizer > 0.0:
                with torch.no_grad():
                    slow_logits = self.slow(model_states)
            return FL.critic_loss_rows(logits, slow_logits, returns, self.bucket_min, self.bucket_max,
                                       max(self.ema_regularizer, 0.0)).mean()
        return self._loss_eager(model_states, returns)

    def _loss_eager(self, model_states: WorldModelState, returns: torch.Tensor) -> torch.Tensor:
        target = self._twohot(symlog(returns.detach()))
        log_probs = F.log_softmax(self(model_states), dim=-1)
        loss = -(target * log_probs).sum(dim=-1).mean()
        if self.ema_regularizer > 0.0:
            with torch.no_grad():
                slow_probs = F.softmax(self.slow(model_states), dim=-1)
            reg = F.kl_div(log_probs, slow_probs, log_target=False, reduction="none").sum(dim=-1).mean()
            loss = loss + self.ema_regularizer * reg
        return loss
