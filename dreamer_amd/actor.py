"""Actor with the reference's surface (lib/actor.py): MLP over cat[h, z_flat] -> unimix
categorical policy; REINFORCE loss with an EMA 5-95 percentile return scale."""
from __future__ import annotations

import torch
import torch.nn as nn

from .utils import log_unimix
from .world_model import WorldModelState, _norm_act_mlp


def state_features(s: WorldModelState) -> torch.Tensor:
    """cat[h, z_flat] on the last dim for (B, .) or (B, T, .) states."""
    return torch.cat([s.h, s.z.reshape(*s.h.shape[:-1], -1)], dim=-1)


class EMAPercentileScale(nn.Module):
    """S = max(EMA(q95) - EMA(q05), min_scale) of the returns (lib/actor.py:8-33).

    ``gather`` (optional callable) maps the local flat returns to the returns of all data-parallel
    ranks, so that every rank tracks the same percentiles as a single process with the full batch.
    """

    def __init__(self, decay: float = 0.99, min_scale: float = 1.0):
        super().__init__()
        self.decay, self.min_scale = decay, min_scale
        self.register_buffer("p05", torch.tensor(0.0))
        self.register_buffer("p95", torch.tensor(0.0))
        self.gather = None

    def update_get_S(self, x: torch.Tensor) -> torch.Tensor:
        x = x.detach().reshape(-1).float()
        if x.numel() == 0:
            return (self.p95 - self.p05).clamp_min(self.min_scale)
        if self.gather is not None:
            x = self.gather(x)
        if x.is_cuda and x.numel() <= 16384:
            from . import _native as N   # sort, both quantiles, the EMA update and the scale in one launch
            return N.percentile_scale(x.contiguous(), self.p05, self.p95, self.decay, self.min_scale, self.training)
        lo = torch.quantile(x, 0.05)
        hi = torch.maximum(torch.quantile(x, 0.95), lo + 1e-8)
        if self.training:
            self.p05.mul_(self.decay).add_((1.0 - self.decay) * lo)
            self.p95.mul_(self.decay).add_((1.0 - self.decay) * hi)
        return (self.p95 - self.p05).clamp_min(self.min_scale)


class Actor(nn.Module):
    def __init__(self, state_size: int, action_size: int, mlp_hidden_units: int = 512, mlp_layers: int = 3,
                 entropy_scale: float = 1e-3, ret_norm_limit: float = 1.0, ret_norm_decay: float = 0.99,
                 unimix_eps: float = 0.01):
        super().__init__()
        self.entropy_scale, self.unimix_eps = entropy_scale, unimix_eps
        self.mlp = _norm_act_mlp(state_size, mlp_hidden_units, mlp_layers)
        self.head = nn.Linear(mlp_hidden_units, action_size)
        nn.init.zeros_(self.head.weight)  # uniform initial policy
        nn.init.zeros_(self.head.bias)
        self.ret_scale = EMAPercentileScale(decay=ret_norm_decay, min_scale=ret_norm_limit)

    _state_vec = staticmethod(state_features)

    def forward(self, model_states: WorldModelState) -> torch.distributions.Distribution:
        x = state_features(model_states)
        logits = self.head(self.mlp(x.reshape(-1, x.size(-1)))).reshape(*x.shape[:-1], -1)
        return torch.distributions.Categorical(logits=log_unimix(logits, self.unimix_eps, dim=-1))

    def loss(self, model_states: WorldModelState, actions: torch.Tensor, returns: torch.Tensor,
             values: torch.Tensor) -> torch.Tensor:
        return self.loss_with_scale(model_states, actions, returns, values, self.ret_scale.update_get_S(returns.detach()))

    def loss_with_scale(self, model_states: WorldModelState, actions: torch.Tensor, returns: torch.Tensor,
                        values: torch.Tensor, scale: torch.Tensor, validate: bool = False) -> torch.Tensor:
        """The loss of lib/actor.py:108-125 for a return scale S that has already been updated.  The
        distribution is built without argument validation (its checks read results back to the host,
        which a recorded graph cannot do); the arithmetic is the same."""
        x = state_features(model_states)
        logits = self.head(self.mlp(x.reshape(-1, x.size(-1)))).reshape(*x.shape[:-1], -1)
        dist = torch.distributions.Categorical(logits=log_unimix(logits, self.unimix_eps, dim=-1), validate_args=validate)
        advantage = ((returns - values) / scale).detach()
        return -(advantage * dist.log_prob(actions)).mean() - self.entropy_scale * dist.entropy().mean()
