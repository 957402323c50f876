"""CPU restatement of the reference Dreamer hot path (test infrastructure only).

Every function is a functional (weights-dict in, tensors out) restatement of
one piece of the reference, in plain PyTorch-CPU fp32 eager ops or numpy, and
names the reference lines it follows.  Weight dicts use the reference's
``state_dict`` key names (``seq.linear.weight``, ``dyn.mlp.0.weight`` ...), so a
state dict taken from the reference modules or from ``dreamer_amd`` modules can
be passed in unchanged.

All randomness enters as explicit ``Exp(1)`` noise tensors, in the order the
reference consumes its generator:
  * ``F.gumbel_softmax`` draws ``E = empty_like(logits).exponential_()`` and
    uses ``g = -log(E)`` (torch/nn/functional.py, gumbel_softmax);
  * ``Categorical.sample`` -> ``torch.multinomial(probs, 1)`` draws
    ``E = empty_like(probs).exponential_()`` and returns ``argmax(probs / E)``.

Parity status: pinned against golden vectors captured from the imported
reference (tests/golden/make_golden.py -> tests/golden/*.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]

__all__ = [
    "symlog", "symexp", "log_unimix", "gumbel_hard", "mlp_stack", "count_mlp_layers",
    "seq_forward", "dyn_logits", "scalar_head", "decoder_forward", "encoder_conv",
    "encoder_logits", "prior_step", "posterior_step", "actor_log_probs", "sample_action",
    "critic_logits", "critic_value", "lambda_return", "imagine_block", "kl_loss",
    "prediction_loss", "wm_loss", "twohot", "critic_loss", "actor_loss", "percentile_scale",
    "update_slow", "ReplayOracle", "init_state",
]


# --------------------------------------------------------------------------- pointwise
def symlog(x: Tensor) -> Tensor:
    """lib/utils.py:12-13."""
    return torch.sign(x) * torch.log1p(torch.abs(x))


def symexp(x: Tensor) -> Tensor:
    """lib/utils.py:16-17."""
    return torch.sign(x) * torch.expm1(torch.abs(x))


def log_unimix(logits: Tensor, eps: float) -> Tensor:
    """log((1-eps)*softmax(logits) + eps/K) over the last dim; lib/utils.py:20-31."""
    k = logits.size(-1)
    ls = F.log_softmax(logits, dim=-1)
    return torch.logaddexp(ls + math.log1p(-float(eps)),
                           logits.new_tensor(math.log(float(eps)) - math.log(float(k))))


def gumbel_hard(logits: Tensor, expo: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """F.gumbel_softmax(logits, tau=1, hard=True) with its Exp(1) draw given as ``expo``.

    Returns (z, idx, y_soft): z is the forward value ``y_hard - y_soft + y_soft`` which
    is exactly one-hot; idx its argmax (first max on ties, as ``y_soft.max(dim)[1]``).
    Used by lib/world_model.py:59, :412, :461.
    """
    g = -torch.log(expo)
    y = torch.softmax(logits + g, dim=-1)
    idx = y.max(dim=-1)[1]
    hard = torch.zeros_like(logits).scatter_(-1, idx.unsqueeze(-1), 1.0)
    z = hard - y.detach() + y
    return z, idx, y


# --------------------------------------------------------------------------- MLP blocks
def count_mlp_layers(w: Weights, prefix: str) -> int:
    n = 0
    while f"{prefix}mlp.{3 * n}.weight" in w:
        n += 1
    return n


def mlp_stack(w: Weights, prefix: str, x: Tensor) -> Tensor:
    """n x (Linear -> LayerNorm(eps 1e-5) -> SiLU); the nn.Sequential at
    lib/world_model.py:153-160, :192-199, :236-243, lib/actor.py:60-67, lib/critic.py:34-41."""
    for i in range(count_mlp_layers(w, prefix)):
        x = F.linear(x, w[f"{prefix}mlp.{3 * i}.weight"], w[f"{prefix}mlp.{3 * i}.bias"])
        x = F.layer_norm(x, (x.size(-1),), w[f"{prefix}mlp.{3 * i + 1}.weight"],
                         w[f"{prefix}mlp.{3 * i + 1}.bias"], 1e-5)
        x = F.silu(x)
    return x


# --------------------------------------------------------------------------- world model pieces
def init_state(batch: int, hidden: int, latents: int, classes: int, expo: Tensor):
    """lib/world_model.py:403-413: h0 = 0, z0 = hard Gumbel sample of constant logits 1/K."""
    h0 = torch.zeros(batch, hidden)
    z0, _, _ = gumbel_hard(torch.full((batch, latents, classes), 1.0 / classes), expo)
    return h0, z0


def seq_forward(w: Weights, h: Tensor, z: Tensor, a: Tensor,
                c_prev: Optional[Tensor] = None, expo_reset: Optional[Tensor] = None) -> Tensor:
    """SequenceModel.forward, lib/world_model.py:39-71.  ``seq.norm`` is never applied.

    GRUCell gate order (r, z, n) as torch.nn.GRUCell.
    """
    if c_prev is not None:
        h = h * c_prev
        z_reset, _, _ = gumbel_hard(torch.full_like(z, 1.0 / z.size(-1)), expo_reset)
        m = c_prev.view(-1, 1, 1)
        z = z * m + (1.0 - m) * z_reset
    a_vec = w["seq.a_emb.weight"][a]
    if c_prev is not None:
        a_vec = a_vec * c_prev
    x = torch.cat([z.reshape(z.size(0), -1), a_vec], dim=-1)
    x = F.linear(x, w["seq.linear.weight"], w["seq.linear.bias"])
    gi = F.linear(x, w["seq.gru.weight_ih"], w["seq.gru.bias_ih"])
    gh = F.linear(h, w["seq.gru.weight_hh"], w["seq.gru.bias_hh"])
    i_r, i_z, i_n = gi.chunk(3, dim=-1)
    h_r, h_z, h_n = gh.chunk(3, dim=-1)
    r = torch.sigmoid(i_r + h_r)
    u = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1.0 - u) * n + u * h


def dyn_logits(w: Weights, h: Tensor, latents: int, classes: int) -> Tensor:
    """DynamicsPredictor.forward, lib/world_model.py:164-174."""
    y = mlp_stack(w, "dyn.", h)
    return F.linear(y, w["dyn.head.weight"], w["dyn.head.bias"]).view(-1, latents, classes)


def scalar_head(w: Weights, prefix: str, h: Tensor, z: Tensor) -> Tensor:
    """RewardPredictor / ContinuePredictor forward, lib/world_model.py:203-217, :247-261.
    prefix is 'rew.' or 'cont.'; input is cat[h, z_flat] (h first)."""
    x = torch.cat([h, z.reshape(z.size(0), -1)], dim=-1)
    y = mlp_stack(w, prefix, x)
    return F.linear(y, w[f"{prefix}head.weight"], w[f"{prefix}head.bias"])


def _spatial_ln(x: Tensor, g: Tensor, b: Tensor) -> Tensor:
    return F.layer_norm(x, tuple(g.shape), g, b, 1e-5)


def encoder_conv(w: Weights, x: Tensor) -> Tensor:
    """Encoder.conv, lib/world_model.py:94-108: 4 x (Conv k4 s2 p1 -> LayerNorm[C,H,W] -> SiLU), flatten."""
    for i in range(4):
        x = F.conv2d(x, w[f"enc.conv.{3 * i}.weight"], w[f"enc.conv.{3 * i}.bias"], stride=2, padding=1)
        x = _spatial_ln(x, w[f"enc.conv.{3 * i + 1}.weight"], w[f"enc.conv.{3 * i + 1}.bias"])
        x = F.silu(x)
    return x.flatten(1)


def encoder_logits(w: Weights, x: Tensor, h: Tensor, latents: int, classes: int) -> Tensor:
    """Encoder.forward, lib/world_model.py:120-134: fc(cat[conv(x), h])."""
    feat = torch.cat([encoder_conv(w, x), h], dim=-1)
    return F.linear(feat, w["enc.fc.0.weight"], w["enc.fc.0.bias"]).view(-1, latents, classes)


def decoder_forward(w: Weights, h: Tensor, z: Tensor) -> Tensor:
    """Decoder.forward, lib/world_model.py:304-320."""
    b = h.size(0)
    x = torch.cat([h, z.reshape(b, -1)], dim=-1)
    x = F.silu(F.linear(x, w["dec.fc.0.weight"], w["dec.fc.0.bias"]))
    c8 = w["dec.deconv.0.weight"].size(0)
    side = int(round(math.sqrt(x.size(1) // c8)))
    x = x.view(b, c8, side, side)
    for i in range(3):
        x = F.conv_transpose2d(x, w[f"dec.deconv.{3 * i}.weight"], w[f"dec.deconv.{3 * i}.bias"],
                               stride=2, padding=1)
        x = _spatial_ln(x, w[f"dec.deconv.{3 * i + 1}.weight"], w[f"dec.deconv.{3 * i + 1}.bias"])
        x = F.silu(x)
    x = F.conv_transpose2d(x, w["dec.deconv.9.weight"], w["dec.deconv.9.bias"], stride=2, padding=1)
    return torch.sigmoid(x)


def prior_step(w: Weights, h: Tensor, z: Tensor, a: Tensor, expo_z: Tensor, eps: float,
               run_decoder: bool = False):
    """WorldModel.step with x_cur=None, c_prev=None (imagination), lib/world_model.py:415-457."""
    latents, classes = z.size(-2), z.size(-1)
    h1 = seq_forward(w, h, z, a)
    logits = dyn_logits(w, h1, latents, classes)
    z1, idx, _ = gumbel_hard(log_unimix(logits, eps), expo_z)
    r_hat = scalar_head(w, "rew.", h1, z1)
    c_hat = scalar_head(w, "cont.", h1, z1)
    x_hat = decoder_forward(w, h1, z1) if run_decoder else None
    return h1, z1, {"prior_logits": logits, "idx": idx, "r_hat": r_hat, "c_hat": c_hat, "x_hat": x_hat}


def posterior_step(w: Weights, h: Tensor, z: Tensor, a: Tensor, x: Tensor, c_prev: Optional[Tensor],
                   expo_reset: Optional[Tensor], expo_z: Tensor, eps: float, run_decoder: bool = True):
    """WorldModel.step with an observation, lib/world_model.py:415-457."""
    latents, classes = z.size(-2), z.size(-1)
    h1 = seq_forward(w, h, z, a, c_prev, expo_reset)
    post = encoder_logits(w, x, h1, latents, classes)
    prior = dyn_logits(w, h1, latents, classes)
    z1, idx, y = gumbel_hard(log_unimix(post, eps), expo_z)
    r_hat = scalar_head(w, "rew.", h1, z1)
    c_hat = scalar_head(w, "cont.", h1, z1)
    x_hat = decoder_forward(w, h1, z1) if run_decoder else None
    return h1, z1, {"prior_logits": prior, "post_logits": post, "idx": idx, "y_soft": y,
                    "r_hat": r_hat, "c_hat": c_hat, "x_hat": x_hat}


# --------------------------------------------------------------------------- actor / critic
def _state_vec(h: Tensor, z: Tensor) -> Tensor:
    """lib/actor.py:76-92, lib/critic.py:55-71: cat[h, z_flat] on the last dim."""
    return torch.cat([h, z.reshape(*h.shape[:-1], -1)], dim=-1)


def actor_log_probs(wa: Weights, h: Tensor, z: Tensor, eps: float) -> Tensor:
    """Actor.forward up to the mixed log-probs, lib/actor.py:94-106; the Categorical built from
    them normalises once more (logits - logsumexp), which is restated here too."""
    x = _state_vec(h, z)
    lead = x.shape[:-1]
    y = mlp_stack(wa, "", x.reshape(-1, x.size(-1)))
    logits = F.linear(y, wa["head.weight"], wa["head.bias"]).reshape(*lead, -1)
    mixed = log_unimix(logits, eps)
    return mixed - mixed.logsumexp(dim=-1, keepdim=True)


def sample_action(logp: Tensor, expo: Tensor) -> Tensor:
    """Categorical.sample -> torch.multinomial(probs, 1): argmax(probs / Exp(1))."""
    probs = torch.softmax(logp, dim=-1)
    return torch.argmax(probs / expo, dim=-1)


def critic_logits(wc: Weights, h: Tensor, z: Tensor, prefix: str = "") -> Tensor:
    """Critic.forward, lib/critic.py:73-85 (prefix 'slow.' selects the EMA copy)."""
    x = _state_vec(h, z)
    lead = x.shape[:-1]
    y = mlp_stack(wc, prefix, x.reshape(-1, x.size(-1)))
    return F.linear(y, wc[f"{prefix}head.weight"], wc[f"{prefix}head.bias"]).reshape(*lead, -1)


def critic_value(wc: Weights, h: Tensor, z: Tensor) -> Tensor:
    """Critic.value, lib/critic.py:94-101."""
    probs = critic_logits(wc, h, z).softmax(dim=-1)
    return symexp((probs * wc["buckets"]).sum(dim=-1))


def lambda_return(R: Tensor, C: Tensor, V: Tensor, V_last: Tensor, gamma: float, lam: float) -> Tensor:
    """lib/trainer.py:217-224.  Uses (1 - C_t) with C the continue probability and V_t of
    the same state, exactly as the reference does."""
    out = torch.zeros_like(V)
    nxt = V_last
    for t in reversed(range(V.size(1))):
        ret = R[:, t] + gamma * (1.0 - C[:, t]) * ((1.0 - lam) * V[:, t] + lam * nxt)
        out[:, t] = ret
        nxt = ret
    return out


def imagine_block(ww: Weights, wa: Weights, wc: Weights, h0: Tensor, z0: Tensor,
                  expo_a: Tensor, expo_z: Tensor, gamma: float, lam: float, eps: float = 0.01,
                  run_decoder: bool = False, forced_actions: Optional[Tensor] = None,
                  forced_zidx: Optional[Tensor] = None):
    """The no-grad imagination block, lib/trainer.py:176-224.

    expo_a: (H, B, A) and expo_z: (H, B, L, K) Exp(1) draws, consumed per step in the order
    [action, z].  ``forced_actions`` (B, H) teacher-forces the sampled actions, ``forced_zidx`` (B, H, L) the
    sampled latent classes (the heads then see the forced latent; what the sampler drew from each step's own
    logits is returned as ``actions_sampled`` / ``zidx_sampled``).
    """
    H = expo_z.size(0)
    h, z = h0, z0
    hs, zs, acts, rews, conts, logps, priors, drawn_a, drawn_z = [], [], [], [], [], [], [], [], []
    for t in range(H):
        logp = actor_log_probs(wa, h, z, eps)
        a = sample_action(logp, expo_a[t])
        drawn_a.append(a)
        if forced_actions is not None:
            a = forced_actions[:, t]
        h, z, info = prior_step(ww, h, z, a, expo_z[t], eps, run_decoder)
        drawn_z.append(info["idx"])
        if forced_zidx is not None:
            z = F.one_hot(forced_zidx[:, t].long(), z.size(-1)).to(h.dtype)
            info["r_hat"] = scalar_head(ww, "rew.", h, z)
            info["c_hat"] = scalar_head(ww, "cont.", h, z)
        hs.append(h); zs.append(z); acts.append(a); logps.append(logp); priors.append(info["prior_logits"])
        rews.append(symexp(info["r_hat"].squeeze(-1)))
        conts.append(torch.sigmoid(info["c_hat"].squeeze(-1)))
    Hh = torch.stack(hs, 1); Hz = torch.stack(zs, 1)
    A = torch.stack(acts, 1); R = torch.stack(rews, 1); C = torch.stack(conts, 1)
    V = critic_value(wc, Hh, Hz)
    V_last = critic_value(wc, hs[-1], zs[-1])
    ret = lambda_return(R, C, V, V_last, gamma, lam)
    return {"h": Hh, "z": Hz, "actions": A, "rewards": R, "conts": C, "values": V, "v_last": V_last,
            "lam_returns": ret, "actor_logp": torch.stack(logps, 1), "prior_logits": torch.stack(priors, 1),
            "actions_sampled": torch.stack(drawn_a, 1), "zidx_sampled": torch.stack(drawn_z, 1)}


# --------------------------------------------------------------------------- world-model loss
def kl_loss(q_logits: Tensor, p_logits: Tensor, eps: float, free_bits: float) -> Tensor:
    """WorldModel._kl_loss, lib/world_model.py:463-475."""
    lq = log_unimix(q_logits, eps)
    lp = log_unimix(p_logits, eps)
    kl = (lq.exp() * (lq - lp)).sum(dim=-1)
    return kl.clamp_min(free_bits / float(q_logits.size(-2))).sum(dim=-1)


def prediction_loss(x, x_hat, r, r_hat, c, c_hat) -> Tensor:
    """WorldModel._prediction_loss, lib/world_model.py:477-499."""
    img = 0.5 * (x_hat - x).pow(2).mean(dim=(-3, -2, -1))
    rew = 0.5 * (r_hat - symlog(r)).pow(2).mean(dim=-1)
    cont = F.binary_cross_entropy_with_logits(c_hat, c, reduction="none").squeeze(-1)
    return img + rew + cont


def wm_loss(w: Weights, obs: Tensor, actions: Tensor, rewards: Tensor, continues: Tensor,
            expo_init: Tensor, expo_reset: Tensor, expo_z: Tensor, *, eps: float = 0.01,
            free_bits: float = 1.0, beta_pred: float = 1.0, beta_dyn: float = 1.0, beta_rep: float = 0.1):
    """WorldModel.loss, lib/world_model.py:530-603.

    expo_init (B,L,K): the init_state draw; expo_reset / expo_z (T,B,L,K): the two draws of
    every observe step in generator order [z_reset, z].
    """
    B, T = obs.size(0), obs.size(1)
    L, K = expo_z.size(-2), expo_z.size(-1)
    D = w["seq.gru.weight_hh"].size(1)
    a_prev = torch.roll(actions, 1, dims=1)
    a_prev[:, 0] = 0
    r = rewards.unsqueeze(-1)
    c = continues.float().unsqueeze(-1)
    c_prev = torch.ones(B, 1)
    h, z = init_state(B, D, L, K, expo_init)
    pred = dyn = rep = torch.zeros(())
    trace = {"h": [], "post": [], "prior": [], "idx": []}
    for t in range(T):
        h, z, info = posterior_step(w, h, z, a_prev[:, t], obs[:, t], c_prev, expo_reset[t], expo_z[t], eps)
        pred = pred + prediction_loss(obs[:, t], info["x_hat"], r[:, t], info["r_hat"], c[:, t], info["c_hat"]).mean()
        dyn = dyn + kl_loss(info["post_logits"].detach(), info["prior_logits"], eps, free_bits).mean()
        rep = rep + kl_loss(info["post_logits"], info["prior_logits"].detach(), eps, free_bits).mean()
        c_prev = c[:, t]
        trace["h"].append(h); trace["post"].append(info["post_logits"])
        trace["prior"].append(info["prior_logits"]); trace["idx"].append(info["idx"])
    pred, dyn, rep = pred / T, dyn / T, rep / T
    total = beta_pred * pred + beta_dyn * dyn + beta_rep * rep
    return total, {"pred_loss": pred, "dyn_loss": dyn, "rep_loss": rep, "h_last": h, "z_last": z,
                   "h": torch.stack(trace["h"], 1), "post_logits": torch.stack(trace["post"], 1),
                   "prior_logits": torch.stack(trace["prior"], 1), "idx": torch.stack(trace["idx"], 1)}


# --------------------------------------------------------------------------- actor / critic losses
def twohot(x_symlog: Tensor, lo_edge: float, hi_edge: float, n: int) -> Tensor:
    """Critic._twohot, lib/critic.py:103-129."""
    delta = (hi_edge - lo_edge) / (n - 1)
    x = x_symlog.clamp(min=lo_edge, max=hi_edge)
    idx = (x - lo_edge) / delta
    lo = torch.floor(idx).long()
    hi = (lo + 1).clamp(max=n - 1)
    w_hi = (idx - lo.float()).clamp(0.0, 1.0)
    w_lo = 1.0 - w_hi
    y = x.new_zeros((x.numel(), n))
    y.scatter_(1, lo.reshape(-1, 1), w_lo.reshape(-1, 1))
    y.scatter_add_(1, hi.reshape(-1, 1), w_hi.reshape(-1, 1))
    return y.view(*x.shape, n)


def critic_loss(wc: Weights, h: Tensor, z: Tensor, returns: Tensor, lo_edge=-20.0, hi_edge=20.0,
                ema_regularizer: float = 1.0) -> Tensor:
    """Critic.loss, lib/critic.py:140-161."""
    n = wc["buckets"].numel()
    y = twohot(symlog(returns), lo_edge, hi_edge, n)
    logp = F.log_softmax(critic_logits(wc, h, z), dim=-1)
    loss = -(y * logp).sum(dim=-1).mean()
    if ema_regularizer > 0.0:
        slow = F.softmax(critic_logits(wc, h, z, "slow."), dim=-1)
        loss = loss + ema_regularizer * F.kl_div(logp, slow, log_target=False, reduction="none").sum(-1).mean()
    return loss


def update_slow(wc: Weights, decay: float) -> None:
    """Critic.update_slow, lib/critic.py:131-138 (in place on the 'slow.' entries)."""
    for k in list(wc.keys()):
        if k.startswith("slow.") and k != "slow.buckets":
            wc[k].mul_(decay).add_(wc[k[len("slow."):]] * (1.0 - decay))


def percentile_scale(x: Tensor, p05: Tensor, p95: Tensor, decay: float, min_scale: float, training: bool = True):
    """EMAPercentileScale.update_get_S, lib/actor.py:16-33.  Returns (S, p05_new, p95_new)."""
    x = x.detach().reshape(-1).float()
    q05 = torch.quantile(x, 0.05)
    q95 = torch.quantile(x, 0.95)
    q95 = torch.maximum(q95, q05 + x.new_tensor(1e-8))
    if training:
        d = 1.0 - decay
        p05 = p05 * decay + d * q05
        p95 = p95 * decay + d * q95
    return torch.maximum(p95 - p05, x.new_tensor(min_scale)), p05, p95


def actor_loss(wa: Weights, h: Tensor, z: Tensor, actions: Tensor, returns: Tensor, values: Tensor,
               scale: Tensor, entropy_scale: float, eps: float = 0.01) -> Tensor:
    """Actor.loss, lib/actor.py:108-125, with the percentile scale S given."""
    logp = actor_log_probs(wa, h, z, eps)
    lp_a = logp.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    adv = (returns - values) / scale
    min_real = torch.finfo(logp.dtype).min
    ent = -(torch.clamp(logp, min=min_real) * torch.softmax(logp, dim=-1)).sum(-1)
    return -(adv * lp_a).mean() - entropy_scale * ent.mean()


# --------------------------------------------------------------------------- replay ring
class ReplayOracle:
    """numpy restatement of lib/replay_buffer.py:8-85 (storage, store, sample)."""

    def __init__(self, capacity: int, obs_shape, seq_len: int):
        self.capacity, self.seq_len = int(capacity), int(seq_len)
        self.observations = np.zeros((self.capacity, *obs_shape), dtype=np.uint8)
        self.actions = np.zeros(self.capacity, dtype=np.int64)
        self.rewards = np.zeros(self.capacity, dtype=np.float32)
        self.continues = np.zeros(self.capacity, dtype=np.bool_)
        self.idx, self.full = 0, False

    def __len__(self):
        return self.capacity if self.full else self.idx

    def store(self, obs, action, reward, cont):
        """lib/replay_buffer.py:32-44."""
        self.observations[self.idx] = np.asarray(obs, dtype=np.uint8)
        self.actions[self.idx] = int(action)
        self.rewards[self.idx] = np.float32(reward)
        self.continues[self.idx] = bool(cont)
        self.idx = (self.idx + 1) % self.capacity
        if self.idx == 0:
            self.full = True

    def draw_starts(self, batch: int) -> np.ndarray:
        """lib/replay_buffer.py:55-62: one global-numpy-RNG draw."""
        assert len(self) >= self.seq_len, "Not enough data to sample."
        if self.full:
            return np.random.randint(0, self.capacity, size=batch, dtype=np.int64)
        max_start = self.idx - self.seq_len
        if max_start < 0:
            raise RuntimeError("Not enough data to sample the requested sequence length.")
        return np.random.randint(0, max_start + 1, size=batch, dtype=np.int64)

    def gather(self, starts: np.ndarray):
        """lib/replay_buffer.py:64-72: idx_mat = (starts[:,None] + arange(T)) % capacity."""
        idx = (starts[:, None] + np.arange(self.seq_len, dtype=np.int64)[None, :]) % self.capacity
        return {"idx_mat": idx, "observations": self.observations[idx], "actions": self.actions[idx],
                "rewards": self.rewards[idx], "continues": self.continues[idx]}

    def sample(self, batch: int):
        out = self.gather(self.draw_starts(batch))
        return out
