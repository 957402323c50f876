"""Shared builders for the tests: models at the golden configurations with closed-form weights,
regenerated inputs / noise, and conversions to oracle weight dicts."""
import os

import numpy as np
import torch

from dreamer_amd.actor import Actor
from dreamer_amd.critic import Critic
from dreamer_amd.testing import fill_module_closed_form, hash_exponential, hash_uniform
from dreamer_amd.world_model import WorldModel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# must match tests/golden/make_golden.py
DIMS = {
    "small": dict(obs_shape=(3, 16, 16), A=4, L=8, K=32, D=64, M=64, C=4, B=4, H=5, T=6),
    "default": dict(obs_shape=(3, 64, 64), A=6, L=32, K=32, D=512, M=512, C=32, B=16, H=15, T=3),
    "c5": dict(obs_shape=(3, 64, 64), A=6, L=32, K=32, D=4096, M=512, C=32, B=16, H=3, T=3),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def build_models(d, salt, actor_layers=2, critic_layers=2):
    wm = WorldModel(d["obs_shape"], d["A"], num_latents=d["L"], classes_per_latent=d["K"], hidden_size=d["D"],
                    base_cnn_channels=d["C"], mlp_hidden_units=d["M"], free_bits=1.0, beta_pred=1.0, beta_dyn=1.0,
                    beta_rep=0.1, unimix_eps=0.01)
    S = d["D"] + d["L"] * d["K"]
    actor = Actor(S, d["A"], d["M"], actor_layers, entropy_scale=3e-4, ret_norm_limit=1.0, ret_norm_decay=0.99,
                  unimix_eps=0.01)
    critic = Critic(S, d["M"], critic_layers, 255, -20.0, 20.0, 0.98, 1.0)
    fill_module_closed_form(wm, salt=salt)
    fill_module_closed_form(actor, salt=salt + 1)
    fill_module_closed_form(critic, salt=salt + 2, skip_prefixes=("slow.",))
    fill_module_closed_form(critic.slow, salt=salt + 3)
    return wm.train(), actor.train(), critic.train()


def weights_of(module):
    """Oracle weight dict (detached CPU fp32 copies keyed by state-dict name)."""
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def start_state(d, salt, B=None):
    B = B or d["B"]
    D, L, K = d["D"], d["L"], d["K"]
    h0 = torch.from_numpy(np.tanh(3.0 * (hash_uniform("h0", (B, D), salt) - 0.5)).astype(np.float32))
    idx0 = torch.from_numpy((hash_uniform("z0", (B, L), salt) * K).astype(np.int64))
    z0 = torch.zeros(B, L, K).scatter_(-1, idx0.unsqueeze(-1), 1.0)
    return h0, z0, idx0


def expo(name, shape, salt=0):
    return torch.from_numpy(hash_exponential(name, shape, salt))


def observe_inputs(d, salt, B, T=None):
    T, L, K, A = (T or d["T"]), d["L"], d["K"], d["A"]
    obs_u8 = (hash_uniform("obs", (B, T, *d["obs_shape"]), salt) * 256).astype(np.uint8)
    actions = (hash_uniform("act", (B, T), salt) * A).astype(np.int64)
    rewards = ((hash_uniform("rew", (B, T), salt) - 0.5) * 4).astype(np.float32)
    continues = hash_uniform("cont", (B, T), salt) > 0.25
    noise = {"init": expo("obs_init", (B, L, K), salt), "reset": expo("obs_reset", (T, B, L, K), salt),
             "z": expo("obs_z", (T, B, L, K), salt)}
    return (torch.from_numpy(obs_u8), torch.from_numpy(actions), torch.from_numpy(rewards),
            torch.from_numpy(continues), noise)


def rel_err(a, b):
    """Norm-wise error: largest absolute difference over the largest reference magnitude.  Entries much smaller
    than the tensor's largest are only loosely bound by it; ``assert_close_elementwise`` is the per-entry check."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close_elementwise(a, b, rtol=1e-5, atol_scale=1e-6, what=""):
    """Every entry: |a - b| <= rtol * |b| + atol_scale * (root-mean-square of b).  The absolute term is the
    rounding floor of an fp32 sum whose terms have the tensor's typical magnitude (an entry that is a near-cancelling
    sum of O(rms) terms cannot be reproduced to a fraction of itself); it is 10 x smaller than the norm-wise bound
    relative to the rms and far smaller relative to the max."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(b.pow(2).mean().sqrt().clamp_min(1e-30))
    excess = (a - b).abs() - (rtol * b.abs() + atol_scale * scale)
    worst = int(excess.argmax())
    assert float(excess.max()) <= 0.0, (what, "entry", worst, float(a.reshape(-1)[worst]), float(b.reshape(-1)[worst]),
                                         "rms", scale)
