#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference implementation.

Runs ONLY in the build container (it imports the reference's Python modules from
/root/reference; that tree does not exist on the GPU box).  The committed ``*.npz`` files are
data: inputs that cannot be regenerated from ``dreamer_amd.testing`` plus the reference's outputs.

How the reference is driven
  * ``gymnasium`` / ``tensorboardX`` are absent here; two empty stand-in modules are registered so
    that ``lib.utils`` imports (nothing from them is executed on the paths used).
  * Weights: every parameter is overwritten with ``dreamer_amd.testing.fill_module_closed_form``
    values, so tests rebuild identical weights without storing them.
  * Gumbel noise: ``torch.Tensor.exponential_`` is wrapped at Python level; while a queue of
    prepared tensors is non-empty each call copies the next one (``F.gumbel_softmax`` calls it
    from Python).  Order per imagine step: [z]; per observe step: [z_reset, z]; ``init_state``: one.
  * Action noise: ``torch.multinomial`` draws inside ATen.  The generator state is saved before
    ``dist.sample()`` and replayed afterwards to recover the Exp(1) tensor it used; the script
    asserts ``sample == argmax(probs / E)`` and stores E.

Usage:  python tests/golden/make_golden.py  [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
from dreamer_amd.testing import fill_module_closed_form, hash_exponential, hash_uniform  # noqa: E402


def _install_stubs():
    gym = types.ModuleType("gymnasium")
    gym.ObservationWrapper = type("ObservationWrapper", (), {"__init__": lambda self, env: None})
    gym.Env = type("Env", (), {})
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Discrete = type("Discrete", (), {})
    spaces.Box = type("Box", (), {})
    gym.spaces = spaces
    gym.wrappers = types.ModuleType("gymnasium.wrappers")
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = type("SummaryWriter", (), {})
    for name, mod in (("gymnasium", gym), ("gymnasium.spaces", spaces),
                      ("gymnasium.wrappers", gym.wrappers), ("tensorboardX", tbx)):
        sys.modules.setdefault(name, mod)


REFERENCE = "/root/reference"


def _bind_reference_lib():
    """Make ``lib`` mean the reference's package, whatever else is importable under that name.

    The reference's ``lib/`` has no ``__init__.py`` (a namespace package); this repository ships a regular
    package of the same name that aliases ``dreamer_amd``, and regular packages win over namespace packages
    regardless of ``sys.path`` order.  So the name is bound explicitly: a module object whose ``__path__`` holds
    only the reference's directory, registered before anything imports ``lib.*``."""
    for name in [n for n in sys.modules if n == "lib" or n.startswith("lib.")]:
        del sys.modules[name]
    pkg = types.ModuleType("lib")
    pkg.__path__ = [os.path.join(REFERENCE, "lib")]
    pkg.__package__ = "lib"
    sys.modules["lib"] = pkg


_install_stubs()
_bind_reference_lib()
from lib.actor import Actor  # noqa: E402
from lib.critic import Critic  # noqa: E402
from lib.replay_buffer import ReplayBuffer  # noqa: E402
from lib.utils import log_unimix, symexp, symlog  # noqa: E402
from lib.world_model import WorldModel, WorldModelState  # noqa: E402

for _cls in (Actor, Critic, ReplayBuffer, WorldModel):
    _file = sys.modules[_cls.__module__].__file__
    assert _file.startswith(REFERENCE + os.sep), f"{_cls.__name__} was imported from {_file}, not from the reference"
assert sys.modules[log_unimix.__module__].__file__.startswith(REFERENCE + os.sep)

# ----------------------------------------------------------------------------- noise injection
_ORIG_EXPONENTIAL = torch.Tensor.exponential_
_QUEUE = []


def _patched_exponential(self, *args, **kwargs):
    if _QUEUE:
        src = _QUEUE.pop(0)
        assert tuple(src.shape) == tuple(self.shape), (src.shape, self.shape)
        return self.copy_(src)
    return _ORIG_EXPONENTIAL(self, *args, **kwargs)


torch.Tensor.exponential_ = _patched_exponential


def expo(name, shape, salt=0):
    return torch.from_numpy(hash_exponential(name, shape, salt))


def sample_with_noise(dist):
    """dist.sample() plus the Exp(1) tensor torch.multinomial consumed for it."""
    before = torch.get_rng_state()
    a = dist.sample()
    after = torch.get_rng_state()
    torch.set_rng_state(before)
    e = _ORIG_EXPONENTIAL(torch.empty_like(dist.probs))
    assert torch.equal(torch.get_rng_state(), after)
    assert torch.equal(torch.argmax(dist.probs / e, dim=-1), a)
    return a, e


# ----------------------------------------------------------------------------- configurations
DIMS = {
    # kernel-compatible reduced dims: 32 classes per latent, widths multiple of 32
    "small": dict(obs_shape=(3, 16, 16), A=4, L=8, K=32, D=64, M=64, C=4, B=4, H=5, T=6),
    "default": dict(obs_shape=(3, 64, 64), A=6, L=32, K=32, D=512, M=512, C=32, B=16, H=15, T=3),
    # the scaled model: recurrent width 4096 with the full 32 x 32 latent (hidden width of the MLPs 512), 16 sequences;
    # short horizon / sequence so that the stored outputs stay small
    "c5": dict(obs_shape=(3, 64, 64), A=6, L=32, K=32, D=4096, M=512, C=32, B=16, H=3, T=3),
}


def build_models(d, salt):
    wm = WorldModel(d["obs_shape"], d["A"], num_latents=d["L"], classes_per_latent=d["K"],
                    hidden_size=d["D"], base_cnn_channels=d["C"], mlp_hidden_units=d["M"],
                    free_bits=1.0, beta_pred=1.0, beta_dyn=1.0, beta_rep=0.1, unimix_eps=0.01)
    S = d["D"] + d["L"] * d["K"]
    actor = Actor(S, d["A"], d["M"], 2, entropy_scale=3e-4, ret_norm_limit=1.0, ret_norm_decay=0.99,
                  unimix_eps=0.01)
    critic = Critic(S, d["M"], 2, 255, -20.0, 20.0, 0.98, 1.0)
    fill_module_closed_form(wm, salt=salt)
    fill_module_closed_form(actor, salt=salt + 1)
    fill_module_closed_form(critic, salt=salt + 2, skip_prefixes=("slow.",))
    fill_module_closed_form(critic.slow, salt=salt + 3)
    return wm.train(), actor.train(), critic.train()


def start_state(d, salt):
    B, D, L, K = d["B"], d["D"], d["L"], d["K"]
    h0 = torch.from_numpy(np.tanh(3.0 * (hash_uniform("h0", (B, D), salt) - 0.5)).astype(np.float32))
    idx0 = torch.from_numpy((hash_uniform("z0", (B, L), salt) * K).astype(np.int64))
    z0 = torch.zeros(B, L, K).scatter_(-1, idx0.unsqueeze(-1), 1.0)
    return h0, z0, idx0


@torch.no_grad()
def golden_imagine(tag, out_dir, salt, min_margin=None):
    d = DIMS[tag]
    wm, actor, critic = build_models(d, salt)
    B, H, L, K, A = d["B"], d["H"], d["L"], d["K"], d["A"]
    h0, z0, _ = start_state(d, salt)
    expo_z = expo("imagine_z", (H, B, L, K), salt)
    gamma, lam = 0.997, 0.95
    torch.manual_seed(1000 + salt)
    s = WorldModelState(h=h0, z=z0)
    hs, zs, acts, rews, conts, noise_a, logps, priors, rhat, chat = [], [], [], [], [], [], [], [], [], []
    for t in range(H):
        dist = actor(s)
        a, e = sample_with_noise(dist)
        _QUEUE.append(expo_z[t])
        s, info = wm.step(s, a_prev_idx=a)
        assert not _QUEUE
        hs.append(s.h); zs.append(s.z); acts.append(a); noise_a.append(e); logps.append(dist.logits)
        priors.append(info["prior_logits"]); rhat.append(info["r_hat"]); chat.append(info["c_hat"])
        rews.append(symexp(info["r_hat"].squeeze(-1)))
        conts.append(torch.sigmoid(info["c_hat"].squeeze(-1)))
    Hs = WorldModelState(h=torch.stack(hs, 1), z=torch.stack(zs, 1))
    Aact = torch.stack(acts, 1); R = torch.stack(rews, 1); C = torch.stack(conts, 1)
    V = critic.value(Hs)
    V_last = critic.value(WorldModelState(h=hs[-1], z=zs[-1]))
    ret = torch.zeros_like(V)
    nxt = V_last
    for t in reversed(range(H)):
        cur = R[:, t] + gamma * (1.0 - C[:, t]) * ((1.0 - lam) * V[:, t] + lam * nxt)
        ret[:, t] = cur
        nxt = cur
    out = dict(salt=salt, gamma=gamma, lam=lam,
               noise_a=torch.stack(noise_a, 0).numpy(), h=Hs.h.numpy(),
               zidx=Hs.z.argmax(-1).numpy().astype(np.uint8), actions=Aact.numpy(),
               rewards=R.numpy(), conts=C.numpy(), values=V.numpy(), v_last=V_last.numpy(),
               lam_returns=ret.numpy(), actor_logp=torch.stack(logps, 1).numpy(),
               r_hat=torch.stack(rhat, 1).squeeze(-1).numpy(), c_hat=torch.stack(chat, 1).squeeze(-1).numpy())
    assert float((Hs.z.sum(-1) - 1).abs().max()) == 0.0 and float((Hs.z * (1 - Hs.z)).abs().max()) == 0.0
    # margins of every categorical draw: tests only demand index equality where the margin is clear
    zm, am = [], []
    s = WorldModelState(h=h0, z=z0)
    for t in range(H):
        lp = log_unimix(priors[t], 0.01) - torch.log(expo_z[t])
        top2 = lp.topk(2, dim=-1).values
        zm.append(top2[..., 0] - top2[..., 1])
        sc = (logps[t].softmax(-1) / noise_a[t]).log()
        t2 = sc.topk(2, dim=-1).values
        am.append(t2[..., 0] - t2[..., 1])
    out["z_margin"] = torch.stack(zm, 1).numpy()
    out["a_margin"] = torch.stack(am, 1).numpy()
    if tag == "small":
        out["prior_logits"] = torch.stack(priors, 1).numpy()

    # actor / critic losses on the imagined batch (lib/trainer.py:227-247)
    closs = critic.loss(model_states=Hs, returns=ret)
    aloss = actor.loss(model_states=Hs, actions=Aact, returns=ret, values=V)
    out.update(critic_loss=closs.numpy(), actor_loss=aloss.numpy(),
               p05=actor.ret_scale.p05.numpy().copy(), p95=actor.ret_scale.p95.numpy().copy())
    critic.update_slow()
    out["slow_head_bias"] = critic.slow.head.bias.numpy().copy()
    out["slow_mlp0_row0"] = critic.slow.mlp[0].weight[0].numpy().copy()
    print(f"imagine_{tag} salt {salt}: min z margin {out['z_margin'].min():.3e}, "
          f"min a margin {out['a_margin'].min():.3e}")
    if min_margin is not None and min(out["z_margin"].min(), out["a_margin"].min()) < min_margin:
        return False
    np.savez_compressed(os.path.join(out_dir, f"imagine_{tag}.npz"), **out)
    return True


def golden_observe(tag, out_dir, salt, B=None, T=None, name=None, compact=False, min_margin=None):
    """World-model loss + backward over a (B, T) batch.  ``compact``: outputs only (losses, last state, every sampled
    index, the recurrent state every 8th step, gradient norms and slices) for the full-size batches, whose
    per-step tensors would be tens of MB."""
    d = DIMS[tag]
    wm, _, _ = build_models(d, salt)
    B = B if B is not None else (d["B"] if tag == "small" else 2)
    T = T if T is not None else d["T"]
    L, K, A = d["L"], d["K"], d["A"]
    name = name or f"observe_{tag}"
    obs_u8 = (hash_uniform("obs", (B, T, *d["obs_shape"]), salt) * 256).astype(np.uint8)
    actions = (hash_uniform("act", (B, T), salt) * A).astype(np.int64)
    rewards = ((hash_uniform("rew", (B, T), salt) - 0.5) * 4).astype(np.float32)
    continues = hash_uniform("cont", (B, T), salt) > 0.25
    e_init = expo("obs_init", (B, L, K), salt)
    e_reset = expo("obs_reset", (T, B, L, K), salt)
    e_z = expo("obs_z", (T, B, L, K), salt)
    _QUEUE.append(e_init)
    for t in range(T):
        _QUEUE.extend([e_reset[t], e_z[t]])
    obs = torch.from_numpy(obs_u8).float() / 255.0
    # record per-step tensors through a forward hook on step()
    trace = {"h": [], "post": [], "prior": [], "idx": []}
    orig_step = wm.step

    def traced_step(*a, **k):
        st, info = orig_step(*a, **k)
        trace["h"].append(st.h.detach()); trace["post"].append(info["post_logits"].detach())
        trace["prior"].append(info["prior_logits"].detach()); trace["idx"].append(st.z.detach().argmax(-1))
        return st, info

    wm.step = traced_step
    loss, info = wm.loss(obs, torch.from_numpy(actions), torch.from_numpy(rewards), torch.from_numpy(continues))
    assert not _QUEUE
    # margin of every posterior draw (the reset draws are argmin(E): exact on any hardware)
    post = torch.stack(trace["post"], 1)
    lp = log_unimix(post, 0.01) - torch.log(e_z.permute(1, 0, 2, 3))
    top2 = lp.topk(2, dim=-1).values
    margin = float((top2[..., 0] - top2[..., 1]).min())
    print(f"{name} salt {salt}: min z margin {margin:.3e}")
    if min_margin is not None and margin < min_margin:
        return False
    wm.zero_grad()
    loss.backward()
    out = dict(salt=salt, B=B, T=T, total=loss.detach().numpy(), pred=info["pred_loss"].detach().numpy(),
               dyn=info["dyn_loss"].detach().numpy(), rep=info["rep_loss"].detach().numpy(),
               zidx=torch.stack(trace["idx"], 1).numpy().astype(np.uint8),
               h_last=info["state"].h.detach().numpy(), z_margin=np.float64(margin))
    if compact and T < 8:
        out.update(h=torch.stack(trace["h"], 1).numpy(), post_last=post[:, -1].numpy(), prior_last=trace["prior"][-1].numpy())
    elif compact:
        out.update(h_every8=torch.stack(trace["h"], 1)[:, 7::8].numpy(), post_last=post[:, -1].numpy(),
                   prior_last=trace["prior"][-1].numpy())
    else:
        out.update(h=torch.stack(trace["h"], 1).numpy(), post_logits=post.numpy(),
                   prior_logits=torch.stack(trace["prior"], 1).numpy())
        del out["T"], out["z_margin"]      # the first fixtures were written without them
    gn = {}
    for k, p in wm.named_parameters():
        gn[k] = float(p.grad.double().norm()) if p.grad is not None else -1.0  # fp64: an fp32 sum over 1e7 squares drifts
    out["grad_names"] = np.array(list(gn.keys()))
    out["grad_norms"] = np.array(list(gn.values()), dtype=np.float64)
    for k in ("seq.gru.weight_hh", "seq.linear.weight", "enc.fc.0.weight", "dyn.head.bias", "seq.a_emb.weight"):
        g = dict(wm.named_parameters())[k].grad
        out["grad:" + k] = g.reshape(-1)[:256].numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **out)
    print(f"{name}: total {float(loss.detach()):.6f}")
    return True


def golden_replay(out_dir):
    out = {}
    obs_shape = (3, 8, 8)
    fbytes = int(np.prod(obs_shape))

    def frame(i):
        return ((np.arange(fbytes, dtype=np.int64) * 7 + i * 13) % 251).astype(np.uint8).reshape(obs_shape)

    cases = {"notfull": (50, 37, 5), "exactfull": (40, 40, 8), "wrapped": (32, 75, 6), "seqcap": (16, 16, 16)}
    for name, (cap, nstore, T) in cases.items():
        buf = ReplayBuffer(cap, obs_shape, T, torch.device("cpu"))
        for i in range(nstore):
            buf.store(frame(i), i % 5, float(i) * 0.25 - 3.0, (i % 7) != 3)
        for seed in (0, 42):
            np.random.seed(seed)
            state = np.random.get_state()
            batch = buf.sample(6)
            np.random.set_state(state)
            if buf.full:
                starts = np.random.randint(0, cap, size=6, dtype=np.int64)
            else:
                starts = np.random.randint(0, buf.idx - T + 1, size=6, dtype=np.int64)
            key = f"{name}_s{seed}"
            out[key + "_starts"] = starts
            out[key + "_obs"] = batch["observations"].numpy()
            out[key + "_act"] = batch["actions"].numpy()
            out[key + "_rew"] = batch["rewards"].numpy()
            out[key + "_cont"] = batch["continues"].numpy()
        out[name + "_meta"] = np.array([cap, nstore, T, buf.idx, int(buf.full), len(buf)], dtype=np.int64)
    # the default-size start stream (capacity 100000, full, B=16, seed 42)
    np.random.seed(42)
    out["default_full_starts"] = np.random.randint(0, 100000, size=16, dtype=np.int64)
    np.savez_compressed(os.path.join(out_dir, "replay.npz"), **out)
    print("replay: default starts", out["default_full_starts"][:6])


@torch.no_grad()
def golden_pointwise(out_dir):
    x = torch.from_numpy(np.concatenate([np.linspace(-30, 30, 241), [0.0, 1e-8, -1e-8, 1e4, -1e4]]).astype(np.float32))
    logits = torch.from_numpy(((hash_uniform("pw_logits", (64, 32), 7) - 0.5) * 12).astype(np.float32))
    logits[0] = 0.0
    logits[1, 3] = 40.0
    logits[2] = -50.0
    e = expo("pw_noise", (64, 32), 7)
    import torch.nn.functional as F
    lp = log_unimix(logits, 0.01)
    orig = _ORIG_EXPONENTIAL
    _QUEUE.append(e)
    z = F.gumbel_softmax(lp, tau=1, hard=True, dim=-1)
    critic = Critic(8, 8, 1)
    rets = torch.from_numpy(np.array([-1e9, -4.85e8, -100.0, -1.0, -0.3, 0.0, 1e-3, 0.5, 1.0, 7.7, 1e3, 4.85e8, 1e9],
                                     dtype=np.float32))
    R = torch.from_numpy(((hash_uniform("lr_R", (7, 15), 3) - 0.5) * 4).astype(np.float32))
    C = torch.from_numpy(hash_uniform("lr_C", (7, 15), 3).astype(np.float32))
    V = torch.from_numpy(((hash_uniform("lr_V", (7, 15), 3) - 0.5) * 20).astype(np.float32))
    VL = torch.from_numpy(((hash_uniform("lr_VL", (7,), 3) - 0.5) * 20).astype(np.float32))
    ret = torch.zeros_like(V)
    nxt = VL
    for t in reversed(range(15)):
        cur = R[:, t] + 0.997 * (1.0 - C[:, t]) * ((1.0 - 0.95) * V[:, t] + 0.95 * nxt)
        ret[:, t] = cur
        nxt = cur
    np.savez_compressed(os.path.join(out_dir, "pointwise.npz"), x=x.numpy(), symlog=symlog(x).numpy(),
                        symexp=symexp(x.clamp(-30, 30)).numpy(), logits=logits.numpy(), log_unimix=lp.numpy(),
                        noise=e.numpy(), z=z.numpy(), twohot_in=rets.numpy(),
                        twohot=critic._twohot(symlog(rets)).numpy(), lr_R=R.numpy(), lr_C=C.numpy(),
                        lr_V=V.numpy(), lr_VL=VL.numpy(), lr_ret=ret.numpy())
    print("pointwise done")


def golden_state_keys(out_dir):
    """Names and shapes of every state-dict entry of the reference modules at default sizes."""
    import json
    wm, actor, critic = build_models(DIMS["default"], 0)
    doc = {name: {k: list(v.shape) for k, v in mod.state_dict().items()}
           for name, mod in (("world_model", wm), ("actor", actor), ("critic", critic))}
    doc["param_order"] = {name: [k for k, _ in mod.named_parameters()]
                          for name, mod in (("world_model", wm), ("actor", actor), ("critic", critic))}
    with open(os.path.join(out_dir, "state_keys.json"), "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only-c5", action="store_true", help="write the scaled-model fixtures only")
    args = ap.parse_args()
    torch.set_num_threads(8)
    if args.only_c5:
        return golden_c5(args.out)
    golden_state_keys(args.out)
    golden_pointwise(args.out)
    golden_replay(args.out)
    golden_imagine("small", args.out, salt=11)
    golden_observe("small", args.out, salt=11)
    # A free-running rollout only reproduces on other hardware when no categorical draw is a
    # near-tie, so take the first salt whose smallest draw margin is clear of fp32 noise.
    salt = 21
    while not golden_imagine("default", args.out, salt=salt, min_margin=1.5e-4):
        salt += 1
    golden_observe("default", args.out, salt=21)
    # the per-GPU training batch (16 sequences x 64 steps, lib/config.py:18-19) and the 50-step variant, default
    # sizes, outputs only; the first salt whose smallest posterior-draw margin is clear of fp32 noise
    for name, T, salt in (("observe_full", 64, 31), ("observe_t50", 50, 41)):
        while not golden_observe("default", args.out, salt=salt, B=16, T=T, name=name, compact=True, min_margin=1.0e-4):
            salt += 1
    golden_c5(args.out)


def golden_c5(out_dir):
    """The scaled model (recurrent width 4096, full 32 x 32 latent) at 16 sequences: a 3-step imagination block and the
    world-model loss + backward over 16 x 3 frames, outputs only."""
    salt = 51
    while not golden_imagine("c5", out_dir, salt=salt, min_margin=1.5e-4):
        salt += 1
    salt = 61
    while not golden_observe("c5", out_dir, salt=salt, B=16, T=3, name="observe_c5", compact=True, min_margin=1.0e-4):
        salt += 1


if __name__ == "__main__":
    main()
