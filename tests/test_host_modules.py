"""Host-side mirror of the reference surface (CPU only): state-dict names, the hoisted /
batched world-model loss against the reference's per-step loss, actor / critic losses."""
import json
import os

import numpy as np
import pytest
import torch

from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.world_model import WorldModelState
from helpers import DIMS, GOLDEN, build_models, load, observe_inputs, rel_err

TOL = 2e-5


def test_state_dict_names_shapes_and_order():
    with open(os.path.join(GOLDEN, "state_keys.json")) as f:
        ref = json.load(f)
    wm, actor, critic = build_models(DIMS["default"], 0)
    for name, mod in (("world_model", wm), ("actor", actor), ("critic", critic)):
        mine = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert mine == ref[name], name
        assert [k for k, _ in mod.named_parameters()] == ref["param_order"][name], name


def test_config_surface():
    cfg = Config()
    assert (cfg.batch_size, cfg.batch_length, cfg.imagination_horizon) == (16, 64, 15)
    assert (cfg.gamma, cfg.lam, cfg.beta_dyn, cfg.beta_rep) == (0.997, 0.95, 1.0, 0.1)
    assert Config.num_latents == 32 and Config.classes_per_latent == 32
    with pytest.raises(TypeError):
        Config(num_latents=8)  # not a dataclass field in the reference either


@pytest.mark.parametrize("tag", ["small", "default"])
def test_world_model_loss_matches_reference(tag):
    g = load(f"observe_{tag}.npz")
    d, salt, B = DIMS[tag], int(g["salt"]), int(g["B"])
    wm, _, _ = build_models(d, salt)
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
    total, info = wm.loss(obs.float() / 255.0, actions, rewards, continues, noise=noise)
    assert rel_err(info["h"].detach(), g["h"]) < TOL
    assert rel_err(info["post_logits"].detach(), g["post_logits"]) < TOL
    assert rel_err(info["prior_logits"].detach(), g["prior_logits"]) < TOL
    assert rel_err(info["state"].h.detach(), g["h_last"]) < TOL
    assert np.array_equal(info["state"].z.detach().argmax(-1).numpy().astype(np.uint8), g["zidx"][:, -1])
    assert rel_err(total.detach(), g["total"]) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(info[key + "_loss"].detach(), g[key]) < TOL, key
    total.backward()
    params = dict(wm.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        grad = params[str(name)].grad
        if ref < 0:
            assert grad is None, name
        else:
            assert abs(float(grad.double().norm()) - ref) <= 5e-5 * max(ref, 1e-3), name
    for key in g.files:
        if key.startswith("grad:"):
            assert rel_err(params[key[5:]].grad.reshape(-1)[:256], g[key]) < 1e-4, key


@pytest.mark.parametrize("tag", ["small", "default"])
def test_actor_critic_losses_match_reference(tag):
    g = load(f"imagine_{tag}.npz")
    d, salt = DIMS[tag], int(g["salt"])
    _, actor, critic = build_models(d, salt)
    zidx = torch.from_numpy(g["zidx"].astype(np.int64))
    Hs = WorldModelState(h=torch.from_numpy(g["h"]), z=torch.nn.functional.one_hot(zidx, d["K"]).float())
    ret, V, A = (torch.from_numpy(g[k]) for k in ("lam_returns", "values", "actions"))
    assert rel_err(critic.value(Hs).detach(), g["values"]) < TOL
    # actor_logp[:, t] is the policy at the state BEFORE imagined step t, i.e. at Hs[:, t-1]
    assert rel_err(actor(Hs).logits.detach()[:, :-1], g["actor_logp"][:, 1:]) < TOL
    assert rel_err(critic.loss(Hs, ret).detach(), g["critic_loss"]) < TOL
    assert rel_err(actor.loss(Hs, A, ret, V).detach(), g["actor_loss"]) < TOL
    assert rel_err(actor.ret_scale.p05, g["p05"]) < TOL and rel_err(actor.ret_scale.p95, g["p95"]) < TOL
    critic.update_slow()
    assert rel_err(critic.slow.head.bias.detach(), g["slow_head_bias"]) < 1e-6
    assert rel_err(critic.slow.mlp[0].weight[0].detach(), g["slow_mlp0_row0"]) < 1e-6


@pytest.mark.parametrize("case", ["notfull", "exactfull", "wrapped", "seqcap"])
def test_replay_buffer_host_device(case):
    g = load("replay.npz")
    cap, nstore, T, idx, full, length = (int(v) for v in g[case + "_meta"])
    obs_shape = (3, 8, 8)
    fbytes = int(np.prod(obs_shape))
    buf = ReplayBuffer(cap, obs_shape, T, torch.device("cpu"))
    for i in range(nstore):
        frame = ((np.arange(fbytes, dtype=np.int64) * 7 + i * 13) % 251).astype(np.uint8).reshape(obs_shape)
        buf.store(frame, i % 5, float(i) * 0.25 - 3.0, (i % 7) != 3)
    assert (buf.idx, int(buf.full), len(buf)) == (idx, full, length)
    for seed in (0, 42):
        np.random.seed(seed)
        batch = buf.sample(6)
        key = f"{case}_s{seed}"
        assert np.array_equal(batch["observations"].numpy(), g[key + "_obs"])
        assert np.array_equal(batch["actions"].numpy(), g[key + "_act"])
        assert np.array_equal(batch["rewards"].numpy(), g[key + "_rew"])
        assert np.array_equal(batch["continues"].numpy(), g[key + "_cont"])


def test_replay_buffer_errors_and_bulk_store():
    buf = ReplayBuffer(10, (1, 4, 4), 4, torch.device("cpu"))
    with pytest.raises(AssertionError):
        buf.sample(2)
    one = ReplayBuffer(10, (1, 4, 4), 4, torch.device("cpu"))
    frames = (np.arange(13 * 16) % 256).astype(np.uint8).reshape(13, 1, 4, 4)
    for i in range(13):
        one.store(frames[i], i, i * 0.5, i % 3 != 0)
    buf.store_many(frames, np.arange(13), np.arange(13) * 0.5, np.arange(13) % 3 != 0)
    assert (buf.idx, buf.full) == (one.idx, one.full)
    for name in ("observations", "actions", "rewards", "continues"):
        assert torch.equal(getattr(buf, name), getattr(one, name))


@pytest.mark.parametrize("with_cont", [True, False])
def test_act_step_is_the_state_of_step(with_cont):
    """The acting shortcut returns the posterior state of ``step`` (lib/world_model.py:415-457) and leaves
    the generator where ``step`` leaves it (reset draw, then sample draw)."""
    from dreamer_amd.world_model import WorldModel
    torch.manual_seed(3)
    wm = WorldModel((3, 64, 64), 6, hidden_size=64, base_cnn_channels=4, mlp_hidden_units=32, num_latents=8,
                    classes_per_latent=8)
    B = 3
    state = wm.init_state(B)
    state.h.normal_()
    x = torch.rand(B, 3, 64, 64)
    a = torch.tensor([1, 5, 0])
    c = torch.tensor([[1.0], [0.0], [1.0]]) if with_cont else None
    with torch.no_grad():
        torch.manual_seed(11)
        ref, _ = wm.step(state, a, x, c)
        after_ref = torch.rand(4)
        torch.manual_seed(11)
        got = wm.act_step(state, a, x, c)
        after_got = torch.rand(4)
    torch.testing.assert_close(got.h, ref.h, rtol=1e-5, atol=1e-6)
    assert torch.equal(got.z, ref.z)
    assert torch.equal(after_ref, after_got)


def test_precision_names_resolve_to_library_codes(monkeypatch):
    """``"f32"`` leaves the form of the fp32-accurate products to the engine: the split form from 256 trajectories on
    (any width), fp32 matrix instructions for fewer; explicit names are taken as given."""
    from dreamer_amd.imagine import resolve_precision
    monkeypatch.delenv("DRM_F32_MFMA", raising=False)
    assert resolve_precision("f32", 16, 512, 512) == 0
    assert resolve_precision("f32", 255, 512, 512) == 0
    assert resolve_precision("f32", 256, 512, 512) == 2
    assert resolve_precision("f32", 4096, 512, 512) == 2
    assert resolve_precision("f32", 4096, 4096, 512) == 2          # wide recurrent state: the split form as well
    assert resolve_precision("f32", 4096, 512, 1024) == 2          # wide hidden layers
    assert resolve_precision("f32", 64, 4096, 512) == 0
    assert resolve_precision("f32_mfma", 4096, 512, 512) == 0
    assert resolve_precision("f32_split", 16, 512, 512) == 2
    assert resolve_precision("bf16", 16, 4096, 512) == 1
    monkeypatch.setenv("DRM_F32_MFMA", "1")
    assert resolve_precision("f32", 4096, 512, 512) == 0
    with pytest.raises(ValueError):
        resolve_precision("fp16", 16, 512, 512)


def test_towers_take_channels_last_only_with_natively_convolved_channel_counts():
    """Channels-last activations are handed out only when every convolution apart from the image-side one runs on the
    library's own kernels; reduced models (4-32 channels) keep NCHW tensors for the backend's routines."""
    from types import SimpleNamespace

    from dreamer_amd import world_model as W

    def probe(n, shape):
        return SimpleNamespace(is_cuda=True, dtype=torch.float32, dim=lambda: 4, shape=(n,) + shape)

    enc32 = W.Encoder((3, 64, 64), base_cnn_channels=32).conv
    dec32 = W.Decoder((3, 64, 64), base_cnn_channels=32, num_latents=8, hidden_size=64).deconv
    enc4 = W.Encoder((3, 16, 16), base_cnn_channels=4).conv
    dec8 = W.Decoder((3, 64, 64), base_cnn_channels=8, num_latents=8, hidden_size=64).deconv
    assert enc32._inner_convs_native(probe(1024, (3, 64, 64)))
    assert dec32._inner_convs_native(probe(1024, (256, 4, 4)))
    assert not enc4._inner_convs_native(probe(84, (3, 16, 16)))
    assert not dec8._inner_convs_native(probe(70, (64, 4, 4)))
    assert not enc32._inner_convs_native(probe(8, (3, 64, 64)))      # below the native kernels' image count
