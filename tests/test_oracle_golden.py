"""The CPU oracle against the golden vectors captured from the reference (CPU only)."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from helpers import DIMS, build_models, expo, load, observe_inputs, rel_err, start_state, weights_of

TOL = 2e-5  # fp32, relative to the tensor's max magnitude (reduction order differs between restatement and reference)


def test_pointwise_tables():
    g = load("pointwise.npz")
    x = torch.from_numpy(g["x"])
    assert torch.equal(O.symlog(x), torch.from_numpy(g["symlog"]))
    assert torch.equal(O.symexp(x.clamp(-30, 30)), torch.from_numpy(g["symexp"]))
    lp = O.log_unimix(torch.from_numpy(g["logits"]), 0.01)
    assert torch.equal(lp, torch.from_numpy(g["log_unimix"]))
    z, idx, _ = O.gumbel_hard(lp, torch.from_numpy(g["noise"]))
    assert torch.equal(z, torch.from_numpy(g["z"]))
    th = O.twohot(O.symlog(torch.from_numpy(g["twohot_in"])), -20.0, 20.0, 255)
    assert torch.equal(th, torch.from_numpy(g["twohot"]))
    ret = O.lambda_return(*(torch.from_numpy(g[k]) for k in ("lr_R", "lr_C", "lr_V", "lr_VL")), 0.997, 0.95)
    assert torch.equal(ret, torch.from_numpy(g["lr_ret"]))


@pytest.mark.parametrize("case", ["notfull", "exactfull", "wrapped", "seqcap"])
def test_replay_bit_exact(case):
    g = load("replay.npz")
    cap, nstore, T, idx, full, length = (int(v) for v in g[case + "_meta"])
    obs_shape = (3, 8, 8)
    fbytes = int(np.prod(obs_shape))
    buf = O.ReplayOracle(cap, obs_shape, T)
    for i in range(nstore):
        frame = ((np.arange(fbytes, dtype=np.int64) * 7 + i * 13) % 251).astype(np.uint8).reshape(obs_shape)
        buf.store(frame, i % 5, float(i) * 0.25 - 3.0, (i % 7) != 3)
    assert (buf.idx, int(buf.full), len(buf)) == (idx, full, length)
    for seed in (0, 42):
        np.random.seed(seed)
        starts = buf.draw_starts(6)
        key = f"{case}_s{seed}"
        assert np.array_equal(starts, g[key + "_starts"])
        out = buf.gather(starts)
        assert np.array_equal(out["observations"], g[key + "_obs"])
        assert np.array_equal(out["actions"], g[key + "_act"])
        assert np.array_equal(out["rewards"], g[key + "_rew"])
        assert np.array_equal(out["continues"], g[key + "_cont"])


def test_replay_default_start_stream():
    g = load("replay.npz")
    buf = O.ReplayOracle(100000, (1, 1, 16), 64)
    buf.full = True
    np.random.seed(42)
    assert np.array_equal(buf.draw_starts(16), g["default_full_starts"])


def test_replay_errors():
    buf = O.ReplayOracle(10, (1, 1, 16), 4)
    with pytest.raises(AssertionError):
        buf.draw_starts(2)


@pytest.mark.parametrize("tag", ["small", "default", "c5"])
def test_imagine_block(tag):
    g = load(f"imagine_{tag}.npz")
    d, salt = DIMS[tag], int(g["salt"])
    wm, actor, critic = build_models(d, salt)
    h0, z0, _ = start_state(d, salt)
    ez = expo("imagine_z", (d["H"], d["B"], d["L"], d["K"]), salt)
    with torch.no_grad():
        out = O.imagine_block(weights_of(wm), weights_of(actor), weights_of(critic), h0, z0,
                              torch.from_numpy(g["noise_a"]), ez, float(g["gamma"]), float(g["lam"]))
    assert np.array_equal(out["actions"].numpy(), g["actions"])
    assert np.array_equal(out["z"].argmax(-1).numpy().astype(np.uint8), g["zidx"])
    for key, ref in (("h", "h"), ("rewards", "rewards"), ("conts", "conts"), ("values", "values"),
                     ("v_last", "v_last"), ("lam_returns", "lam_returns"), ("actor_logp", "actor_logp")):
        assert rel_err(out[key], g[ref]) < TOL, key
    if tag == "small":
        assert rel_err(out["prior_logits"].reshape(g["prior_logits"].shape), g["prior_logits"]) < TOL
    # actor / critic losses on the imagined batch
    Hh, Hz = torch.from_numpy(g["h"]), out["z"]
    A, ret, V = (torch.from_numpy(g[k]) for k in ("actions", "lam_returns", "values"))
    wc, wa = weights_of(critic), weights_of(actor)
    with torch.no_grad():
        cl = O.critic_loss(wc, Hh, Hz, ret)
        S, p05, p95 = O.percentile_scale(ret, torch.tensor(0.0), torch.tensor(0.0), 0.99, 1.0)
        al = O.actor_loss(wa, Hh, Hz, A, ret, V, S, 3e-4)
        O.update_slow(wc, 0.98)
    assert rel_err(cl, g["critic_loss"]) < TOL and rel_err(al, g["actor_loss"]) < TOL
    assert rel_err(p05, g["p05"]) < TOL and rel_err(p95, g["p95"]) < TOL
    assert rel_err(wc["slow.head.bias"], g["slow_head_bias"]) < 1e-6
    assert rel_err(wc["slow.mlp.0.weight"][0], g["slow_mlp0_row0"]) < 1e-6


@pytest.mark.parametrize("tag", ["small", "default"])
def test_observe_loss_and_grads(tag):
    g = load(f"observe_{tag}.npz")
    d, salt, B = DIMS[tag], int(g["salt"]), int(g["B"])
    wm, _, _ = build_models(d, salt)
    w = {k: v.requires_grad_(v.is_floating_point()) for k, v in weights_of(wm).items()}
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
    total, info = O.wm_loss(w, obs.float() / 255.0, actions, rewards, continues, noise["init"], noise["reset"], noise["z"])
    assert np.array_equal(info["idx"].numpy().astype(np.uint8), g["zidx"])
    for key, ref in (("h", "h"), ("post_logits", "post_logits"), ("prior_logits", "prior_logits"), ("h_last", "h_last")):
        assert rel_err(info[key].detach(), g[ref]) < TOL, key
    assert rel_err(total.detach(), g["total"]) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(info[key + "_loss"].detach(), g[key]) < TOL, key
    total.backward()
    names, norms = list(g["grad_names"]), g["grad_norms"]
    for name, ref in zip(names, norms):
        grad = w[str(name)].grad
        if ref < 0:
            assert grad is None or float(grad.abs().max()) == 0.0, name  # seq.norm.* receive no gradient
        else:
            assert abs(float(grad.double().norm()) - ref) <= 5e-5 * max(ref, 1e-3), name
    for key in g.files:
        if key.startswith("grad:"):
            got = w[key[5:]].grad.reshape(-1)[:256]
            assert rel_err(got, g[key]) < 1e-4, key


@pytest.mark.parametrize("name", ["observe_full", "observe_t50"])
def test_observe_full_size_batches(name):
    """The per-GPU training batch at default sizes (16 sequences x 64 steps) and the 50-step variant: losses, every
    sampled index, recurrent states and gradient norms of the reference."""
    g = load(f"{name}.npz")
    d, salt, B, T = DIMS["default"], int(g["salt"]), int(g["B"]), int(g["T"])
    wm, _, _ = build_models(d, salt)
    w = {k: v.requires_grad_(v.is_floating_point()) for k, v in weights_of(wm).items()}
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B, T)
    total, info = O.wm_loss(w, obs.float() / 255.0, actions, rewards, continues, noise["init"], noise["reset"], noise["z"])
    assert np.array_equal(info["idx"].numpy().astype(np.uint8), g["zidx"])
    assert rel_err(info["h"].detach()[:, 7::8], g["h_every8"]) < TOL
    assert rel_err(info["h_last"].detach(), g["h_last"]) < TOL
    assert rel_err(info["post_logits"].detach()[:, -1], g["post_last"]) < TOL
    assert rel_err(info["prior_logits"].detach().reshape(B, T, d["L"], d["K"])[:, -1], g["prior_last"]) < TOL
    assert rel_err(total.detach(), g["total"]) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(info[key + "_loss"].detach(), g[key]) < TOL, key
    total.backward()
    for name_, ref in zip(g["grad_names"], g["grad_norms"]):
        grad = w[str(name_)].grad
        if ref < 0:
            assert grad is None or float(grad.abs().max()) == 0.0, name_
        else:
            assert abs(float(grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-3), (name_, float(grad.double().norm()), ref)


def test_observe_scaled_model():
    """The scaled model (recurrent width 4096, 32 x 32 latent) over 16 x 3 frames: the oracle's loss terms, draws, states
    and gradient norms equal the reference's."""
    g = load("observe_c5.npz")
    d, salt, B, T = DIMS["c5"], int(g["salt"]), int(g["B"]), int(g["T"])
    wm, _, _ = build_models(d, salt)
    w = {k: v.requires_grad_(v.is_floating_point()) for k, v in weights_of(wm).items()}
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B, T)
    total, info = O.wm_loss(w, obs.float() / 255.0, actions, rewards, continues, noise["init"], noise["reset"], noise["z"])
    assert np.array_equal(info["idx"].numpy().astype(np.uint8), g["zidx"])
    assert rel_err(info["h"].detach(), g["h"]) < TOL
    assert rel_err(info["post_logits"].detach()[:, -1], g["post_last"]) < TOL
    assert rel_err(total.detach(), g["total"]) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(info[key + "_loss"].detach(), g[key]) < TOL, key
    total.backward()
    for name_, ref in zip(g["grad_names"], g["grad_norms"]):
        grad = w[str(name_)].grad
        if ref < 0:
            assert grad is None or float(grad.abs().max()) == 0.0, name_
        else:
            assert abs(float(grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-3), (name_, float(grad.double().norm()), ref)


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="the reference tree exists only in the build container")
def test_recipe_imports_the_reference_and_reproduces_a_fixture(tmp_path):
    """tests/golden/make_golden.py, run as committed, must bind ``lib`` to the reference's package (not to this
    repository's alias package of the same name) and regenerate a committed fixture bit for bit."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.argv=['make_golden']; sys.path.insert(0, %r); import make_golden as mg; "
            "assert sys.modules['lib.world_model'].__file__.startswith('/root/reference/'); "
            "mg.golden_observe('small', %r, salt=11); mg.golden_imagine('small', %r, salt=11)"
            % (os.path.join(here, "golden"), str(tmp_path), str(tmp_path)))
    subprocess.run([sys.executable, "-c", code], check=True, cwd=os.path.dirname(here), timeout=600)
    for f in ("observe_small.npz", "imagine_small.npz"):
        new, old = np.load(tmp_path / f), load(f)
        assert sorted(new.files) == sorted(old.files)
        for k in old.files:
            assert np.array_equal(new[k], old[k]), (f, k)
