"""Noise drawn inside the imagination kernels (counter-based generator): distribution of the draws it drives,
reproducibility from torch's device generator state, fresh numbers on every replay of the recorded block."""
import numpy as np
import pytest
import torch

from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.world_model import WorldModelState
from helpers import DIMS, build_models, start_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _models(d, salt, uniform=False, head_bias=None):
    wm, actor, critic = build_models(d, salt)
    with torch.no_grad():
        if uniform or head_bias is not None:
            wm.dyn.head.weight.zero_(); wm.dyn.head.bias.zero_()
            actor.head.weight.zero_(); actor.head.bias.zero_()
        if head_bias is not None:
            wm.dyn.head.bias.copy_(head_bias.repeat(d["L"]))
    return wm, actor, critic


def _chi2(counts, probs):
    n = counts.sum()
    return float(((counts - n * probs) ** 2 / (n * probs)).sum())


@pytest.mark.parametrize("B,precision", [(64, "f32"), (7, "f32"), (512, "f32_mfma"), (2048, "f32")])
def test_draws_follow_their_distributions(B, precision):
    """Constant logits make every latent draw a draw from one known distribution (the unimix of softmax(bias)) and
    every action draw uniform: chi-square of the observed class counts over all (trajectory, step, latent) draws, in
    the few-row kernels, the many-row kernels and the two side-by-side chains of a 2048-trajectory call."""
    d, salt, H = dict(DIMS["small"], D=64, M=64, L=8, A=6), 3, 6
    bias = torch.linspace(-2.0, 2.0, 32)
    wm, actor, critic = _models(d, salt, head_bias=bias)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    torch.manual_seed(11)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, precision=precision, want_dense_z=False,
                  device_noise=True)
    torch.cuda.synchronize()
    z = out["zidx"].cpu().numpy().reshape(-1)
    p = (0.99 * torch.softmax(bias, 0) + 0.01 / 32).double().numpy()
    cz = np.bincount(z, minlength=32).astype(np.float64)
    assert z.size == B * H * d["L"]
    assert _chi2(cz, p) < 31 + 6 * np.sqrt(2 * 31), (_chi2(cz, p), cz)
    a = out["actions"].cpu().numpy().reshape(-1)
    ca = np.bincount(a, minlength=d["A"]).astype(np.float64)
    assert _chi2(ca, np.full(d["A"], 1.0 / d["A"])) < 5 + 6 * np.sqrt(2 * 5), ca
    # distinct addresses for every draw: neighbouring latents, steps and trajectories are not copies of one another
    zz = out["zidx"].cpu().numpy()
    if B >= 64:
        assert (zz[:, :, :-1] == zz[:, :, 1:]).mean() < 0.2 and (zz[:, :-1] == zz[:, 1:]).mean() < 0.2
        assert (zz[:-1] == zz[1:]).mean() < 0.2 and (zz[: B // 2] == zz[B // 2:]).mean() < 0.2


def test_reproducible_from_the_generator_state_and_fresh_on_every_call():
    d, salt, B, H = DIMS["default"], 9, 16, 15
    wm, actor, critic = _models(d, salt)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    start = WorldModelState(h=h0.to(DEV), z=z0.to(DEV))
    run = lambda **kw: {k: v.clone() for k, v in eng.run(start, H, 0.997, 0.95, **dict(dict(device_noise=True), **kw)).items()}
    torch.manual_seed(5)
    a1, a2 = run(), run()
    torch.manual_seed(5)
    b1 = run()
    state = torch.cuda.get_rng_state()
    b2 = run()
    assert torch.equal(a1["zidx"], b1["zidx"]) and torch.equal(a1["actions"], b1["actions"]) and torch.equal(a1["h"], b1["h"])
    assert torch.equal(a2["zidx"], b2["zidx"]) and torch.equal(a2["lam_returns"], b2["lam_returns"])
    assert not torch.equal(a1["zidx"], a2["zidx"])                    # a new block of the generator's stream per call
    torch.cuda.set_rng_state(state)
    assert torch.equal(run()["zidx"], b2["zidx"])                     # a restored generator state replays the call
    # the recorded block (fixed buffers, one graph) draws the same numbers as direct launches, fresh on every replay
    torch.manual_seed(5)
    p1, p2 = run(persistent=True), run(persistent=True)
    assert torch.equal(p1["zidx"], a1["zidx"]) and torch.equal(p1["h"], a1["h"])
    assert torch.equal(p2["zidx"], a2["zidx"]) and torch.equal(p2["actions"], a2["actions"])
    # the host-noise path is still there and is a different stream
    torch.manual_seed(5)
    h1 = run(device_noise=False)
    assert not torch.equal(h1["zidx"], a1["zidx"]) and torch.isfinite(h1["lam_returns"]).all()


def test_shapes_without_the_in_kernel_generator_fall_back_to_noise_tensors():
    d, salt, B, H = dict(DIMS["small"], D=64, M=64, L=8, A=6), 4, 100, 2      # 65..255 trajectories: not the step-wise sequence
    wm, actor, critic = _models(d, salt)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    assert not eng._device_noise_ok(B, 0) and eng._device_noise_ok(16, 0) and eng._device_noise_ok(300, 2)
    h0, z0, _ = start_state(d, salt, B)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95)
    assert torch.isfinite(out["lam_returns"]).all()
