"""The native observe recurrence and the world-model loss built on it, against the reference's
golden vectors and against the host eager path (GPU)."""
import numpy as np
import pytest
import torch

from helpers import DIMS, assert_close_elementwise, build_models, load, observe_inputs, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


@pytest.mark.parametrize("tag", ["small", "default"])
def test_world_model_loss_and_grads_match_reference(tag):
    g = load(f"observe_{tag}.npz")
    d, salt, B = DIMS[tag], int(g["salt"]), int(g["B"])
    wm, _, _ = build_models(d, salt)
    wm = wm.to(DEV)
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
    noise = {k: v.to(DEV) for k, v in noise.items()}
    total, info = wm.loss(obs.to(DEV).float() / 255.0, actions.to(DEV), rewards.to(DEV), continues.to(DEV), noise=noise)
    assert np.array_equal(info["state"].z.detach().argmax(-1).cpu().numpy().astype(np.uint8), g["zidx"][:, -1])
    for key, ref in (("h", "h"), ("post_logits", "post_logits"), ("prior_logits", "prior_logits")):
        assert rel_err(info[key].detach().cpu(), g[ref]) < TOL, key
    assert rel_err(total.detach().cpu(), g["total"]) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(info[key + "_loss"].detach().cpu(), g[key]) < TOL, key
    total.backward()
    params = dict(wm.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        grad = params[str(name)].grad
        if ref < 0:
            assert grad is None, name
        else:
            assert abs(float(grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-3), (name, float(grad.double().norm()), ref)
    for key in g.files:
        if key.startswith("grad:"):
            assert rel_err(params[key[5:]].grad.reshape(-1)[:256].cpu(), g[key]) < 2e-4, key


@pytest.mark.parametrize("name", ["observe_full", "observe_t50"])
def test_full_size_batches_match_reference(name):
    """The per-GPU training batch at default sizes (16 sequences x 64 steps, lib/config.py:18-19) and the 50-step
    variant through ``WorldModel.loss`` + backward: every sampled index, the recurrent states, the three loss terms
    and every parameter's gradient norm against the reference's."""
    g = load(f"{name}.npz")
    d, salt, B, T = DIMS["default"], int(g["salt"]), int(g["B"]), int(g["T"])
    wm, _, _ = build_models(d, salt)
    wm = wm.to(DEV)
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B, T)
    noise = {k: v.to(DEV) for k, v in noise.items()}
    total, info = wm.loss(obs.to(DEV).float() / 255.0, actions.to(DEV), rewards.to(DEV), continues.to(DEV), noise=noise)
    assert np.array_equal(info["z"].detach().argmax(-1).cpu().numpy().astype(np.uint8), g["zidx"])
    assert np.array_equal(info["state"].z.detach().argmax(-1).cpu().numpy().astype(np.uint8), g["zidx"][:, -1])
    h = info["h"].detach().cpu()
    assert rel_err(h[:, 7::8], g["h_every8"]) < TOL
    assert_close_elementwise(h[:, 7::8], g["h_every8"], rtol=1e-5, atol_scale=2e-6, what="h")
    assert rel_err(info["state"].h.detach().cpu(), g["h_last"]) < TOL
    assert rel_err(info["post_logits"].detach()[:, -1].cpu(), g["post_last"]) < TOL
    assert rel_err(info["prior_logits"].detach()[:, -1].cpu(), g["prior_last"]) < TOL
    assert rel_err(total.detach().cpu(), g["total"]) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(info[key + "_loss"].detach().cpu(), g[key]) < TOL, key
    total.backward()
    params = dict(wm.named_parameters())
    for pname, ref in zip(g["grad_names"], g["grad_norms"]):
        grad = params[str(pname)].grad
        if ref < 0:
            assert grad is None, pname
        else:
            assert abs(float(grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-3), (pname, float(grad.double().norm()), ref)
    for key in g.files:
        if key.startswith("grad:"):
            assert rel_err(params[key[5:]].grad.reshape(-1)[:256].cpu(), g[key]) < 2e-4, key


def test_scan_matches_host_eager_path_with_episode_boundaries():
    d = dict(DIMS["small"], D=128, L=16, A=5, T=12)
    salt, B = 31, 7
    wm_gpu, _, _ = build_models(d, salt)
    wm_cpu, _, _ = build_models(d, salt)
    wm_gpu = wm_gpu.to(DEV)
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
    continues[:, 3] = False   # force resets in every sequence
    continues[2, 7] = False
    lc, ic = wm_cpu.loss(obs.float() / 255.0, actions, rewards, continues, noise=noise)
    lg, ig = wm_gpu.loss(obs.to(DEV).float() / 255.0, actions.to(DEV), rewards.to(DEV), continues.to(DEV),
                         noise={k: v.to(DEV) for k, v in noise.items()})
    assert rel_err(ig["h"].detach().cpu(), ic["h"].detach()) < TOL
    assert rel_err(ig["post_logits"].detach().cpu(), ic["post_logits"].detach()) < TOL
    assert torch.equal(ig["state"].z.detach().cpu().argmax(-1), ic["state"].z.detach().argmax(-1))
    assert rel_err(lg.detach().cpu(), lc.detach()) < TOL
    lc.backward(); lg.backward()
    pc, pg = dict(wm_cpu.named_parameters()), dict(wm_gpu.named_parameters())
    for name, p in pc.items():
        if p.grad is None:
            assert pg[name].grad is None, name
        else:
            assert rel_err(pg[name].grad.cpu(), p.grad) < 2e-4, name


@pytest.mark.parametrize("B", [1, 3])
def test_act_step_runs_the_scan_kernels_and_matches_step(B):
    """One acting step (B = 1 in the trainer) through the fused scan with T = 1 against the eager
    ``step`` (lib/world_model.py:415-457) on the same generator state."""
    wm, _, _ = build_models(DIMS["default"], 5)
    wm = wm.to(DEV)
    g = torch.Generator().manual_seed(B)
    state = wm.init_state(B, device=DEV)
    state.h.copy_(torch.randn(B, wm.hidden_size, generator=g).tanh())
    x = torch.rand(B, 3, 64, 64, generator=g).to(DEV)
    a = torch.randint(0, wm.action_size, (B,), generator=g).to(DEV)
    c = torch.ones(B, 1, device=DEV)
    c[B - 1] = 0.0
    for _ in range(2):   # two consecutive steps: the second starts from a sampled state
        with torch.no_grad():
            torch.manual_seed(100 + B)
            ref, _ = wm.step(state, a, x, c)
            after_ref = torch.rand(4, device=DEV)
            torch.manual_seed(100 + B)
            got = wm.act_step(state, a, x, c)
            after_got = torch.rand(4, device=DEV)
        assert rel_err(got.h.cpu(), ref.h.cpu()) < TOL
        assert torch.equal(got.z.argmax(-1), ref.z.argmax(-1))
        assert torch.equal(got.z, (ref.z > 0.5).float())
        assert torch.equal(after_ref, after_got)
        state, c = got, torch.ones(B, 1, device=DEV)
