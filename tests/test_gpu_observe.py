"""The native observe recurrence and the world-model loss built on it, against the reference's
golden vectors and against the host eager path (GPU)."""
import numpy as np
import pytest
import torch

from helpers import DIMS, build_models, load, observe_inputs, rel_err

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
