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


@pytest.mark.parametrize("D,B,T", [(1024, 5, 4), (640, 70, 3), (4096, 2, 3)])
def test_wide_recurrent_state_matches_oracle_and_host_path(D, B, T):
    """Recurrent widths beyond 512 (up to the scaled model's 4096) run the scan with every product on the large-row
    kernel: forward values against the oracle's step-by-step loss, every gradient against the host eager path."""
    import oracle.dreamer_oracle as O
    from helpers import weights_of
    d = dict(DIMS["small"], D=D, M=64, L=8, A=5, T=T)
    salt = 900 + B
    wm_gpu, _, _ = build_models(d, salt)
    wm_cpu, _, _ = build_models(d, salt)
    ww = weights_of(wm_cpu)
    wm_gpu = wm_gpu.to(DEV)
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
    continues[:, 1] = False   # a reset in every sequence
    with torch.no_grad():
        ref_total, ref = O.wm_loss(ww, obs.float() / 255.0, actions, rewards, continues, noise["init"], noise["reset"], noise["z"])
    lg, ig = wm_gpu.loss(obs.to(DEV).float() / 255.0, actions.to(DEV), rewards.to(DEV), continues.to(DEV),
                         noise={k: v.to(DEV) for k, v in noise.items()})
    assert torch.equal(ig["z"].detach().argmax(-1).cpu(), ref["idx"])
    assert rel_err(ig["h"].detach().cpu(), ref["h"]) < TOL
    assert rel_err(ig["post_logits"].detach().cpu(), ref["post_logits"]) < TOL
    assert rel_err(ig["prior_logits"].detach().cpu(), ref["prior_logits"]) < TOL
    assert rel_err(lg.detach().cpu(), ref_total) < TOL
    for key in ("pred", "dyn", "rep"):
        assert rel_err(ig[key + "_loss"].detach().cpu(), ref[key + "_loss"]) < TOL, key
    lc, _ = wm_cpu.loss(obs.float() / 255.0, actions, rewards, continues, noise=noise)
    lc.backward(); lg.backward()
    pc, pg = dict(wm_cpu.named_parameters()), dict(wm_gpu.named_parameters())
    for name, p in pc.items():
        if p.grad is None:
            assert pg[name].grad is None, name
        else:
            assert rel_err(pg[name].grad.cpu(), p.grad) < 2e-4, name


def test_wide_recurrent_state_with_bf16_matrix_inputs():
    """The scaled model's width (4096) with the operands of the scan's products rounded to bf16 (fp32 accumulation):
    same draws, states, logits, loss and gradients within the bf16 tolerance of 1e-2 of the fp32 path."""
    d = dict(DIMS["small"], D=4096, M=64, L=8, A=5, T=3)
    salt, B = 941, 3
    wm, _, _ = build_models(d, salt)
    wm = wm.to(DEV)
    obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
    args = (obs.to(DEV).float() / 255.0, actions.to(DEV), rewards.to(DEV), continues.to(DEV))
    noise = {k: v.to(DEV) for k, v in noise.items()}
    l32, i32 = wm.loss(*args, noise=noise)
    l32.backward()
    g32 = {n: p.grad.clone() for n, p in wm.named_parameters() if p.grad is not None}
    wm.zero_grad(set_to_none=True)
    wm.observe_precision = "bf16"
    l16, i16 = wm.loss(*args, noise=noise)
    l16.backward()
    assert torch.equal(i16["z"].argmax(-1), i32["z"].argmax(-1))
    assert not torch.equal(i16["h"], i32["h"])      # the other kernels ran
    for key in ("h", "post_logits", "prior_logits"):
        assert rel_err(i16[key].detach().cpu(), i32[key].detach().cpu()) < 1e-2, key
    assert rel_err(l16.detach().cpu(), l32.detach().cpu()) < 1e-2
    for n, p in wm.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad.cpu(), g32[n].cpu()) < 3e-2, n
