"""Flat clip + Adam step of the native library against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(fused)
(the optimiser calls of lib/trainer.py:95-103, :170-173) on the same gradients."""
import pytest
import torch
import torch.nn as nn

from dreamer_amd.dist import FlatGrads
from dreamer_amd.optim import FlatAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(37, 129), nn.LayerNorm(129), nn.SiLU(), nn.Linear(129, 5), nn.Embedding(7, 3)).to(DEV)


@pytest.mark.parametrize("clip,scale", [(1000.0, 1.0), (0.5, 1.0), (100.0, 50.0)])
def test_flat_adam_matches_torch_fused_adam(clip, scale):
    ref, mine = _model(1), _model(1)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=3e-4, eps=1e-8, fused=True)
    grads = FlatGrads(mine, static=True)
    opt = FlatAdam(grads, lr=3e-4, eps=1e-8, clip=clip)
    assert all(p.data_ptr() % 256 == 0 for p in mine.parameters())
    g = torch.Generator(device=DEV).manual_seed(7)
    for step in range(6):
        opt_ref.zero_grad(set_to_none=False)
        grads.zero()
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            gr = torch.randn(pr.shape, device=DEV, generator=g) * scale * (0.0 if step == 3 and pr.dim() == 1 else 1.0)
            pr.grad = gr.clone()
            if not (step == 4 and pm.dim() == 1):   # a parameter without a gradient counts as a zero gradient
                pm.grad = gr.clone()
            else:
                pr.grad.zero_()
        norm_ref = nn.utils.clip_grad_norm_(ref.parameters(), clip)
        opt_ref.step()
        grads.sync()
        opt.step()
        assert abs(float(opt.grad_norm) - float(norm_ref)) <= 1e-5 * float(norm_ref)
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            err = float((pr - pm).abs().max())
            assert err <= 2e-6 * max(float(pr.abs().max()), 1e-3) + 1e-9, (step, err)
    sd_ref, sd = opt_ref.state_dict(), opt.state_dict()
    assert sd["param_groups"][0]["params"] == sd_ref["param_groups"][0]["params"]
    for i in sd_ref["state"]:
        assert float(sd["state"][i]["step"]) == float(sd_ref["state"][i]["step"]) == 6.0
        for key in ("exp_avg", "exp_avg_sq"):
            a, b = sd["state"][i][key], sd_ref["state"][i][key]
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12


def test_flat_adam_in_a_recorded_graph():
    m = _model(2)
    grads = FlatGrads(m, static=True)
    opt = FlatAdam(grads, lr=1e-3, eps=1e-8, clip=10.0)
    ref = _model(2)
    opt_ref = FlatAdam(FlatGrads(ref, static=True), lr=1e-3, eps=1e-8, clip=10.0)
    graph = torch.cuda.CUDAGraph()
    grads.flat.normal_()
    torch.cuda.synchronize()
    from dreamer_amd.trainer import _recording
    with _recording(graph):
        opt.step()
    for i in range(4):
        gr = torch.randn_like(grads.flat)
        grads.flat.copy_(gr)
        opt_ref.grads.flat.copy_(gr)
        graph.replay()
        opt_ref.step()
    assert float(opt.step_count) == float(opt_ref.step_count) == 4.0
    assert torch.equal(grads.flat_params, opt_ref.grads.flat_params)


def test_state_dict_round_trips_through_torch_adam():
    """A checkpoint written from the flat optimiser loads into torch.optim.Adam (the reference's optimiser,
    lib/trainer.py:95-103, saved at :326-335) and into a fresh flat optimiser; both continue identically."""
    a, b, c = _model(3), _model(3), _model(3)
    ga = FlatGrads(a, static=True)
    oa = FlatAdam(ga, lr=1e-3, eps=1e-8, clip=0.0)
    g = torch.Generator(device=DEV).manual_seed(1)
    for _ in range(3):
        ga.flat.copy_(torch.randn(ga.flat.numel(), device=DEV, generator=g))
        oa.step()
    sd = oa.state_dict()
    with torch.no_grad():
        for pa, pb, pc in zip(a.parameters(), b.parameters(), c.parameters()):
            pb.copy_(pa); pc.copy_(pa)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, eps=1e-8, fused=True)
    ob.load_state_dict(sd)
    gc_ = FlatGrads(c, static=True)
    oc = FlatAdam(gc_, lr=1e-3, eps=1e-8, clip=0.0)
    oc.load_state_dict(sd)
    grad = torch.randn(ga.flat.numel(), device=DEV, generator=g)
    ga.flat.copy_(grad); gc_.flat.copy_(grad)
    for pb, view in zip(b.parameters(), ga.views):
        pb.grad = view.clone()
    oa.step(); ob.step(); oc.step()
    for pa, pb, pc in zip(a.parameters(), b.parameters(), c.parameters()):
        assert torch.equal(pa, pc)
        assert float((pa - pb).abs().max()) <= 2e-6 * max(float(pa.abs().max()), 1e-3)


def test_critic_checkpoint_loads_into_the_reference_optimiser():
    """The reference builds Adam(critic.parameters()) (lib/trainer.py:101), whose single group also lists the frozen
    ``slow.*`` copy (lib/critic.py:51-53).  The flat optimiser's state dict must list as many parameter ids, carry
    state only for parameters that received gradients, load into that torch optimiser, and accept its state dict."""
    from dreamer_amd.critic import Critic
    from dreamer_amd.world_model import WorldModelState
    torch.manual_seed(0)
    mk = lambda: Critic(64 + 4 * 32, 32, 2, 255, -20.0, 20.0, 0.98, 1.0).to(DEV)
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    ga = FlatGrads(a, static=True)
    oa = FlatAdam(ga, lr=3e-5, eps=1e-5, clip=100.0)
    ob = torch.optim.Adam(b.parameters(), lr=3e-5, eps=1e-5, fused=True)
    h = torch.randn(6, 5, 64, device=DEV)
    z = torch.nn.functional.one_hot(torch.randint(0, 32, (6, 5, 4), device=DEV), 32).float()
    ret = torch.randn(6, 5, device=DEV) * 3
    for _ in range(2):
        ga.zero(); ob.zero_grad()
        a.loss(WorldModelState(h=h, z=z), ret).backward()
        b.loss(WorldModelState(h=h, z=z), ret).backward()
        ga.sync(); oa.step()
        nn.utils.clip_grad_norm_(b.parameters(), 100.0); ob.step()
    sd, sd_ref = oa.state_dict(), ob.state_dict()
    n_all = len(list(a.parameters()))
    assert sd["param_groups"][0]["params"] == sd_ref["param_groups"][0]["params"] == list(range(n_all))
    assert sorted(sd["state"]) == sorted(sd_ref["state"]) and len(sd["state"]) == n_all // 2
    fresh = torch.optim.Adam(mk().parameters(), lr=3e-5, eps=1e-5, fused=True)
    fresh.load_state_dict(sd)                      # what the reference-side loader would do with the checkpoint
    for i, st in sd_ref["state"].items():
        for key in ("exp_avg", "exp_avg_sq"):
            assert float((sd["state"][i][key] - st[key]).abs().max()) <= 1e-5 * float(st[key].abs().max()) + 1e-12
    c = mk()
    c.load_state_dict(a.state_dict())
    oc = FlatAdam(FlatGrads(c, static=True), lr=3e-5, eps=1e-5, clip=100.0)
    oc.load_state_dict(sd_ref)                     # and the other way round
    assert float(oc.step_count) == 2.0
    assert float((oc.exp_avg - oa.exp_avg).abs().max()) <= 1e-5 * float(oa.exp_avg.abs().max())
