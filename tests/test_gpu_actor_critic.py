"""Row kernels of the critic / actor updates against the eager formulas of lib/critic.py:103-161 and
lib/actor.py:16-33 (GPU)."""
import copy

import numpy as np
import pytest
import torch

from dreamer_amd.actor import EMAPercentileScale
from dreamer_amd.critic import Critic
from dreamer_amd.world_model import WorldModelState
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _critic(seed, reg=1.0, nb=255):
    torch.manual_seed(seed)
    c = Critic(64 + 4 * 32, 32, 2, nb, -20.0, 20.0, 0.98, reg)
    with torch.no_grad():
        torch.nn.init.normal_(c.head.weight, std=0.3)
        torch.nn.init.normal_(c.slow.head.weight, std=0.3)
        torch.nn.init.normal_(c.slow.head.bias, std=0.3)
    return c


@pytest.mark.parametrize("reg,nb", [(1.0, 255), (0.0, 255), (0.5, 41)])
def test_critic_loss_rows_match_the_eager_loss(reg, nb):
    c_gpu, c_cpu = _critic(3, reg, nb).to(DEV), _critic(3, reg, nb)
    g = torch.Generator().manual_seed(1)
    h = torch.randn(7, 5, 64, generator=g)
    z = torch.nn.functional.one_hot(torch.randint(0, 32, (7, 5, 4), generator=g), 32).float()
    # returns spread over the whole bucket range, bucket edges and the clamped ends included
    ret = torch.cat([torch.randn(30, generator=g) * 50, torch.tensor([0.0, -1e9, 1e9, 4.85e8, -4.85e8])]).reshape(7, 5)
    ref = c_cpu._loss_eager(WorldModelState(h=h, z=z), ret)
    got = c_gpu.loss(WorldModelState(h=h.to(DEV), z=z.to(DEV)), ret.to(DEV))
    assert rel_err(got.detach().cpu(), ref.detach()) < 1e-5
    ref.backward(); got.backward()
    for (k, p), (_, q) in zip(c_gpu.named_parameters(), c_cpu.named_parameters()):
        if q.grad is None:
            assert p.grad is None, k
        else:
            assert rel_err(p.grad.cpu(), q.grad) < 2e-5, k


def test_ema_multi_equals_the_tensor_loop_bit_for_bit():
    c = _critic(5).to(DEV)
    with torch.no_grad():
        for p in c.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    ref = copy.deepcopy(c)
    c.update_slow()
    with torch.no_grad():
        for p, ps in zip(ref.parameters(), ref.slow.parameters()):
            ps.data.mul_(ref.ema_decay)
            ps.data.add_(p.data * (1.0 - ref.ema_decay))
    for (k, a), (_, b) in zip(c.slow.named_parameters(), ref.slow.named_parameters()):
        assert torch.equal(a, b), k
    for (k, a), (_, b) in zip(c.named_parameters(), ref.named_parameters()):
        if not k.startswith("slow."):
            assert torch.equal(a, b), k


@pytest.mark.parametrize("n", [1, 2, 3, 240, 241, 1920, 5000, 16384])
def test_percentile_scale_matches_torch_quantile(n):
    g = torch.Generator().manual_seed(n)
    cpu, gpu = EMAPercentileScale(0.99, 1.0).train(), EMAPercentileScale(0.99, 1.0).train().to(DEV)
    for step in range(3):
        x = torch.randn(n, generator=g) * (40.0 if step == 1 else 1.0) + step
        s_ref = cpu.update_get_S(x)
        s_got = gpu.update_get_S(x.to(DEV))
        assert rel_err(s_got.cpu(), s_ref) < 1e-6
        assert rel_err(gpu.p05.cpu(), cpu.p05) < 1e-5 and rel_err(gpu.p95.cpu(), cpu.p95) < 1e-5
    gpu.eval(); cpu.eval()
    before = gpu.p05.clone()
    assert rel_err(gpu.update_get_S(x.to(DEV)).cpu(), cpu.update_get_S(x)) < 1e-6
    assert torch.equal(gpu.p05, before)      # no EMA update outside training
