"""Fused LayerNorm[C,H,W] + SiLU (forward and backward) against torch's own ops (GPU)."""
import pytest
import torch
import torch.nn.functional as F

from dreamer_amd.world_model import _LayerNormSiLUFn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N,shape", [(1024, (32, 32, 32)), (96, (64, 16, 16)), (37, (128, 8, 8)), (5, (256, 4, 4)), (3, (4, 8, 8)),
                                     (9, (2, 2, 1))])
def test_matches_torch_layer_norm_silu(N, shape):
    g = torch.Generator().manual_seed(N)
    x = (torch.randn(N, *shape, generator=g) * 1.7 + 0.3).to(DEV).requires_grad_(True)
    w = (1.0 + 0.2 * torch.randn(*shape, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(*shape, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(N, *shape, generator=g).to(DEV)
    ref = F.silu(F.layer_norm(x, shape, w, b, 1e-5))
    gx, gw, gb = torch.autograd.grad(ref, (x, w, b), dy)
    out = _LayerNormSiLUFn.apply(x, w, b, 1e-5)
    hx, hw, hb = torch.autograd.grad(out, (x, w, b), dy)
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert rel(out, ref) < 2e-6
    assert rel(hx, gx) < 1e-5
    assert rel(hw, gw) < 1e-5 and rel(hb, gb) < 1e-5
