"""Fused KL-pair and per-frame squared-error kernels against the eager formulas of
lib/world_model.py:463-528 (values and gradients, fp32 tolerance 1e-5 relative)."""
import pytest
import torch

from dreamer_amd import losses as FL
from dreamer_amd.world_model import WorldModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("rows,L,scale", [(1024, 32, 1.0), (7, 6, 4.0), (33, 8, 0.05), (1, 1, 2.0)])
def test_kl_pair_matches_eager_losses(rows, L, scale):
    g = torch.Generator().manual_seed(rows + L)
    wm = WorldModel((3, 16, 16), 4, num_latents=L, classes_per_latent=32, hidden_size=32, base_cnn_channels=4,
                    mlp_hidden_units=32)
    post = (torch.randn(rows, L, 32, generator=g) * scale).to(DEV).requires_grad_(True)
    prior = (torch.randn(rows, L, 32, generator=g) * scale).to(DEV).requires_grad_(True)
    if rows > 4:   # some categoricals below the free-bits floor: identical logits give KL = 0
        with torch.no_grad():
            prior[::3] = post[::3]
    w_d = torch.rand(rows, generator=g).to(DEV)
    w_r = torch.rand(rows, generator=g).to(DEV)
    dyn_ref, rep_ref = wm._dynamics_loss(post, prior), wm._representation_loss(post, prior)
    ((dyn_ref * w_d).sum() + (rep_ref * w_r).sum()).backward()
    gp_ref, gq_ref = prior.grad.clone(), post.grad.clone()
    prior.grad = post.grad = None
    dyn, rep = FL.kl_pair(post, prior, wm.unimix_eps, wm.free_bits / L)
    ((dyn * w_d).sum() + (rep * w_r).sum()).backward()
    assert _rel(dyn, dyn_ref) < 1e-5 and _rel(rep, rep_ref) < 1e-5
    assert _rel(prior.grad, gp_ref) < 2e-5 and _rel(post.grad, gq_ref) < 2e-5
    # only one of the two uses
    prior.grad = post.grad = None
    FL.kl_pair(post, prior, wm.unimix_eps, wm.free_bits / L)[0].sum().backward()
    assert float(post.grad.abs().max()) == 0.0
    if scale >= 1.0:   # at scale 0.05 every KL is under the free-bits floor: no gradient at all
        assert float(prior.grad.abs().max()) > 0.0


@pytest.mark.parametrize("rows,shape", [(1024, (3, 64, 64)), (5, (3, 16, 16)), (1, (1, 2, 2))])
def test_mse_rows_matches_eager(rows, shape):
    g = torch.Generator().manual_seed(rows)
    x_hat = torch.randn(rows, *shape, generator=g).to(DEV).requires_grad_(True)
    x = torch.rand(rows, *shape, generator=g).to(DEV)
    w = torch.rand(rows, generator=g).to(DEV)
    ref = 0.5 * (x_hat - x).pow(2).mean(dim=(-3, -2, -1))
    (ref * w).sum().backward()
    g_ref = x_hat.grad.clone()
    x_hat.grad = None
    out = FL.mse_rows(x_hat, x)
    (out * w).sum().backward()
    assert _rel(out, ref) < 1e-5 and _rel(x_hat.grad, g_ref) < 1e-5
