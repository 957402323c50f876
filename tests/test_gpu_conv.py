"""The towers' stride-2 4x4 convolutions on the implicit-product kernels against torch's convolutions in fp64 (GPU)."""
import pytest
import torch
import torch.nn.functional as F

from dreamer_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, tol=3e-6):
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= tol * scale, (float((got.double() - ref).abs().max()), scale)


# (images, small H, small W, cs, cb): the six middle layers of the default towers at a reduced batch, odd sizes, one pixel,
# fewer than 32 big-resolution channels
LAYERS = [(20, 16, 16, 64, 32), (20, 8, 8, 128, 64), (20, 4, 4, 256, 128), (3, 5, 7, 32, 32), (2, 1, 1, 64, 96), (70, 2, 3, 36, 64), (5, 3, 5, 32, 12)]


@pytest.mark.parametrize("ni,hs,ws,cs,cb", LAYERS)
def test_down_up_wgrad_against_torch(ni, hs, ws, cs, cb):
    g = torch.Generator().manual_seed(ni * 100 + cs)
    big = torch.randn(ni, cb, 2 * hs, 2 * ws, generator=g).to(DEV)
    small = torch.randn(ni, cs, hs, ws, generator=g).to(DEV)
    kern = (torch.randn(cs, cb, 4, 4, generator=g) * 0.1).to(DEV)
    if N.conv4s2_ok(cs, cb, "down"):
        got = N.conv4s2_down(big, kern)
        assert got.is_contiguous(memory_format=torch.channels_last)
        _close(got, F.conv2d(big.double(), kern.double(), stride=2, padding=1))
    if N.conv4s2_ok(cs, cb, "up"):
        got = N.conv4s2_up(small, kern)
        _close(got, F.conv_transpose2d(small.double(), kern.double(), stride=2, padding=1))
    # weight gradient of the convolution big -> small with upstream gradient `small`
    bigd = big.double().requires_grad_(True)
    kd = kern.double().requires_grad_(True)
    F.conv2d(bigd, kd, stride=2, padding=1).backward(small.double())
    got = N.conv4s2_wgrad(small, big)
    _close(got, kd.grad, 3e-6 * max(1.0, (ni * hs * ws / 4096) ** 0.5))
    assert torch.equal(got, N.conv4s2_wgrad(small, big))      # fixed-order reduction: same bits every time


def test_tower_with_native_convolutions_matches_the_backend(monkeypatch):
    """Encoder and decoder towers (forward and every gradient) with the six middle layers on the library's kernels against
    the same modules on the backend's convolutions."""
    from dreamer_amd import world_model as W
    torch.manual_seed(0)
    enc = W.Encoder((3, 64, 64)).to(DEV)
    dec = W.Decoder((3, 64, 64)).to(DEV)
    x = torch.rand(24, 3, 64, 64, device=DEV)
    h = torch.randn(24, 512, device=DEV)
    z = torch.nn.functional.one_hot(torch.randint(0, 32, (24, 32), device=DEV), 32).float()

    def run():
        for p in list(enc.parameters()) + list(dec.parameters()):
            p.grad = None
        post = enc(x, h)
        img = dec(h, z)
        (post.square().mean() + (img - x).square().mean()).backward()
        return post.detach(), img.detach(), [p.grad.clone() for p in list(enc.parameters()) + list(dec.parameters())]

    monkeypatch.setattr(W, "NATIVE_CONV_MIN_IMAGES", 10 ** 9)
    p0, i0, g0 = run()
    monkeypatch.setattr(W, "NATIVE_CONV_MIN_IMAGES", 1)
    p1, i1, g1 = run()
    assert not torch.equal(p0, p1)      # different kernels ran
    _close(p1, p0.double(), 1e-5)
    _close(i1, i0.double(), 1e-5)
    for a, b in zip(g1, g0):
        _close(a, b.double(), 5e-5)


@pytest.mark.parametrize("ni,hs,ws,cs,cb", [(20, 16, 16, 64, 32), (20, 8, 8, 128, 64), (20, 4, 4, 256, 128), (3, 5, 7, 64, 96)])
def test_bf16_input_form_of_the_three_products(ni, hs, ws, cs, cb):
    """precision 1: operands rounded to bf16, fp32 accumulation -- within 1e-2 of the fp64 result, and no longer the exact form."""
    g = torch.Generator().manual_seed(ni + cs)
    big = torch.randn(ni, cb, 2 * hs, 2 * ws, generator=g).to(DEV)
    small = torch.randn(ni, cs, hs, ws, generator=g).to(DEV)
    kern = (torch.randn(cs, cb, 4, 4, generator=g) * 0.1).to(DEV)
    got = N.conv4s2_down(big, kern, bf16=True)
    _close(got, F.conv2d(big.double(), kern.double(), stride=2, padding=1), 1e-2)
    assert not torch.equal(got, N.conv4s2_down(big, kern))
    _close(N.conv4s2_up(small, kern, bf16=True), F.conv_transpose2d(small.double(), kern.double(), stride=2, padding=1), 1e-2)
    bigd = big.double().requires_grad_(True)
    kd = kern.double().requires_grad_(True)
    F.conv2d(bigd, kd, stride=2, padding=1).backward(small.double())
    _close(N.conv4s2_wgrad(small, big, bf16=True), kd.grad, 1e-2)
