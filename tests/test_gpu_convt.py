"""Direct ConvTranspose2d(k4, s2, p1) kernel for few output channels against torch (forward values, and the
decoder's gradients through the autograd wrapper that uses it)."""
import pytest
import torch
import torch.nn.functional as F

from dreamer_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,cin,cout,h,w", [(1024, 32, 3, 32, 32), (3, 5, 1, 4, 4), (2, 8, 4, 7, 9), (5, 16, 2, 1, 1), (4, 3, 3, 16, 20)])
def test_forward_matches_torch(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(n + cin + h)
    x = torch.randn(n, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cin, cout, 4, 4, generator=g) * 0.1).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    out = N.convt_k4s2_forward(x, wt, b)
    assert out.shape == ref.shape
    assert float((out.double() - ref).abs().max()) < 1e-5 * max(float(ref.abs().max()), 1.0)
    out0 = N.convt_k4s2_forward(x, wt, None)
    ref0 = F.conv_transpose2d(x.double(), wt.double(), None, stride=2, padding=1)
    assert float((out0.double() - ref0).abs().max()) < 1e-5 * max(float(ref0.abs().max()), 1.0)


def test_decoder_with_native_last_stage_matches_torch(monkeypatch):
    from dreamer_amd import world_model as W
    torch.manual_seed(2)
    dec = W.Decoder((3, 64, 64), base_cnn_channels=8, num_latents=8, hidden_size=64).to(DEV)
    h = torch.randn(6, 64, device=DEV)
    z = F.one_hot(torch.randint(0, 32, (6, 8), device=DEV), 32).float()
    monkeypatch.setattr(W._ConvBiasFn, "native_convt", False)
    ref = dec(h, z)
    g_ref = torch.autograd.grad(ref.square().sum(), list(dec.parameters()))
    monkeypatch.setattr(W._ConvBiasFn, "native_convt", True)
    out = dec(h, z)
    g_out = torch.autograd.grad(out.square().sum(), list(dec.parameters()))
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert rel(out, ref) < 1e-5
    for (name, _), a, r in zip(dec.named_parameters(), g_out, g_ref):
        assert rel(a, r) < 2e-4, name


@pytest.mark.parametrize("n,cin,cout,h,w", [(64, 32, 3, 32, 32), (3, 32, 1, 4, 4), (2, 32, 4, 7, 9), (5, 32, 2, 1, 1), (4, 5, 3, 6, 5), (2, 8, 3, 5, 6)])
def test_input_gradient_matches_torch_in_both_memory_formats(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(7 * n + cin + h)
    x = torch.randn(n, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cin, cout, 4, 4, generator=g) * 0.1).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    gy = torch.randn(n, cout, 2 * h, 2 * w, generator=g).to(DEV)
    xd = x.double().requires_grad_(True)
    ref = F.conv_transpose2d(xd, wt.double(), b.double(), stride=2, padding=1)
    ref.backward(gy.double())
    tol = lambda r: 1e-5 * max(float(r.abs().max()), 1.0)
    for cl in (False, True):
        gx = N.convt_k4s2_input_grad(gy, wt, cl)
        assert gx.shape == x.shape
        if cl and cin % 4 == 0 and n * h * w > 1 and cin > 1:
            assert gx.is_contiguous(memory_format=torch.channels_last)
        assert float((gx.double() - xd.grad).abs().max()) < tol(xd.grad)
