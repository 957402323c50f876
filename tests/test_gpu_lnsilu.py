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


@pytest.mark.parametrize("N,C,HW", [(64, 32, (32, 32)), (9, 8, (4, 4)), (130, 16, (8, 8)), (3, 4, (2, 2)),
                                    (261, 32, (16, 16)), (300, 256, (4, 4)), (257, 8, (64, 64))])
def test_channel_bias_variant_matches_bias_add_then_layer_norm_silu(N, C, HW):
    """SiLU(LayerNorm(x + b[c])) with the producing convolution's bias folded into the kernels, and its gradient
    (the last three cases: many long samples, which take the passes that walk the samples)."""
    g = torch.Generator().manual_seed(N + C)
    shape = (C, *HW)
    x = (torch.randn(N, *shape, generator=g) * 1.3).to(DEV).requires_grad_(True)
    w = (1.0 + 0.2 * torch.randn(*shape, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(*shape, generator=g)).to(DEV).requires_grad_(True)
    cb = (0.5 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(N, *shape, generator=g).to(DEV)
    ref = F.silu(F.layer_norm(x + cb.view(1, C, 1, 1), shape, w, b, 1e-5))
    gx, gw, gb, gc = torch.autograd.grad(ref, (x, w, b, cb), dy)
    out = _LayerNormSiLUFn.apply(x, w, b, 1e-5, cb)
    hx, hw, hb, hc = torch.autograd.grad(out, (x, w, b, cb), dy)
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert rel(out, ref) < 2e-6
    assert rel(hx, gx) < 1e-5 and rel(hw, gw) < 1e-5 and rel(hb, gb) < 1e-5
    assert rel(hc, gc) < 5e-5   # a sum of N*H*W values that largely cancel


def test_tower_with_folded_conv_bias_matches_plain_sequential(monkeypatch):
    from dreamer_amd import world_model as W
    torch.manual_seed(4)
    enc = W.Encoder((3, 64, 64), base_cnn_channels=8).to(DEV)
    x = torch.rand(6, 3, 64, 64, device=DEV)
    monkeypatch.setattr(W._Tower, "fuse", False)
    ref = enc.conv(x)
    g_ref = torch.autograd.grad(ref.square().sum(), list(enc.conv.parameters()))
    monkeypatch.setattr(W._Tower, "fuse", True)
    out = enc.conv(x)
    g_out = torch.autograd.grad(out.square().sum(), list(enc.conv.parameters()))
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert rel(out, ref) < 1e-5
    for (name, _), a, r in zip(enc.conv.named_parameters(), g_out, g_ref):
        assert rel(a, r) < 2e-4, name


@pytest.mark.parametrize("rows,k,width", [(1024, 1536, 512), (240, 512, 512), (7, 96, 64), (1, 32, 2048), (50, 64, 100)])
def test_fused_mlp_block_matches_linear_layer_norm_silu(rows, k, width):
    """Linear -> LayerNorm -> SiLU block (lib/world_model.py:153-160) with the library's row kernels: values and all gradients."""
    from dreamer_amd.world_model import _LinearNormActFn
    g = torch.Generator().manual_seed(rows + width)
    x = torch.randn(rows, k, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(width, k, generator=g) / k ** 0.5).to(DEV).requires_grad_(True)
    b = (0.3 * torch.randn(width, generator=g)).to(DEV).requires_grad_(True)
    ga = (1.0 + 0.2 * torch.randn(width, generator=g)).to(DEV).requires_grad_(True)
    be = (0.1 * torch.randn(width, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(rows, width, generator=g).to(DEV)
    ref = F.silu(F.layer_norm(F.linear(x, w, b), (width,), ga, be, 1e-5))
    g_ref = torch.autograd.grad(ref, (x, w, b, ga, be), dy)
    out = _LinearNormActFn.apply(x, w, b, ga, be, 1e-5)
    g_out = torch.autograd.grad(out, (x, w, b, ga, be), dy)
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert rel(out, ref) < 5e-6
    for name, a, r in zip("x w b gamma beta".split(), g_out, g_ref):
        assert rel(a, r) < 3e-5, name


@pytest.mark.parametrize("N,C,HW,with_cb", [(64, 32, (32, 32), True), (9, 8, (4, 4), True), (130, 16, (8, 8), False), (3, 4, (2, 3), True),
                                            (5, 256, (4, 4), False), (261, 32, (16, 16), True), (300, 256, (4, 4), True),
                                            (1024, 64, (16, 16), True), (259, 128, (8, 8), False)])
def test_channels_last_samples_match_torch(N, C, HW, with_cb):
    """The same block on channels-last activations (parameters stay in (C, H, W) order)."""
    g = torch.Generator().manual_seed(N + C)
    shape = (C, *HW)
    x = (torch.randn(N, *shape, generator=g) * 1.3).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (1.0 + 0.2 * torch.randn(*shape, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(*shape, generator=g)).to(DEV).requires_grad_(True)
    cb = (0.5 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(N, *shape, generator=g).to(DEV)
    pre = x + cb.view(1, C, 1, 1) if with_cb else x
    ref = F.silu(F.layer_norm(pre, shape, w, b, 1e-5))
    wrt = (x, w, b, cb) if with_cb else (x, w, b)
    g_ref = torch.autograd.grad(ref, wrt, dy)
    out = _LayerNormSiLUFn.apply(x, w, b, 1e-5, cb if with_cb else None)
    assert out.is_contiguous(memory_format=torch.channels_last)
    g_out = torch.autograd.grad(out, wrt, dy)
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert rel(out, ref) < 2e-6
    for name, a, r in zip("x gamma beta cbias".split(), g_out, g_ref):
        assert rel(a, r) < (5e-5 if name == "cbias" else 1e-5), name


@pytest.mark.parametrize("which", ["encoder", "decoder"])
def test_channels_last_towers_match_plain_sequential(monkeypatch, which):
    from dreamer_amd import world_model as W
    torch.manual_seed(4)
    n = 70
    if which == "encoder":
        mod = W.Encoder((3, 64, 64), base_cnn_channels=32).to(DEV).conv
        args = (torch.rand(n, 3, 64, 64, device=DEV),)
        fn = lambda: mod(*args)
    else:
        dec = W.Decoder((3, 64, 64), base_cnn_channels=32, num_latents=8, hidden_size=64).to(DEV)
        h = torch.randn(n, 64, device=DEV)
        z = F.one_hot(torch.randint(0, 32, (n, 8), device=DEV), 32).float()
        mod, fn = dec, (lambda: dec(h, z))
    monkeypatch.setattr(W._Tower, "fuse", False)
    ref = fn()
    g_ref = torch.autograd.grad(ref.square().sum(), list(mod.parameters()))
    monkeypatch.setattr(W._Tower, "fuse", True)
    monkeypatch.setattr(W._Tower, "channels_last", True)
    out = fn()
    g_out = torch.autograd.grad(out.square().sum(), list(mod.parameters()))
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    assert out.shape == ref.shape and rel(out, ref) < 1e-5
    for (name, _), a, r in zip(mod.named_parameters(), g_out, g_ref):
        assert rel(a, r) < 2e-4, name
