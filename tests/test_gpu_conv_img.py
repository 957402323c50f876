"""The image-side ends of the conv towers (first encoder layer, last decoder layer) on the library's kernels against
fp64 host references: forward of both layers and the weight gradient they share."""
import pytest
import torch
import torch.nn.functional as F

from dreamer_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (images, feature channels, image channels, H, W) with the image at (2H, 2W)
SHAPES = [(20, 32, 3, 32, 32), (3, 64, 1, 5, 7), (2, 32, 4, 1, 1), (5, 96, 2, 2, 3), (7, 32, 3, 3, 1), (1, 128, 3, 4, 4),
          (33, 32, 3, 1, 2)]


def _rel(a, r):
    return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))


def _case(n, cs, cimg, h, w, seed=0):
    g = torch.Generator().manual_seed(seed + n * 131 + cs)
    big = torch.randn(n, cimg, 2 * h, 2 * w, generator=g)
    small = torch.randn(n, cs, h, w, generator=g)
    weight = torch.randn(cs, cimg, 4, 4, generator=g) / (4.0 * cimg ** 0.5)
    bias = torch.randn(cimg, generator=g)
    return big, small, weight, bias


@pytest.mark.parametrize("n,cs,cimg,h,w", SHAPES)
def test_down_matches_conv2d(n, cs, cimg, h, w):
    big, _, weight, _ = _case(n, cs, cimg, h, w)
    ref = F.conv2d(big.double(), weight.double(), None, stride=2, padding=1)
    out = N.conv_img_down(big.to(DEV), weight.to(DEV))
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize("n,cs,cimg,h,w", SHAPES + [(2, 64, 3, 9, 33), (1, 32, 3, 17, 70)])
def test_up_matches_conv_transpose2d(n, cs, cimg, h, w):
    _, small, weight, bias = _case(n, cs, cimg, h, w)
    ref = F.conv_transpose2d(small.double(), weight.double(), bias.double(), stride=2, padding=1)
    x = small.to(DEV).contiguous(memory_format=torch.channels_last)
    out = N.conv_img_up(x, weight.to(DEV), bias.to(DEV))
    assert out.shape == ref.shape and out.is_contiguous()
    assert _rel(out, ref) < 1e-5
    out0 = N.conv_img_up(x, weight.to(DEV), None)
    assert _rel(out0, ref - bias.double().view(1, -1, 1, 1)) < 1e-5


@pytest.mark.parametrize("n,cs,cimg,h,w", SHAPES)
def test_weight_gradient_matches_autograd_of_both_layers(n, cs, cimg, h, w):
    big, small, weight, _ = _case(n, cs, cimg, h, w)
    # as the conv's weight gradient: small is the output gradient, big the input
    wd = weight.double().requires_grad_(True)
    F.conv2d(big.double(), wd, None, stride=2, padding=1).backward(small.double())
    got = N.conv_img_wgrad(small.to(DEV).contiguous(memory_format=torch.channels_last), big.to(DEV))
    assert got.shape == weight.shape and _rel(got, wd.grad) < 1e-5
    # as the transposed conv's weight gradient: small is the input, big the output gradient -- the same sum
    wt = weight.double().requires_grad_(True)
    F.conv_transpose2d(small.double(), wt, None, stride=2, padding=1).backward(big.double())
    assert _rel(got, wt.grad) < 1e-5


def test_weight_gradient_is_reproducible_and_large_batch():
    """The training shape (1024 frames of 32 x 32 x 32 features): partial sums are added in a fixed order, so two calls
    agree bit for bit; against an fp64 reference on a subset of the images the fp32 sum over a million positions holds
    1e-5 of the largest entry."""
    n, cs, cimg, h, w = 1024, 32, 3, 32, 32
    g = torch.Generator().manual_seed(5)
    big = torch.randn(n, cimg, 2 * h, 2 * w, generator=g).to(DEV)
    small = torch.randn(n, cs, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    a, b = N.conv_img_wgrad(small, big), N.conv_img_wgrad(small, big)
    assert torch.equal(a, b)
    wd = torch.zeros(cs, cimg, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv2d(big.double().cpu(), wd, None, stride=2, padding=1).backward(small.double().cpu())
    assert _rel(a, wd.grad) < 1e-5
    out = N.conv_img_down(big, torch.randn(cs, cimg, 4, 4, generator=g).to(DEV))
    assert out.shape == (n, cs, h, w)


def test_shapes_outside_the_kernels_range_are_refused():
    x = torch.zeros(2, 3, 8, 8, device=DEV)
    with pytest.raises(N.NativeError):
        N.conv_img_down(x, torch.zeros(16, 3, 4, 4, device=DEV))           # 16 feature channels
    with pytest.raises(N.NativeError):
        N.conv_img_down(torch.zeros(2, 5, 8, 8, device=DEV), torch.zeros(32, 5, 4, 4, device=DEV))   # 5 image channels
    with pytest.raises(N.NativeError):
        N.conv_img_up(torch.zeros(2, 48, 4, 4, device=DEV).contiguous(memory_format=torch.channels_last),
                      torch.zeros(48, 3, 4, 4, device=DEV), None)          # 48 feature channels
