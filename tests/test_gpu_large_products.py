"""The batched Linear products that are large enough for the large-row kernels (world_model._PRODUCT_FORM): the split form
(fp32 accuracy on the bf16 matrix pipe) and the bf16 form against an fp64 product -- forward, input gradient and weight
gradient of a plain Linear and of a Linear -> LayerNorm -> SiLU block at a shape that takes the large-row path."""
import pytest
import torch
import torch.nn.functional as F

from dreamer_amd import _native as N
from dreamer_amd import world_model as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROWS, K_IN, N_OUT = 1024, 2048, 4096      # all three products of this layer take the large-row path


def _rel(a, r):
    return float((a.double() - r).abs().max() / (r.abs().max() + 1e-30))


def _run(form, fn, *tensors):
    leaves = [t.detach().clone().requires_grad_(True) for t in tensors[:-1]]
    dy = tensors[-1]
    before = W._PRODUCT_FORM
    W._PRODUCT_FORM = form
    try:
        y = fn(*leaves)
        grads = torch.autograd.grad(y, leaves, dy)
    finally:
        W._PRODUCT_FORM = before
    return (y.detach(), *grads)


@pytest.mark.parametrize("form,tol", [("f32", 1e-5), ("split", 1e-5), ("bf16", 1e-2)])
def test_plain_linear_in_every_form(form, tol):
    assert W._large_product_ok(ROWS, N_OUT, K_IN) and W._large_product_ok(ROWS, K_IN, N_OUT) and W._large_product_ok(N_OUT, K_IN, ROWS)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(ROWS, K_IN, generator=g).to(DEV)
    w = (torch.randn(N_OUT, K_IN, generator=g) / K_IN ** 0.5).to(DEV)
    b = (0.1 * torch.randn(N_OUT, generator=g)).to(DEV)
    dy = torch.randn(ROWS, N_OUT, generator=g).to(DEV)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(xd, wd, bd)
    ref_g = torch.autograd.grad(ref, (xd, wd, bd), dy.double())
    out = _run(form, W._LinearFn.apply, x, w, b, dy)
    for name, a, r in zip(("y", "dx", "dw", "db"), out, (ref.detach(), *ref_g)):
        assert _rel(a, r) < tol, (form, name, _rel(a, r))
    if form == "bf16":
        assert _rel(out[0], ref.detach()) > 1e-5      # the reduced-precision product really ran


@pytest.mark.parametrize("form,tol", [("split", 2e-5), ("bf16", 2e-2)])
def test_linear_norm_act_block_in_every_form(form, tol):
    """5120 -> 512 at 1024 rows: the forward product is too narrow for the large-row path, both gradients take it."""
    rows, k_in, width = 1024, 5120, 512
    assert not W._large_product_ok(rows, width, k_in) and W._large_product_ok(rows, k_in, width) and W._large_product_ok(width, k_in, rows)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(rows, k_in, generator=g).to(DEV)
    w = (torch.randn(width, k_in, generator=g) / k_in ** 0.5).to(DEV)
    b = (0.3 * torch.randn(width, generator=g)).to(DEV)
    ga = (1.0 + 0.2 * torch.randn(width, generator=g)).to(DEV)
    be = (0.1 * torch.randn(width, generator=g)).to(DEV)
    dy = torch.randn(rows, width, generator=g).to(DEV)
    dd = [t.double().requires_grad_(True) for t in (x, w, b, ga, be)]
    ref = F.silu(F.layer_norm(F.linear(dd[0], dd[1], dd[2]), (width,), dd[3], dd[4], 1e-5))
    ref_g = torch.autograd.grad(ref, dd, dy.double())
    out = _run(form, lambda *t: W._LinearNormActFn.apply(*t, 1e-5), x, w, b, ga, be, dy)
    for name, a, r in zip(("y", "dx", "dw", "db", "dgamma", "dbeta"), out, (ref.detach(), *ref_g)):
        assert _rel(a, r) < tol, (form, name, _rel(a, r))


@pytest.mark.parametrize("M,Nn,K,ta", [(4096, 1536, 1024, True), (1024, 4096, 1024, True), (1024, 4096, 1024, False), (132, 260, 96, True),
                                       (70, 36, 64, False), (8, 4, 5, True), (300, 516, 1000, True), (257, 388, 32, False)])
def test_gradient_products_without_transposed_copies(M, Nn, K, ta):
    """dW = dY^T X and dX = dY W on the split kernel's staging against fp64: full tiles, ragged tiles, a split reduction."""
    g = torch.Generator().manual_seed(M + Nn + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    b = torch.randn(K, Nn, generator=g).to(DEV)
    assert N.gemm_split_ok(a, b, ta)
    got = N.gemm_split(a, b, ta)
    ref = (a.double().t() if ta else a.double()) @ b.double()
    assert got.shape == (M, Nn)
    assert float((got.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max()) * max(1.0, (K / 1024) ** 0.5)
    assert torch.equal(got, N.gemm_split(a, b, ta))      # fixed-order reduction


@pytest.mark.parametrize("M,Nn,K,ta", [(4096, 1536, 1024, True), (1024, 4096, 1024, False), (132, 260, 96, True)])
def test_gradient_products_with_bf16_inputs(M, Nn, K, ta):
    g = torch.Generator().manual_seed(M + Nn + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    b = torch.randn(K, Nn, generator=g).to(DEV)
    got = N.gemm_split(a, b, ta, bf16=True)
    ref = (a.double().t() if ta else a.double()) @ b.double()
    assert float((got.double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
    assert not torch.equal(got, N.gemm_split(a, b, ta))
