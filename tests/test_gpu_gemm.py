"""The general fp32 MFMA product (forward, input-gradient and weight-gradient forms of a Linear layer) against fp64
products of the same operands (GPU)."""
import pytest
import torch

from dreamer_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(1024, 512, 512), (1024, 512, 1536), (1024, 4096, 1536), (6, 64, 96), (70, 130, 36), (1, 1, 4), (333, 1, 512),
          (512, 1536, 1024), (512, 1088, 1024), (64, 64, 8192), (24, 1024, 64), (1024, 1024, 4608)]


@pytest.mark.parametrize("M,Nn,K", SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_forms_against_fp64(M, Nn, K, ta, tb):
    if (not ta or not tb) and K % 4:
        pytest.skip("k-contiguous operands need K % 4 == 0")
    g = torch.Generator().manual_seed(M * 7 + Nn * 3 + K)
    # leading dimensions padded to multiples of 4 (and a little beyond): views with a row stride
    def operand(rows, cols):
        ld = (cols + 3) // 4 * 4 + 4
        return torch.randn(rows, ld, generator=g).to(DEV)[:, :cols]
    a = operand(K, M) if ta else operand(M, K)
    b = operand(K, Nn) if tb else operand(Nn, K)
    bias = torch.randn(Nn, generator=g).to(DEV)
    assert N.gemm_ok(a, b, ta, tb)
    got = N.gemm(a, b, ta, tb, bias=bias)
    A = (a.t() if ta else a).double()
    B = (b if tb else b.t()).double()
    ref = A @ B + bias.double()
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, (K / 512) ** 0.5)
    # deterministic: the same call gives the same bits (fixed-order split-K reduction)
    assert torch.equal(got, N.gemm(a, b, ta, tb, bias=bias))


def test_linear_autograd_function_matches_torch():
    from dreamer_amd.world_model import _linear
    torch.manual_seed(0)
    x = torch.randn(1024, 1536, device=DEV, requires_grad=True)
    lin = torch.nn.Linear(1536, 512).to(DEV)
    y = _linear(x, lin.weight, lin.bias)
    ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, lin.weight, lin.bias), gy)
    rx, rw, rb = torch.autograd.grad(ref, (x, lin.weight, lin.bias), gy.double())
    for got, want in ((gx, rx), (gw, rw), (gb, rb)):
        assert float((got.double() - want.double()).abs().max()) <= 3e-6 * float(want.abs().max())


@pytest.mark.parametrize("rows,cols,ld", [(1024, 512, 512), (128, 1536, 1536), (32, 98304, 98304), (7, 4, 4), (1000, 20, 36), (513, 4096, 4096), (1, 8, 8)])
def test_column_sums_against_fp64(rows, cols, ld):
    """Bias gradients / slice sums: out[c] = sum_r x[r, c], also for a matrix whose rows are longer than its columns."""
    g = torch.Generator().manual_seed(rows + cols)
    full = torch.randn(rows, ld, generator=g).to(DEV)
    x = full[:, :cols]
    assert N.colsum_ok(x)
    got = N.colsum(x)
    ref = x.double().sum(0)
    assert got.shape == (cols,)
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(x.abs().sum(0).max())
    assert torch.equal(got, N.colsum(x))            # fixed order of the rows: the same bits every time
    assert torch.equal(N.sum_rows(x), got)
    odd = torch.randn(5, 7, device=DEV)             # shapes the kernel does not take fall back to torch
    assert not N.colsum_ok(odd) and torch.equal(N.sum_rows(odd), odd.sum(0))
