"""The fused linear block (fp32 MFMA) against plain fp32 PyTorch on the host (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dreamer_amd import _native as N
from dreamer_amd.testing import hash_uniform

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(name, shape, scale=1.0, salt=0):
    return torch.from_numpy(((hash_uniform(name, shape, salt) * 2 - 1) * scale).astype(np.float32))


@pytest.mark.parametrize("M,N_,K", [(16, 512, 512), (16, 1536, 512), (1, 6, 512), (5, 255, 64), (240, 512, 512),
                                    (37, 70, 96), (16, 1024, 512), (4, 64, 64), (1000, 96, 128), (100, 70, 96), (130, 255, 320),
                                    (97, 33, 512)])
@pytest.mark.parametrize("ln", [False, True])
def test_linear_vs_torch(M, N_, K, ln):
    x = _t("x", (M, K), 2.0, M)
    w = _t("w", (N_, K), 1.0 / np.sqrt(K), N_)
    b = _t("b", (N_,), 0.1)
    g, be = 1.0 + _t("g", (K,), 0.2), _t("be", (K,), 0.1)
    a = F.silu(F.layer_norm(x, (K,), g, be, 1e-5)) if ln else x
    ref = (a.double() @ w.double().T + b.double()).float()
    y = N.fused_linear(x.to(DEV), w.to(DEV), b.to(DEV), ln=(g.to(DEV), be.to(DEV)) if ln else None)
    scale = float(ref.abs().max())
    assert float((y.cpu() - ref).abs().max()) < 2e-6 * max(scale, 1.0) * np.sqrt(K / 64)


def test_asymmetric_identity_layout():
    # X = I picks out columns of W: catches any row/column swap in the fragment maps
    K = 64
    x = torch.eye(K)[:16]
    w = torch.arange(48 * K, dtype=torch.float32).reshape(48, K)
    y = N.fused_linear(x.to(DEV), w.to(DEV))
    assert torch.equal(y.cpu(), w[:, :16].T.contiguous())


@pytest.mark.parametrize("M", [23, 150])
def test_linear_with_onehot_table_and_strided_weight(M):
    D, L, K, Nn = 128, 8, 32, 96
    x = _t("x", (M, D))
    w_full = _t("w", (Nn, D + L * K), 0.05)          # Linear over cat[h, z_flat]
    b = _t("b", (Nn,), 0.1)
    idx = torch.from_numpy((hash_uniform("i", (M, L), 5) * K).astype(np.int64))
    z = F.one_hot(idx, K).float().reshape(M, L * K)
    ref = (torch.cat([x, z], -1).double() @ w_full.double().T + b.double()).float()
    table = w_full[:, D:].T.contiguous()
    wd = w_full.to(DEV)
    y = N.fused_linear(x.to(DEV), wd[:, :D], b.to(DEV), table=table.to(DEV), zidx=idx.to(torch.uint8).to(DEV))
    assert float((y.cpu() - ref).abs().max()) < 3e-6 * float(ref.abs().max())


def test_linear_rejects_bad_shapes():
    x = torch.zeros(4, 48, device=DEV)
    w = torch.zeros(8, 48, device=DEV)
    with pytest.raises(N.NativeError):
        N.fused_linear(x, w)   # K not a multiple of 32


@pytest.mark.parametrize("M,N,K", [(300, 192, 64), (4096, 512, 512), (1000, 1100, 1536), (37, 40, 96)])
def test_bf16_input_product_matches_rounded_reference(M, N, K):
    """bf16 matrix inputs, fp32 accumulation: equal to the fp64 product of the bf16-rounded operands to fp32
    accumulation error, and to the fp32 product within the bf16 tolerance (1e-2)."""
    from dreamer_amd import _native as N_
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    y = N_.fused_linear(x.cuda(), w.cuda(), b.cuda(), bf16=True).cpu()
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    ref_rounded = (xr @ wr.T + b.double()).float()
    ref_f32 = (x.double() @ w.double().T + b.double()).float()
    assert float((y - ref_rounded).abs().max() / ref_rounded.abs().max()) < 2e-5
    assert float((y - ref_f32).abs().max() / ref_f32.abs().max()) < 1e-2


@pytest.mark.parametrize("M,N,K", [(300, 192, 64), (4096, 512, 512), (1000, 1100, 1536), (37, 40, 96), (64, 128, 32)])
def test_split_product_has_fp32_accuracy(M, N, K):
    """fp32 product on the bf16 matrix pipe (exact three-way split of both operands, six partial products):
    as close to the fp64 product as the fp32 MFMA form is, far below the bf16 tolerance."""
    from dreamer_amd import _native as N_
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * 3
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = (x.double() @ w.double().T + b.double())
    y = N_.fused_linear(x.cuda(), w.cuda(), b.cuda(), split=True).cpu().double()
    err_split = float((y - ref).abs().max() / ref.abs().max())
    assert err_split < 2e-6 * np.sqrt(K / 64)
    if M >= 256:    # rows the fp32 MFMA form of the same kernel family takes
        y32 = N_.fused_linear(x.cuda(), w.cuda(), b.cuda()).cpu().double()
        err_f32 = float((y32 - ref).abs().max() / ref.abs().max())
        assert err_split < 4 * err_f32 + 1e-7


def test_split_product_layout_and_table():
    # X = I picks out columns of W: the split must reproduce fp32 values exactly (each is h + m + l)
    K = 64
    x = torch.eye(K)[:48]
    w = (torch.arange(160 * K, dtype=torch.float32).reshape(160, K) * 1.000123 + 0.37)
    y = N.fused_linear(x.to(DEV), w.to(DEV), split=True)
    assert torch.equal(y.cpu(), w[:, :48].T.contiguous())
    M, D, L, Kc, Nn = 150, 128, 8, 32, 96
    xx = _t("x", (M, D))
    w_full = _t("w", (Nn, D + L * Kc), 0.05)
    b = _t("b", (Nn,), 0.1)
    idx = torch.from_numpy((hash_uniform("i", (M, L), 5) * Kc).astype(np.int64))
    z = F.one_hot(idx, Kc).float().reshape(M, L * Kc)
    ref = (torch.cat([xx, z], -1).double() @ w_full.double().T + b.double()).float()
    wd = w_full.to(DEV)
    y = N.fused_linear(xx.to(DEV), wd[:, :D], b.to(DEV), table=w_full[:, D:].T.contiguous().to(DEV),
                       zidx=idx.to(torch.uint8).to(DEV), split=True)
    assert float((y.cpu() - ref).abs().max()) < 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(16, 100, 1056), (5, 33, 1028), (40, 257, 2048), (64, 48, 544), (1, 16, 4096), (23, 700, 520)])
def test_few_rows_long_inner_dimension(M, N, K):
    """At most 64 rows over an inner dimension beyond 512: the memory-bound matrix-core kernel (16-row tiles, K split over
    the waves of a workgroup).  Ragged rows, columns that are no multiple of the 16-wide tile and inner dimensions that are
    no multiple of its 64-wide blocks; fp32 to fp32 accumulation error, the bf16 form equal to the product of the rounded
    operands (K a multiple of 8 or not: with and without the packed 16-byte reads)."""
    from dreamer_amd import _native as N_
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = (x.double() @ w.double().T + b.double()).float()
    y = N_.fused_linear(x.cuda(), w.cuda(), b.cuda()).cpu()
    assert float((y - ref).abs().max() / ref.abs().max()) < 3e-6
    yb = N_.fused_linear(x.cuda(), w.cuda(), b.cuda(), bf16=True).cpu()
    ref_rounded = (x.bfloat16().double() @ w.bfloat16().double().T + b.double()).float()
    assert float((yb - ref_rounded).abs().max() / ref_rounded.abs().max()) < 2e-5
    # strided operands: a column window of a wider weight and of a wider input
    wide_w = torch.randn(N, K + 64, generator=g).cuda()
    wide_x = torch.randn(M, K + 32, generator=g).cuda()
    y2 = N_.fused_linear(wide_x[:, 32:], wide_w[:, 64:], b.cuda()).cpu()
    ref2 = (wide_x[:, 32:].double().cpu() @ wide_w[:, 64:].double().cpu().T + b.double()).float()
    assert float((y2 - ref2).abs().max() / ref2.abs().max()) < 3e-6
