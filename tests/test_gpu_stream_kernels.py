"""HIP lambda-return, replay gather and categorical sampling against the oracle (GPU)."""
import numpy as np
import pytest
import torch

import oracle as O
from dreamer_amd import _native as N
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.testing import hash_exponential, hash_uniform, synthetic_frames
from helpers import load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_arch_is_gfx950():
    torch.cuda.init()
    N.check(N.lib().drm_arch_ok(), "drm_arch_ok")


def test_lambda_return_golden_bit_exact():
    g = load("pointwise.npz")
    R, C, V, VL = (torch.from_numpy(g[k]).to(DEV) for k in ("lr_R", "lr_C", "lr_V", "lr_VL"))
    out = N.lambda_return(R, C, V, VL, 0.997, 0.95)
    assert torch.equal(out.cpu(), torch.from_numpy(g["lr_ret"]))


@pytest.mark.parametrize("B,H", [(1, 1), (16, 15), (300, 15), (4096, 15), (7, 16), (33, 64), (5, 333), (70, 22), (37, 43), (130, 85), (19, 170), (70, 401)])
def test_lambda_return_vs_oracle_bit_exact(B, H):
    R = torch.from_numpy(((hash_uniform("R", (B, H), B) - 0.5) * 6).astype(np.float32))
    C = torch.from_numpy(hash_uniform("C", (B, H), B).astype(np.float32))
    V = torch.from_numpy(((hash_uniform("V", (B, H), B) - 0.5) * 40).astype(np.float32))
    VL = torch.from_numpy(((hash_uniform("VL", (B,), B) - 0.5) * 40).astype(np.float32))
    ref = O.lambda_return(R, C, V, VL, 0.997, 0.95)
    out = N.lambda_return(R.to(DEV), C.to(DEV), V.to(DEV), VL.to(DEV), 0.997, 0.95)
    assert torch.equal(out.cpu(), ref)


def test_lambda_return_linearity_full_size():
    # returns are affine in (R, V, V_last) for fixed C: f(aX) = a f(X) up to rounding
    B, H = 4096, 15
    gen = torch.Generator().manual_seed(3)
    R, V = torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen)
    C, VL = torch.rand(B, H, generator=gen), torch.randn(B, generator=gen)
    f = lambda s: N.lambda_return((R * s).to(DEV), C.to(DEV), (V * s).to(DEV), (VL * s).to(DEV), 0.997, 0.95).cpu()
    assert torch.allclose(f(2.0), 2.0 * f(1.0), rtol=0, atol=0)  # scaling by 2 is exact in fp32
    assert torch.allclose(f(3.0), 3.0 * f(1.0), rtol=1e-5, atol=1e-5)


def _filled(cap, obs_shape, T, n, device):
    frames = synthetic_frames(n, obs_shape, salt=cap)
    acts = (hash_uniform("a", (n,), cap) * 6).astype(np.int64)
    rews = ((hash_uniform("r", (n,), cap) - 0.5) * 8).astype(np.float32)
    conts = hash_uniform("c", (n,), cap) > 0.1
    dev_buf = ReplayBuffer(cap, obs_shape, T, torch.device(device))
    ora = O.ReplayOracle(cap, obs_shape, T)
    dev_buf.store_many(frames, acts, rews, conts)
    for i in range(n):
        ora.store(frames[i], acts[i], rews[i], conts[i])
    assert (dev_buf.idx, dev_buf.full) == (ora.idx, ora.full)
    return dev_buf, ora


@pytest.mark.parametrize("cap,n,T,B,shape", [(50, 37, 5, 6, (3, 8, 8)), (40, 40, 8, 6, (3, 8, 8)), (32, 75, 6, 9, (3, 8, 8)),
                                              (16, 16, 16, 4, (1, 4, 4)), (500, 1200, 64, 16, (3, 64, 64))])
def test_gather_bit_exact_vs_oracle(cap, n, T, B, shape):
    dev_buf, ora = _filled(cap, shape, T, n, DEV)
    for seed in (0, 42):
        np.random.seed(seed)
        ref = ora.sample(B)
        np.random.seed(seed)
        got = dev_buf.sample(B, want_f32=True)
        assert np.array_equal(got["observations"].cpu().numpy(), ref["observations"])
        assert np.array_equal(got["actions"].cpu().numpy(), ref["actions"])
        assert np.array_equal(got["rewards"].cpu().numpy(), ref["rewards"])
        assert np.array_equal(got["continues"].cpu().numpy(), ref["continues"])
        f32 = torch.from_numpy(ref["observations"]).float() / 255.0   # lib/trainer.py:163
        assert torch.equal(got["observations_f32"].cpu(), f32)


def test_gather_golden_and_seam():
    g = load("replay.npz")
    cap, nstore, T = (int(v) for v in g["wrapped_meta"][:3])
    shape = (3, 8, 8)
    fbytes = int(np.prod(shape))
    buf = ReplayBuffer(cap, shape, T, torch.device(DEV))
    for i in range(nstore):
        frame = ((np.arange(fbytes, dtype=np.int64) * 7 + i * 13) % 251).astype(np.uint8).reshape(shape)
        buf.store(frame, i % 5, float(i) * 0.25 - 3.0, (i % 7) != 3)
    np.random.seed(42)
    got = buf.sample(6)
    assert np.array_equal(got["observations"].cpu().numpy(), g["wrapped_s42_obs"])
    assert np.array_equal(got["actions"].cpu().numpy(), g["wrapped_s42_act"])
    # every start, including the ones that straddle the end of the ring and the write head
    all_starts = np.arange(cap, dtype=np.int64)
    out = buf.gather(all_starts)
    idx = (all_starts[:, None] + np.arange(T)[None, :]) % cap
    assert torch.equal(out["observations"].cpu(), buf.observations.cpu()[torch.from_numpy(idx)])


def test_gather_all_byte_values_divide_exactly():
    shape = (1, 16, 16)
    buf = ReplayBuffer(4, shape, 1, torch.device(DEV))
    frame = np.arange(256, dtype=np.uint8).reshape(shape)
    for _ in range(4):
        buf.store(frame, 0, 0.0, True)
    out = buf.gather(np.array([0, 3], dtype=np.int64), want_f32=True)
    ref = torch.arange(256, dtype=torch.uint8).float() / 255.0
    assert torch.equal(out["observations_f32"].cpu().reshape(2, 256)[1], ref)


@pytest.mark.parametrize("rows", [1, 7, 32, 512, 16 * 32, 4096 * 32])
def test_sampling_vs_oracle(rows):
    logits = torch.from_numpy(((hash_uniform("lg", (rows, 32), rows) - 0.5) * 12).astype(np.float32))
    e = torch.from_numpy(hash_exponential("e", (rows, 32), rows))
    lp = O.log_unimix(logits, 0.01)
    z_ref, idx_ref, y_ref = O.gumbel_hard(lp, e)
    z, idx, y = N.sample_categorical_st(logits.to(DEV), e.to(DEV), 0.01, want_y=True)
    top2 = (lp - e.log()).topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(idx.cpu().long()[clear], idx_ref[clear])
    assert float((idx.cpu().long() != idx_ref).float().mean()) < 1e-4
    assert torch.allclose(y.cpu(), y_ref, rtol=2e-5, atol=1e-7)
    zc = z.cpu()
    assert torch.equal(zc.sum(-1), torch.ones(rows)) and torch.equal(zc.argmax(-1), idx.cpu().long())
    assert set(zc.unique().tolist()) <= {0.0, 1.0}


def test_sampling_golden_and_plain_logits():
    g = load("pointwise.npz")
    logits, e = torch.from_numpy(g["logits"]), torch.from_numpy(g["noise"])
    z, idx, _ = N.sample_categorical_st(logits.to(DEV), e.to(DEV), 0.01)
    assert torch.equal(z.cpu(), torch.from_numpy(g["z"]))
    # constant logits without unimix: the reset / init draw of lib/world_model.py:59, :412
    flat = torch.full((64, 32), 1.0 / 32)
    z2, idx2, _ = N.sample_categorical_st(flat.to(DEV), e.to(DEV), None)
    z_ref, idx_ref, _ = O.gumbel_hard(flat, e)
    assert torch.equal(idx2.cpu().long(), idx_ref) and torch.equal(z2.cpu(), z_ref)


def test_onehot_to_index():
    idx = torch.from_numpy((hash_uniform("oh", (777,), 1) * 32).astype(np.int64))
    z = torch.nn.functional.one_hot(idx, 32).float().to(DEV)
    assert torch.equal(N.onehot_to_index(z).cpu().long(), idx)


@pytest.mark.parametrize("na,nb", [(16 * 512, 16 * 32), (5, 3), (4096 * 512, 4096 * 32), (0, 7)])
def test_copy_pair_is_bit_exact(na, nb):
    g = torch.Generator().manual_seed(na + nb)
    a = torch.randn(na, generator=g).to(DEV)
    b = torch.randint(0, 256, (nb,), generator=g, dtype=torch.uint8).to(DEV)
    a_out, b_out = torch.full_like(a, -1.0), torch.zeros_like(b)
    N.copy_pair(a_out, a, b_out, b)
    assert torch.equal(a_out, a) and torch.equal(b_out, b)
