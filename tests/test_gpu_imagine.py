"""The native imagination block against the reference's golden vectors and the oracle (GPU)."""
import numpy as np
import pytest
import torch

import oracle as O
from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.testing import hash_exponential
from dreamer_amd.world_model import WorldModelState
from helpers import DIMS, assert_close_elementwise, build_models, expo, load, rel_err, start_state, weights_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5   # north-star tolerance for fp32 rollout states, lambda-returns and losses


def _engine(d, salt, **kw):
    wm, actor, critic = build_models(d, salt, **kw)
    return wm, actor, critic, ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))


@pytest.mark.parametrize("tag", ["small", "default"])
def test_free_running_rollout_matches_reference(tag):
    g = load(f"imagine_{tag}.npz")
    d, salt = DIMS[tag], int(g["salt"])
    wm, actor, critic, eng = _engine(d, salt)
    h0, z0, _ = start_state(d, salt)
    noise = {"a": torch.from_numpy(g["noise_a"]).to(DEV), "z": expo("imagine_z", (d["H"], d["B"], d["L"], d["K"]), salt).to(DEV)}
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), d["H"], float(g["gamma"]), float(g["lam"]), noise=noise,
                  want_prior_logits=True)
    torch.cuda.synchronize()
    assert np.array_equal(out["actions"].cpu().numpy(), g["actions"])
    assert np.array_equal(out["zidx"].cpu().numpy(), g["zidx"])
    assert torch.equal(out["z"].cpu().argmax(-1).to(torch.uint8), out["zidx"].cpu())
    assert float(out["z"].sum()) == d["B"] * d["H"] * d["L"]
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp"):
        assert rel_err(out[key].cpu(), g[key]) < TOL, key
    for key in ("h", "lam_returns", "values"):   # entry by entry, small entries included
        assert_close_elementwise(out[key].cpu(), g[key], rtol=1e-5, atol_scale=2e-6, what=key)
    if tag == "small":
        assert rel_err(out["prior_logits"].cpu(), g["prior_logits"].reshape(out["prior_logits"].shape)) < TOL
    # losses computed from the native outputs (lib/trainer.py:227-247) within the same tolerance
    Hs = WorldModelState(h=out["h"], z=out["z"])
    cl = critic.loss(Hs, out["lam_returns"])
    al = actor.loss(Hs, out["actions"], out["lam_returns"], out["values"])
    assert rel_err(cl.detach().cpu(), g["critic_loss"]) < TOL and rel_err(al.detach().cpu(), g["actor_loss"]) < TOL


@pytest.mark.parametrize("layers", [(1, 1), (3, 3), (2, 4)])
@pytest.mark.parametrize("B,H,A", [(1, 1, 7), (5, 3, 7), (16, 15, 7), (37, 4, 7), (300, 3, 7), (16, 3, 18), (100, 2, 18),
                                   (260, 2, 18), (3, 2, 2)])
def test_teacher_forced_vs_oracle(B, H, A, layers):
    d = dict(DIMS["small"], D=96, M=128, L=6, A=A)   # A = 18: the full Atari action set
    salt = 100 + B + H
    wm, actor, critic = build_models(d, salt, actor_layers=layers[0], critic_layers=layers[1])
    ww, wa, wc = weights_of(wm), weights_of(actor), weights_of(critic)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    with torch.no_grad():
        ref = O.imagine_block(ww, wa, wc, h0, z0, ea, ez, 0.997, 0.95)
    ref_zidx = ref["z"].argmax(-1).to(torch.uint8)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, noise={"a": ea.to(DEV), "z": ez.to(DEV)},
                  forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV), want_prior_logits=True,
                  want_sampled=True)
    torch.cuda.synchronize()
    assert torch.equal(out["zidx"].cpu(), ref_zidx) and torch.equal(out["actions"].cpu(), ref["actions"])
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
        assert rel_err(out[key].cpu(), ref[key]) < TOL, key
    if B >= 256:   # the default form there is the split one: the fp32 matrix-instruction form against the oracle as well
        mf = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, noise={"a": ea.to(DEV), "z": ez.to(DEV)},
                     forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV), want_prior_logits=True,
                     precision="f32_mfma")
        for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
            assert rel_err(mf[key].cpu(), ref[key]) < TOL, key
        assert not torch.equal(mf["h"], out["h"])   # two different kernels ran
    # what the sampler drew must agree with the oracle wherever the draw is not a near-tie
    lp = O.log_unimix(ref["prior_logits"], 0.01) - ez.permute(1, 0, 2, 3).log()
    top2 = lp.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(out["zidx_sampled"].cpu()[clear], ref_zidx[clear])
    sc = (ref["actor_logp"].softmax(-1) / ea.permute(1, 0, 2)).log()
    t2 = sc.topk(2, dim=-1).values
    clear_a = (t2[..., 0] - t2[..., 1]) > 1e-4
    assert torch.equal(out["actions_sampled"].cpu()[clear_a], ref["actions"][clear_a])


@pytest.mark.parametrize("B,H,D,M", [(6, 3, 1024, 768), (20, 2, 640, 1024), (260, 2, 1024, 544),
                                     (16, 3, 4096, 512), (37, 2, 1056, 256), (3, 2, 2048, 64)])
def test_wide_models_teacher_forced_vs_oracle(B, H, D, M):
    """Widths beyond 512 (a scaled-up recurrent width) run on the large-row kernels for any batch, or, at few rows with
    a hidden width of at most 512, on the K-looped few-row kernel (the last three cases)."""
    d = dict(DIMS["small"], D=D, M=M, L=6, A=5)
    salt = 500 + B
    wm, actor, critic = build_models(d, salt)
    ww, wa, wc = weights_of(wm), weights_of(actor), weights_of(critic)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    with torch.no_grad():
        ref = O.imagine_block(ww, wa, wc, h0, z0, ea, ez, 0.997, 0.95)
    ref_zidx = ref["z"].argmax(-1).to(torch.uint8)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, noise={"a": ea.to(DEV), "z": ez.to(DEV)},
                  forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV), want_prior_logits=True,
                  want_sampled=True)
    torch.cuda.synchronize()
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
        assert rel_err(out[key].cpu(), ref[key]) < TOL, key
    # the draws agree with the oracle's wherever its top-two margin is clear of fp32 rounding
    lp = O.log_unimix(ref["prior_logits"], 0.01) - ez.permute(1, 0, 2, 3).log()
    top2 = lp.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(out["zidx_sampled"].cpu()[clear], ref_zidx[clear])
    sc = (ref["actor_logp"].softmax(-1) / ea.permute(1, 0, 2)).log()
    t2 = sc.topk(2, dim=-1).values
    clear_a = (t2[..., 0] - t2[..., 1]) > 1e-4
    assert torch.equal(out["actions_sampled"].cpu()[clear_a], ref["actions"][clear_a])


def test_bf16_matrix_inputs_within_bf16_tolerance():
    """Large-row path with bf16 matrix inputs (fp32 accumulation), teacher-forced along the oracle's
    trajectory: states, rewards, values and lambda-returns within the bf16 tolerance of 1e-2."""
    d = dict(DIMS["small"], D=128, M=128, L=8, A=5)
    B, H, salt = 300, 4, 77
    wm, actor, critic = build_models(d, salt)
    ww, wa, wc = weights_of(wm), weights_of(actor), weights_of(critic)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    with torch.no_grad():
        ref = O.imagine_block(ww, wa, wc, h0, z0, ea, ez, 0.997, 0.95)
    ref_zidx = ref["z"].argmax(-1).to(torch.uint8)
    kw = dict(noise={"a": ea.to(DEV), "z": ez.to(DEV)}, forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV),
              want_prior_logits=True)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, precision="bf16", **kw)
    exact = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, **kw)
    torch.cuda.synchronize()
    worst = 0.0
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
        err = rel_err(out[key].cpu(), ref[key])
        worst = max(worst, err)
        assert err < 1e-2, (key, err)
        assert rel_err(exact[key].cpu(), ref[key]) < TOL, key
    assert worst > 1e-5   # the reduced-precision products really ran


@pytest.mark.parametrize("B,H,D,M", [(16, 3, 4096, 512), (5, 2, 1056, 256)])
def test_wide_model_few_rows_bf16_matrix_inputs(B, H, D, M):
    """Few rows of a wide model with bf16 matrix inputs (the products over the recurrent width read bf16 weight images):
    within the bf16 tolerance of the oracle, and not the fp32 result."""
    d = dict(DIMS["small"], D=D, M=M, L=6, A=5)
    salt = 700 + B
    wm, actor, critic = build_models(d, salt)
    ww, wa, wc = weights_of(wm), weights_of(actor), weights_of(critic)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    with torch.no_grad():
        ref = O.imagine_block(ww, wa, wc, h0, z0, ea, ez, 0.997, 0.95)
    ref_zidx = ref["z"].argmax(-1).to(torch.uint8)
    kw = dict(noise={"a": ea.to(DEV), "z": ez.to(DEV)}, forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV),
              want_prior_logits=True)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, precision="bf16", **kw)
    torch.cuda.synchronize()
    worst = 0.0
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
        err = rel_err(out[key].cpu(), ref[key])
        worst = max(worst, err)
        assert err < 1e-2, (key, err)
    assert worst > 1e-5   # the reduced-precision products really ran


@pytest.mark.parametrize("B,H,D,M,L", [(300, 4, 128, 128, 8), (260, 2, 1024, 544, 6), (20, 2, 640, 1024, 6)])
def test_split_products_meet_the_fp32_tolerance(B, H, D, M, L):
    """Large-row path with the matrix products formed on the bf16 pipe from an exact three-way split of both
    operands (six partial products, fp32 accumulation), teacher-forced along the oracle's trajectory: the
    fp32 tolerance of the rollout holds, as it does for the fp32 MFMA form."""
    d = dict(DIMS["small"], D=D, M=M, L=L, A=5)
    salt = 900 + B
    wm, actor, critic = build_models(d, salt)
    ww, wa, wc = weights_of(wm), weights_of(actor), weights_of(critic)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    with torch.no_grad():
        ref = O.imagine_block(ww, wa, wc, h0, z0, ea, ez, 0.997, 0.95)
    ref_zidx = ref["z"].argmax(-1).to(torch.uint8)
    kw = dict(noise={"a": ea.to(DEV), "z": ez.to(DEV)}, forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV),
              want_prior_logits=True)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, precision="f32_split", **kw)
    exact = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, precision="f32_mfma", **kw)
    torch.cuda.synchronize()
    differs = False
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
        assert rel_err(out[key].cpu(), ref[key]) < TOL, key
        differs = differs or not torch.equal(out[key], exact[key])
    assert differs   # the split kernels really ran (same values to rounding, not the same bits)


@pytest.mark.parametrize("precision", ["f32_mfma", "f32_split"])
@pytest.mark.parametrize("M", [128, 160])
def test_layernorm_on_load_with_shifted_preactivations(precision, M):
    """Rows whose mean is large against their spread (pre-activations around 10 with a standard deviation of ~0.3, as a
    trained bias can produce): the row statistics the large-row kernels hand from layer to layer are per-slice sums and
    centred sums of squares merged pairwise, so the normalisation keeps the rollout's 1e-5 tolerance (a one-pass
    E[x^2] - mean^2 loses ~1e-4 of the variance here, ten times the tolerance).  M = 160: slices of unequal width."""
    d = dict(DIMS["small"], D=96, M=M, L=6, A=5)
    salt, B, H = 1300 + M, 300, 2
    wm, actor, critic = build_models(d, salt)
    with torch.no_grad():
        for mod in (wm.dyn.mlp[0], wm.dyn.mlp[3], actor.mlp[0], critic.mlp[0], wm.rew.mlp[0], wm.rew.mlp[3], wm.cont.mlp[0]):
            mod.weight.mul_(0.3)
            mod.bias.add_(10.0)
    ww, wa, wc = weights_of(wm), weights_of(actor), weights_of(critic)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    dbl = lambda w: {k: (v.double() if v.is_floating_point() else v) for k, v in w.items()}
    with torch.no_grad():
        ref = O.imagine_block(ww, wa, wc, h0, z0, ea, ez, 0.997, 0.95)
        ref_zidx = ref["z"].argmax(-1).to(torch.uint8)
        # the same trajectory in fp64: what both fp32 evaluations approximate.  With pre-activations of 10 +- 0.3 the
        # rounding of the fp32 inputs of a LayerNorm alone is ~3e-6 of its output, so the fp32 oracle itself is up to
        # ~3e-5 away from the fp64 values here; the device path is held to the same distance (1e-5, or twice the
        # oracle's own where that is larger)
        truth = O.imagine_block(dbl(ww), dbl(wa), dbl(wc), h0.double(), z0.double(), ea.double(), ez.double(), 0.997, 0.95,
                                forced_actions=ref["actions"], forced_zidx=ref_zidx)
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, precision=precision,
                  noise={"a": ea.to(DEV), "z": ez.to(DEV)}, forced_actions=ref["actions"].to(DEV), forced_zidx=ref_zidx.to(DEV),
                  want_prior_logits=True)
    torch.cuda.synchronize()
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns", "actor_logp", "prior_logits"):
        own = rel_err(ref[key], truth[key])
        got = rel_err(out[key].cpu(), truth[key])
        assert got < max(TOL, 2.0 * own), (key, got, own)


def _check_rows_against_oracle(wm, actor, critic, h0, z0, ea, ez, got, drew=True):
    """A few rows of a device rollout against the oracle, unconditionally: the oracle is led along the device's own
    trajectory (its actions and latent classes forced), so states, heads and returns must agree to the tolerance
    whatever was drawn; and each draw the device made must be the oracle's draw from the same state wherever the
    oracle's top-two margin is clear of fp32 rounding (``drew``: the rows were drawn by this run, not forced)."""
    with torch.no_grad():
        ref = O.imagine_block(weights_of(wm), weights_of(actor), weights_of(critic), h0, z0, ea, ez, 0.997, 0.95,
                              forced_actions=got["actions"], forced_zidx=got["zidx"])
    for key in ("h", "rewards", "conts", "values", "lam_returns"):
        assert rel_err(got[key], ref[key]) < TOL, key
    assert_close_elementwise(got["h"], ref["h"], rtol=1e-5, atol_scale=2e-6, what="h")
    if not drew:
        return
    lp = O.log_unimix(ref["prior_logits"], 0.01) - ez.permute(1, 0, 2, 3).log()
    top2 = lp.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert float(clear.float().mean()) > 0.99
    assert torch.equal(got["zidx"][clear], ref["zidx_sampled"].to(torch.uint8)[clear])
    sc = (ref["actor_logp"].softmax(-1) / ea.permute(1, 0, 2)).log()
    t2 = sc.topk(2, dim=-1).values
    clear_a = (t2[..., 0] - t2[..., 1]) > 1e-4
    assert float(clear_a.float().mean()) > 0.95
    assert torch.equal(got["actions"][clear_a], ref["actions_sampled"][clear_a])


@pytest.mark.parametrize("precision", ["f32", "f32_mfma"])
def test_large_batch_properties(precision):
    """B = 4096 trajectories (the large-batch case), in the default form of the fp32 products there (split) and in the
    fp32 matrix-instruction form: rows are independent, so any slice of the batch must reproduce, bit for bit, what the
    same rows give when run alone."""
    d, salt, B, H = DIMS["default"], 5, 4096, 15
    wm, actor, critic, eng = _engine(d, salt)
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt)).to(DEV)
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt)).to(DEV)
    full = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, noise={"a": ea, "z": ez}, want_dense_z=False,
                   precision=precision)
    rows = slice(1000, 1300)   # 300 rows: still the large-row kernels, different tile positions
    part = eng.run(WorldModelState(h=h0[rows].to(DEV), z=z0[rows].to(DEV)), H, 0.997, 0.95,
                   noise={"a": ea[:, rows].contiguous(), "z": ez[:, rows].contiguous()}, want_dense_z=False, precision=precision)
    torch.cuda.synchronize()
    for key in ("h", "zidx", "actions", "rewards", "conts", "values", "lam_returns"):
        assert torch.equal(full[key][rows].cpu(), part[key].cpu()), key
    assert torch.isfinite(full["lam_returns"]).all()
    # 16 of the rows against the oracle, which follows the device's own trajectory (its actions and latents forced)
    sub = slice(1000, 1016)
    got = {k: v[:16].cpu() for k, v in part.items()}
    _check_rows_against_oracle(wm, actor, critic, h0[sub], z0[sub], ea[:, sub].cpu(), ez[:, sub].cpu(), got)
    # rows of the full run itself at other tile positions: either side of the first row-tile boundary (63 | 64), either
    # side of the boundary between the two chains (2047 | 2048), inside a middle tile and the last rows of the last tile
    pick = torch.tensor([63, 64, 65, 1500, 1501, 2047, 2048, 2049, 3000, 4032, 4064, 4093, 4094, 4095])
    got_full = {k: v[pick.to(v.device)].cpu() for k, v in full.items()}
    _check_rows_against_oracle(wm, actor, critic, h0[pick], z0[pick], ea[:, pick.to(DEV)].cpu(), ez[:, pick.to(DEV)].cpu(), got_full)
    # the small-row kernels on the same 16 rows: same values to tolerance (different summation order)
    small = eng.run(WorldModelState(h=h0[sub].to(DEV), z=z0[sub].to(DEV)), H, 0.997, 0.95,
                    noise={"a": ea[:, sub].contiguous(), "z": ez[:, sub].contiguous()}, want_dense_z=False,
                    forced_actions=part["actions"][:16].contiguous(), forced_zidx=part["zidx"][:16].contiguous())
    for key in ("h", "rewards", "conts", "values", "lam_returns"):
        assert rel_err(small[key].cpu(), got[key]) < TOL, key


@pytest.mark.parametrize("precision", ["f32", "f32_mfma"])
def test_large_batch_with_a_ragged_last_tile(precision):
    """4066 trajectories: 63 full row tiles and one of 34 rows (and no split into two chains: not a multiple of 128).
    Rows of the last, partial tile and of full tiles against the oracle led along the device's trajectory."""
    d, salt, B, H = DIMS["default"], 6, 4066, 3
    wm, actor, critic, eng = _engine(d, salt)
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt))
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt))
    full = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, noise={"a": ea.to(DEV), "z": ez.to(DEV)},
                   want_dense_z=False, precision=precision)
    torch.cuda.synchronize()
    assert torch.isfinite(full["lam_returns"]).all()
    pick = torch.tensor([0, 63, 64, 2000, 4031, 4032, 4033, 4050, 4064, 4065])
    got = {k: v[pick.to(v.device)].cpu() for k, v in full.items()}
    _check_rows_against_oracle(wm, actor, critic, h0[pick], z0[pick], ea[:, pick], ez[:, pick], got)


def test_large_batch_split_products_follow_the_fp32_form():
    """B = 4096 x H = 15 at the default widths: the split products (both tile forms of the kernel run at this size,
    64 x 128 and 64 x 64, with the weights read from their split images) along the fp32 form's own trajectory give
    its states, heads and returns to the fp32 tolerance; 16 rows also against the oracle."""
    d, salt, B, H = DIMS["default"], 6, 4096, 15
    wm, actor, critic, eng = _engine(d, salt)
    h0, z0, _ = start_state(d, salt, B)
    ea = torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt)).to(DEV)
    ez = torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt)).to(DEV)
    st = WorldModelState(h=h0.to(DEV), z=z0.to(DEV))
    base = eng.run(st, H, 0.997, 0.95, noise={"a": ea, "z": ez}, want_dense_z=False, precision="f32_mfma")
    split = eng.run(st, H, 0.997, 0.95, noise={"a": ea, "z": ez}, want_dense_z=False, precision="f32_split",
                    forced_actions=base["actions"].contiguous(), forced_zidx=base["zidx"].contiguous())
    torch.cuda.synchronize()
    for key in ("h", "rewards", "conts", "values", "v_last", "lam_returns"):
        assert rel_err(split[key].cpu(), base[key].cpu()) < TOL, key
    assert not torch.equal(split["h"], base["h"])
    sub = slice(2000, 2016)
    _check_rows_against_oracle(wm, actor, critic, h0[sub], z0[sub], ea[:, sub].cpu(), ez[:, sub].cpu(),
                               {k: v[sub].cpu() for k, v in split.items()}, drew=False)
    _check_rows_against_oracle(wm, actor, critic, h0[sub], z0[sub], ea[:, sub].cpu(), ez[:, sub].cpu(),
                               {k: v[sub].cpu() for k, v in base.items()})


def test_graph_replay_equals_direct_launches():
    d, salt, B, H = DIMS["default"], 9, 16, 15
    wm, actor, critic, eng = _engine(d, salt)
    h0, z0, _ = start_state(d, salt, B)
    st = WorldModelState(h=h0.to(DEV), z=z0.to(DEV))
    outs = []
    for i in range(3):   # call 0 records the graph, calls 1-2 replay it with new noise
        noise = {"a": torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt + i)).to(DEV),
                 "z": torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt + i)).to(DEV)}
        direct = {k: v.clone() for k, v in eng.run(st, H, 0.997, 0.95, noise=noise).items()}
        replay = eng.run(st, H, 0.997, 0.95, noise=noise, persistent=True)
        torch.cuda.synchronize()
        for k in direct:
            assert torch.equal(direct[k], replay[k]), (i, k)
        outs.append(direct["h"])
    assert not torch.equal(outs[0], outs[1])
