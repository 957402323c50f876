"""The allocation-free imagination entry point (static device workspace inside the library) against the
entry point that takes a caller workspace: same kernels, so the same bits (GPU)."""
import ctypes as C

import pytest
import torch

from dreamer_amd import _native as N
from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.testing import hash_exponential
from dreamer_amd.world_model import WorldModelState
from helpers import DIMS, build_models, start_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(eng, d, salt, B, H, **kw):
    h0, z0, _ = start_state(d, salt, B)
    noise = {"a": torch.from_numpy(hash_exponential("ea", (H, B, d["A"]), salt)).to(DEV),
             "z": torch.from_numpy(hash_exponential("ez", (H, B, d["L"], d["K"]), salt)).to(DEV)}
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), H, 0.997, 0.95, noise=noise, want_prior_logits=True, **kw)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in out.items()}


def test_noalloc_entry_point_equals_the_workspace_entry_point_twice_in_a_row():
    d, salt = DIMS["default"], 13
    wm, actor, critic = build_models(d, salt)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    # two calls in a row on the shared static workspace, the second with a smaller input (fewer trajectories and a
    # shorter horizon: every region of the workspace shrinks, stale contents of the first call lie behind them)
    for B, H, s in ((16, 15, salt), (5, 7, salt + 1)):
        ref = _run(eng, d, s, B, H)
        got = _run(eng, d, s, B, H, noalloc=True)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (B, H, k)
        assert torch.isfinite(got["lam_returns"]).all()


def test_noalloc_entry_point_rejects_what_does_not_fit():
    d, salt = DIMS["default"], 13
    wm, actor, critic = build_models(d, salt)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    need = N.lib().drm_imagine_workspace_bytes(C.byref(eng.dims), 300, 15)
    assert need > (16 << 20)
    with pytest.raises(N.NativeError, match="static workspace"):
        _run(eng, d, salt, 300, 15, noalloc=True)
    # a size that fits still works afterwards
    ref = _run(eng, d, salt, 3, 2)
    got = _run(eng, d, salt, 3, 2, noalloc=True)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k


def test_noalloc_entry_point_runs_one_chain_at_many_trajectories():
    """Narrow widths let thousands of trajectories fit the static workspace; the workspace entry point would split such
    a call into side-by-side chains on streams of its own (created on first use), the allocation-free one must not:
    it runs one chain on the caller's stream, and since rows are independent the results are the same bits."""
    d, salt, B, H = dict(DIMS["small"], D=32, M=32, L=1, A=4), 29, 2048, 1
    wm, actor, critic = build_models(d, salt)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    assert N.lib().drm_imagine_workspace_bytes(C.byref(eng.dims), B, H) <= (16 << 20)
    ref = _run(eng, d, salt, B, H)
    got = _run(eng, d, salt, B, H, noalloc=True)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
