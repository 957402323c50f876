"""Self-test of the red-zone allocator the GPU suite can run under (DRM_GUARD_ALLOC=1): a deliberate 24-byte write
past the end of a tensor must be reported, with the tensor's size and the side, and must not fault."""
import pytest
import torch

import conftest
from dreamer_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not conftest.GUARD or conftest.GUARD_FLUSH,
                    reason="runs under DRM_GUARD_ALLOC=1 only (and not in the flush form, where the write would fault)")
def test_guard_allocator_reports_a_write_past_the_end():
    conftest.guard_report()
    dst = torch.zeros(10, dtype=torch.float32, device="cuda:0")       # 40 bytes
    src = torch.ones(16, dtype=torch.float32, device="cuda:0")        # 64 bytes
    other = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    assert torch.isnan(torch.empty(8, device="cuda:0")).all()          # fresh memory is poisoned, not zero
    # raw library call that copies 64 bytes into the 40-byte tensor: 24 bytes land in its margin
    N.check(N.lib().drm_copy_pair(src.data_ptr(), dst.data_ptr(), 64, src.data_ptr(), other.data_ptr(), 64,
                                  N.stream_ptr(dst.device)), "drm_copy_pair")
    rep = conftest.guard_report()
    assert rep["corrupt_words"] == 6 and rep["tensor_bytes"] == 40 and rep["side"] == "after" and rep["distance_bytes"] == 0, rep
    assert conftest.guard_report()["corrupt_words"] == 0               # reported once, then healed
