import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# DRM_GUARD_ALLOC=1: every device tensor of the test process comes from the red-zone allocator in tools/guard_alloc
# (regions of their own with canary-filled margins, bodies pre-filled with 0xFF); after each test the margins of all
# freed and live tensors are compared with the canary and a write outside any tensor fails that test.
# GUARD_FLUSH=1 on top of it: every tensor ends 0-15 bytes before an unmapped page, so a read or write of 16 bytes or
# more past the end of any tensor is a memory fault at that kernel (diagnostic runs; with DRM_SYNC_CHECK=2 the last
# "[drm] launched" line on stderr names the library launch).
GUARD = os.environ.get("DRM_GUARD_ALLOC", "0") == "1"
GUARD_FLUSH = GUARD and os.environ.get("GUARD_FLUSH", "0") == "1"
_guard = None
if GUARD:
    import ctypes

    import torch

    _so = os.path.join(ROOT, "tools", "guard_alloc", "libguard_alloc.so")
    if not os.path.exists(_so):
        raise RuntimeError(f"{_so} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(_so, "guard_malloc", "guard_free"))
    _guard = ctypes.CDLL(_so)
    _guard.guard_report.restype = ctypes.c_int
    _guard.guard_report.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]

# tests that record torch.cuda graphs: the pluggable-allocator interface has no graph memory pools
_NEEDS_GRAPH_POOLS = ("test_gpu_trainer.py", "test_flat_adam_in_a_recorded_graph")


def guard_report(reset=True):
    """None without the guard allocator, else a dict describing writes found outside tensors since the last reset."""
    if _guard is None:
        return None
    out = (ctypes.c_ulonglong * 8)()
    if _guard.guard_report(0, out, 1 if reset else 0) != 0:
        return {"error": "device not reachable"}
    return {"corrupt_words": out[0], "alloc_ordinal": out[1], "tensor_bytes": out[2], "side": "after" if out[3] else "before",
            "distance_bytes": out[4], "value": hex(out[5]), "allocations": out[6], "reused_regions": out[7]}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# DRM_SYNC_CHECK: the library synchronises after every launch, which a recording does not allow; the library and the
# trainer then launch directly, and the tests that record library calls (or assert that a recording was made) have nothing to check
_RECORDS_LIBRARY_CALLS = ("test_flat_adam_in_a_recorded_graph", "test_graph_replayed_updates_match_eager",
                          "test_acting_policy_graph_replay_matches_eager_steps", "test_train_loop_end_to_end_on_the_gpu")
SYNC_CHECK = os.environ.get("DRM_SYNC_CHECK", "0") not in ("", "0")


def pytest_collection_modifyitems(config, items):
    if GUARD:
        skip = pytest.mark.skip(reason="records torch.cuda graphs, which the guard allocator cannot serve")
        for item in items:
            if any(k in item.nodeid for k in _NEEDS_GRAPH_POOLS):
                item.add_marker(skip)
    if SYNC_CHECK:
        skip = pytest.mark.skip(reason="records a library call, which DRM_SYNC_CHECK's synchronising launch check cannot be part of")
        for item in items:
            if any(k in item.nodeid for k in _RECORDS_LIBRARY_CALLS):
                item.add_marker(skip)


@pytest.fixture(autouse=True)
def _guard_check(request):
    yield
    if GUARD and request.node.get_closest_marker("gpu") is not None:
        rep = guard_report()
        if rep.get("error") or rep["corrupt_words"]:
            pytest.fail(f"guard allocator: write outside a tensor: {rep}")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
