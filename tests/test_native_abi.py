"""The C-ABI library loads and exports every symbol the header declares (no GPU needed)."""
import os
import re

from dreamer_amd import _native

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    text = open(os.path.join(ROOT, "include", "dreamer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 10
    handle = _native.lib()
    for n in names:
        assert hasattr(handle, n), f"{n} declared in the header but not exported"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_native.SIGNATURES) == names


def test_version_and_error_text():
    handle = _native.lib()
    assert handle.drm_version() >= 100
    assert isinstance(handle.drm_last_error(), bytes)


def test_argument_validation_without_gpu():
    # shape checks run before anything touches a device
    handle = _native.lib()
    assert handle.drm_lambda_return(None, None, None, None, None, 4, 15, 0.997, 0.95, None) == -22
    assert b"null" in handle.drm_last_error()
    assert handle.drm_sample_categorical_st(1, 1, 4, 16, 0.01, None, None, None, None) == -95
    d = _native.drm_dims(D=512, L=32, K=16, M=512, A=6, E=64, NB=255, unimix_eps=0.01)
    import ctypes
    assert handle.drm_pack_imagine_bytes(ctypes.byref(d)) == 0
    d.K = 32
    assert handle.drm_pack_imagine_bytes(ctypes.byref(d)) == 4 * ((4 * 512 + 512) * 1024 + 6 * 512 + 6 * 3 * 512)
