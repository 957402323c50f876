"""Deterministic closed-form generators for weights, noise and synthetic replay data.

Everything here is a pure function of (name, shape, salt): a splitmix64 hash of the element
index, so the same values can be regenerated on any machine and under any numpy version
without storing them.  Used by the golden-vector generator, the tests and the benchmark.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def hash_uniform(name: str, shape: Iterable[int], salt: int = 0) -> np.ndarray:
    """float64 uniform values in (0, 1), one per element, from a hash of (name, salt, index)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    seed = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) << np.uint64(32)
    seed = seed ^ np.uint64(salt & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        bits = _splitmix64(np.arange(n, dtype=np.uint64) ^ _splitmix64(np.full(n, seed, dtype=np.uint64)))
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    return u.reshape(shape)


def hash_exponential(name: str, shape: Iterable[int], salt: int = 0) -> np.ndarray:
    """float32 Exp(1) draws (strictly positive) from the hash stream."""
    e = -np.log1p(-hash_uniform(name, shape, salt))
    return np.maximum(e, 1e-30).astype(np.float32)


def closed_form_param(name: str, shape: Tuple[int, ...], salt: int = 0, gain: float = 1.0) -> np.ndarray:
    """A deterministic value for the parameter called ``name``.

    Matrices / conv kernels: uniform with variance gain^2/fan_in (fan_in = prod(shape[1:])).
    LayerNorm scales (names ending in a norm's ``.weight`` with 1-D or [C,H,W] shape are not
    told apart here, so the caller passes ``kind``-free names): 1-D ``weight`` -> 1 + small,
    ``bias`` -> small.
    """
    u = hash_uniform(name, shape, salt) * 2.0 - 1.0
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("bias", "bias_ih", "bias_hh"):
        return (0.1 * u).astype(np.float32)
    if name.endswith("a_emb.weight"):
        return u.astype(np.float32)
    return (u * (gain * np.sqrt(3.0) / np.sqrt(max(1, int(np.prod(shape[1:])))))).astype(np.float32)


def closed_form_state(shapes: Dict[str, Tuple[int, ...]], norm_keys: Iterable[str] = (), salt: int = 0,
                      gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Values for a whole state dict.  ``norm_keys`` lists the LayerNorm ``weight`` entries, which get
    1 + 0.1*u instead of a fan-in scaled matrix."""
    norm_keys = set(norm_keys)
    out = {}
    for k, shp in shapes.items():
        if k in norm_keys:
            out[k] = (1.0 + 0.1 * (hash_uniform(k, shp, salt) * 2.0 - 1.0)).astype(np.float32)
        else:
            out[k] = closed_form_param(k, tuple(shp), salt, gain)
    return out


def fill_module_closed_form(module, salt: int = 0, gain: float = 1.0, skip_prefixes: Tuple[str, ...] = ()) -> None:
    """Overwrite every floating parameter of a torch module (reference's or this package's) in place with
    the closed-form values for its state-dict name.  LayerNorm weights are detected by module type."""
    import torch
    norm_keys = set()
    for mod_name, mod in module.named_modules():
        if isinstance(mod, torch.nn.LayerNorm):
            norm_keys.add(f"{mod_name}.weight" if mod_name else "weight")
    with torch.no_grad():
        for k, p in module.named_parameters():
            if any(k.startswith(s) for s in skip_prefixes):
                continue
            vals = closed_form_state({k: tuple(p.shape)}, norm_keys & {k}, salt, gain)[k]
            p.copy_(torch.from_numpy(vals).to(p.dtype))


def synthetic_frames(n: int, obs_shape=(3, 64, 64), salt: int = 1234) -> np.ndarray:
    """uint8 frames whose bytes are a hash of (frame index, byte index)."""
    return (hash_uniform("frames", (n, *obs_shape), salt) * 256.0).astype(np.uint8)
