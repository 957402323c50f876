"""Pointwise helpers on the hot path plus the environment factory.

symlog / symexp / log_unimix follow lib/utils.py:12-31 of the reference.  ``make_env`` returns
a gymnasium environment when gymnasium (and the ALE) are importable, and otherwise a
synthetic pixel environment with the same observation / action interface, so that the trainer
can be run and benchmarked on a machine without them.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def symlog(x: torch.Tensor) -> torch.Tensor:
    return torch.sign(x) * torch.log1p(torch.abs(x))


def symexp(x: torch.Tensor) -> torch.Tensor:
    return torch.sign(x) * torch.expm1(torch.abs(x))


def log_unimix(logits: torch.Tensor, eps: float, dim: int = -1) -> torch.Tensor:
    """log((1 - eps) * softmax(logits) + eps / K), computed in log space."""
    k = logits.size(dim)
    shifted = F.log_softmax(logits, dim=dim) + math.log1p(-float(eps))
    # a fill kernel rather than new_tensor(): no host-to-device copy, so the op can be recorded into a graph
    floor = torch.full((), math.log(float(eps)) - math.log(float(k)), dtype=logits.dtype, device=logits.device)
    return torch.logaddexp(shifted, floor)


class _Discrete:
    def __init__(self, n: int, rng: np.random.Generator):
        self.n = int(n)
        self._rng = rng

    def sample(self) -> int:
        return int(self._rng.integers(0, self.n))


class _Box:
    def __init__(self, shape):
        self.shape = tuple(shape)
        self.dtype = np.uint8
        self.low, self.high = 0, 255


class SyntheticPixelEnv:
    """A deterministic stand-in for a 64x64 RGB Atari-like environment (CHW uint8 observations).

    A bright square moves over a textured background; the action shifts it; the reward is +1 / -1
    when it hits the right / left wall; episodes end after ``episode_len`` steps.
    """

    def __init__(self, frame_size: int = 64, n_actions: int = 6, episode_len: int = 200, seed: int = 0):
        self._rng = np.random.default_rng(seed)
        self.observation_space = _Box((3, frame_size, frame_size))
        self.action_space = _Discrete(n_actions, self._rng)
        self._size, self._len = frame_size, episode_len
        yy, xx = np.mgrid[0:frame_size, 0:frame_size]
        self._bg = np.stack([(xx * 3 + yy) % 256, (xx + yy * 5) % 256, (xx * yy) % 256]).astype(np.uint8)
        self._t, self._pos = 0, np.array([frame_size // 2, frame_size // 2])

    def _frame(self) -> np.ndarray:
        f = self._bg.copy()
        y, x = self._pos
        f[:, max(0, y - 3):y + 3, max(0, x - 3):x + 3] = 255
        return f

    def reset(self, seed=None, options=None):
        if seed is not None:
            self._rng = np.random.default_rng(seed)
            self.action_space._rng = self._rng
        self._t = 0
        m = min(8, self._size // 4)
        self._pos = self._rng.integers(m, self._size - m, size=2)
        return self._frame(), {}

    def step(self, action: int):
        moves = [(0, 0), (0, 2), (0, -2), (2, 0), (-2, 0), (0, 0)]
        dy, dx = moves[int(action) % len(moves)]
        self._pos = np.clip(self._pos + np.array([dy, dx]), 3, self._size - 4)
        self._t += 1
        reward = 1.0 if self._pos[1] >= self._size - 4 else (-1.0 if self._pos[1] <= 3 else 0.0)
        terminated = reward != 0.0
        truncated = self._t >= self._len
        return self._frame(), reward, terminated, truncated, {}

    def render(self):
        return np.transpose(self._frame(), (1, 2, 0))

    def close(self):
        pass


def make_env(env_id: str, frame_size: int = 64):
    """Environment factory with the reference's signature (lib/utils.py:50-57)."""
    if env_id.startswith("synthetic"):
        return SyntheticPixelEnv(frame_size)
    try:
        import gymnasium as gym
    except ImportError:
        print(f"[dreamer_amd] gymnasium is not installed; using SyntheticPixelEnv instead of {env_id}")
        return SyntheticPixelEnv(frame_size)

    class _ToCHW(gym.ObservationWrapper):
        def __init__(self, env):
            super().__init__(env)
            h, w, c = self.observation_space.shape
            self.observation_space = gym.spaces.Box(low=0, high=255, shape=(c, h, w), dtype=np.uint8)

        def observation(self, observation):
            return np.transpose(observation, (2, 0, 1))

    env = gym.make(env_id, render_mode="rgb_array")
    env = gym.wrappers.ResizeObservation(env, (frame_size, frame_size))
    return _ToCHW(env)


def log_episode_video(cfg, summary_writer, env, world_model, actor, step):
    """Evaluation-episode video (lib/utils.py:60-117): visual diagnostics, not part of the hot path."""
    return None


def log_wm_reconstruction_video(cfg, summary_writer, env, world_model, actor, step):
    """Posterior reconstruction video (lib/utils.py:129-190): visual diagnostics, not part of the hot path."""
    return None


def log_wm_imagination_video(cfg, summary_writer, env, world_model, actor, step):
    """Prior imagination video (lib/utils.py:193-266): visual diagnostics, not part of the hot path."""
    return None
