"""Ring replay buffer with the reference's surface (lib/replay_buffer.py), stored in device memory.

Start indices are drawn on the host with the global numpy generator, in the same call with the
same arguments as the reference (lib/replay_buffer.py:55-62), so for a given ``np.random.seed`` the
sampled sequences are the same.  On a HIP device the (B, T) gather runs as one kernel over the
HBM-resident ring; on a host device it is the plain index gather.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


class ReplayBuffer:
    def __init__(self, capacity: int, obs_shape: Tuple[int, ...], seq_len: int, device: torch.device) -> None:
        self.capacity, self.obs_shape, self.seq_len = int(capacity), tuple(obs_shape), int(seq_len)
        self.device = torch.device(device)
        self.observations = torch.zeros((self.capacity, *self.obs_shape), dtype=torch.uint8, device=self.device)
        self.actions = torch.zeros(self.capacity, dtype=torch.long, device=self.device)
        self.rewards = torch.zeros(self.capacity, dtype=torch.float32, device=self.device)
        self.continues = torch.zeros(self.capacity, dtype=torch.bool, device=self.device)
        self.idx, self.full = 0, False

    def __len__(self) -> int:
        return self.capacity if self.full else self.idx

    def store(self, obs: np.ndarray, action: int, reward: float, cont: bool) -> None:
        i = self.idx
        self.observations[i].copy_(torch.as_tensor(obs, dtype=torch.uint8), non_blocking=True)
        self.actions[i] = int(action)
        self.rewards[i] = float(reward)
        self.continues[i] = bool(cont)
        self.idx = (i + 1) % self.capacity
        self.full = self.full or self.idx == 0

    def store_many(self, obs, actions, rewards, conts) -> None:
        """Append n transitions at once (same final state as n ``store`` calls)."""
        n = len(actions)
        if n > self.capacity:  # only the last `capacity` transitions survive; keep ring positions distinct
            skip = n - self.capacity
            self.idx, self.full = (self.idx + skip) % self.capacity, True
            obs, actions, rewards, conts = obs[skip:], actions[skip:], rewards[skip:], conts[skip:]
            n = self.capacity
        pos = (self.idx + np.arange(n)) % self.capacity
        pos_t = torch.as_tensor(pos, device=self.device)
        self.observations[pos_t] = torch.as_tensor(obs, dtype=torch.uint8).to(self.device)
        self.actions[pos_t] = torch.as_tensor(actions, dtype=torch.long).to(self.device)
        self.rewards[pos_t] = torch.as_tensor(rewards, dtype=torch.float32).to(self.device)
        self.continues[pos_t] = torch.as_tensor(conts, dtype=torch.bool).to(self.device)
        self.full = self.full or self.idx + n >= self.capacity
        self.idx = (self.idx + n) % self.capacity

    # ---- persistence (the reference keeps its buffer in memory only and saves nothing: lib/trainer.py:326-335) ----
    def state_dict(self, stamp=None) -> Dict[str, torch.Tensor]:
        """The filled part of the ring and its write position, as host tensors only (a file written with
        ``torch.save`` of this dict loads with ``torch.load(..., weights_only=True)``).  ``stamp``: integers that
        identify the checkpoint the ring belongs to (kept under "stamp"; compared by the loader of the pair)."""
        n = len(self)
        sd = {"observations": self.observations[:n].cpu(), "actions": self.actions[:n].cpu(),
              "rewards": self.rewards[:n].cpu(), "continues": self.continues[:n].cpu(),
              "meta": torch.tensor([self.capacity, self.idx, int(self.full), self.seq_len], dtype=torch.int64)}
        if stamp is not None:
            sd["stamp"] = torch.tensor([int(v) for v in stamp], dtype=torch.int64)
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        cap, idx, full, seq_len = (int(v) for v in sd["meta"])
        if cap != self.capacity or tuple(sd["observations"].shape[1:]) != self.obs_shape:
            raise ValueError(f"replay file holds capacity {cap}, frames {tuple(sd['observations'].shape[1:])}; this ring has "
                             f"capacity {self.capacity}, frames {self.obs_shape}")
        if seq_len != self.seq_len:
            raise ValueError(f"replay file was written for sequences of {seq_len} steps; this ring samples {self.seq_len}")
        n = sd["actions"].numel()
        if n != (cap if full else idx):
            raise ValueError("replay file is inconsistent: stored rows do not match its write position")
        self.observations[:n].copy_(sd["observations"])
        self.actions[:n].copy_(sd["actions"])
        self.rewards[:n].copy_(sd["rewards"])
        self.continues[:n].copy_(sd["continues"])
        self.idx, self.full = idx, bool(full)

    def save(self, path: str, stamp=None) -> None:
        torch.save(self.state_dict(stamp), path)

    def load(self, path: str, expect_stamp=None) -> bool:
        """Load a ring written by ``save``.  With ``expect_stamp`` the file is taken only when it carries that stamp
        (a ring left behind by another checkpoint is not paired with newer counters); returns whether it was taken."""
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if expect_stamp is not None:
            have = [int(v) for v in sd["stamp"]] if "stamp" in sd else None
            if have != [int(v) for v in expect_stamp]:
                return False
        self.load_state_dict(sd)
        return True

    def draw_starts(self, batch_size: int) -> np.ndarray:
        assert len(self) >= self.seq_len, "Not enough data to sample."
        if self.full:
            return np.random.randint(0, self.capacity, size=batch_size, dtype=np.int64)
        last = self.idx - self.seq_len
        if last < 0:
            raise RuntimeError("Not enough data to sample the requested sequence length.")
        return np.random.randint(0, last + 1, size=batch_size, dtype=np.int64)

    def gather(self, starts: np.ndarray, want_f32: bool = False) -> Dict[str, torch.Tensor]:
        T = self.seq_len
        if self.device.type == "cuda":
            from . import _native
            st = self._starts_to_device(np.ascontiguousarray(starts, dtype=np.int64))
            o, a, r, c, f = _native.seq_gather(self.observations, self.actions, self.rewards, self.continues, st, T,
                                               want_f32)
        else:
            rows = torch.from_numpy((starts[:, None] + np.arange(T, dtype=np.int64)[None, :]) % self.capacity)
            o, a, r, c = self.observations[rows], self.actions[rows], self.rewards[rows], self.continues[rows]
            f = o.float() / 255.0 if want_f32 else None
        out = {"observations": o, "actions": a, "rewards": r, "continues": c}
        if want_f32:
            out["observations_f32"] = f
        return out

    def _starts_to_device(self, starts: np.ndarray) -> torch.Tensor:
        """Start indices to the device through a small ring of pinned staging rows: a copy from pageable memory would
        make the host wait until the stream has run everything queued before it (once per update: the host could never
        queue the next update's launches while the device works on this one)."""
        n = int(starts.shape[0])
        ring = getattr(self, "_pin_ring", None)
        if ring is None or ring[0].shape[1] < n:
            slots = 16
            ring = (torch.empty(slots, max(n, 64), dtype=torch.int64, pin_memory=True), [None] * slots, [0])
            self._pin_ring = ring
        rows, events, cursor = ring
        k = cursor[0] % rows.shape[0]
        cursor[0] += 1
        if events[k] is not None:
            events[k].synchronize()          # the copy that last read this row (16 samples ago) has run
        rows[k, :n].copy_(torch.from_numpy(starts))
        st = rows[k, :n].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        events[k] = ev
        return st

    def sample(self, batch_size: int, want_f32: bool = False) -> Dict[str, torch.Tensor]:
        return self.gather(self.draw_starts(batch_size), want_f32)
