"""Imagination block on the GPU: the no-grad part of an update between the world-model step and
the actor / critic steps (lib/trainer.py:176-224) as one native call.

    actor -> sample action -> prior RSSM step        (H times, B trajectories)
    reward / continue heads, critic value            (once, over all B*H imagined states)
    lambda-return                                    (reverse scan over H)

The decoder is not evaluated: the reference computes it inside ``step`` and discards the result
(lib/world_model.py:447, lib/trainer.py:192); it consumes no random numbers, so skipping it does not
change any value.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _native as N
from .world_model import WorldModelState


def _mlp_struct(mlp, head, in_dim: int) -> N.drm_mlp:
    linears = [m for m in mlp if isinstance(m, torch.nn.Linear)]
    norms = [m for m in mlp if isinstance(m, torch.nn.LayerNorm)]
    s = N.drm_mlp()
    s.n_layers, s.in_dim, s.hidden, s.out_dim = len(linears), in_dim, linears[0].out_features, head.out_features
    for i, (lin, ln) in enumerate(zip(linears, norms)):
        s.w[i], s.b[i], s.ln_g[i], s.ln_b[i] = (lin.weight.data_ptr(), lin.bias.data_ptr(), ln.weight.data_ptr(),
                                                ln.bias.data_ptr())
    s.head_w, s.head_b = head.weight.data_ptr(), head.bias.data_ptr()
    return s


def resolve_precision(precision: str, trajectories: int, recurrent_width: int, hidden_width: int) -> int:
    """The library's ``precision`` code (0 fp32 matrix instructions, 1 bf16 matrix inputs, 2 exact three-way bf16 split) for a
    precision name.  ``"f32"`` asks for fp32 results and leaves the form to the engine: the split form from 256
    trajectories on (the faster fp32-accurate form there at every width: 6.3 against 8.3 ms per block of 4096 default-width
    trajectories, 28 against 34 ms per block of 1024 at a recurrent width of 4096; ``DRM_F32_MFMA=1`` disables the choice),
    fp32 matrix instructions otherwise."""
    if precision == "f32":
        split_wins = trajectories >= 256 and os.environ.get("DRM_F32_MFMA", "0") != "1"
        precision = "f32_split" if split_wins else "f32_mfma"
    try:
        return {"f32_mfma": 0, "bf16": 1, "f32_split": 2}[precision]
    except KeyError:
        raise ValueError(f"unknown precision {precision!r}: expected 'f32', 'f32_mfma', 'f32_split' or 'bf16'") from None


class ImagineEngine:
    """Owns the packed gather tables and the workspace for one (world model, actor, critic) triple."""

    def __init__(self, world_model, actor, critic):
        self.wm, self.actor, self.critic = world_model, actor, critic
        p = next(world_model.parameters())
        if not p.is_cuda:
            raise N.NativeError("ImagineEngine needs the models on a HIP device")
        self.device = p.device
        wm = world_model
        self.dims = N.drm_dims(D=wm.hidden_size, L=wm.num_latents, K=wm.classes_per_latent,
                               M=wm.dyn.mlp[0].out_features, A=wm.action_size, E=wm.seq.a_emb.embedding_dim,
                               NB=critic.num_buckets, unimix_eps=wm.unimix_eps)
        # the rollout reads vectors and matrices; the convolution kernels of the towers (4-D, possibly kept in the
        # channels-last format) are not part of it
        for m in list(wm.parameters()) + list(actor.parameters()) + list(critic.parameters()):
            if m.dim() == 4:
                continue
            if m.dtype != torch.float32 or not m.is_contiguous():
                raise N.NativeError("parameters must be contiguous fp32")
        nbytes = N.lib().drm_pack_imagine_bytes(C.byref(self.dims))
        if nbytes == 0:
            raise N.NativeError(f"unsupported dimensions: {N.lib().drm_last_error().decode()}")
        self.pack = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        self._ws = None
        self._ws_key = None
        self._slots = {}
        self.repack()

    def _weights(self) -> N.drm_imagine_weights:
        wm, d = self.wm, self.dims
        S = d.D + d.L * d.K
        w = N.drm_imagine_weights()
        w.dims = d
        w.a_emb, w.lin_w, w.lin_b = wm.seq.a_emb.weight.data_ptr(), wm.seq.linear.weight.data_ptr(), wm.seq.linear.bias.data_ptr()
        g = wm.seq.gru
        w.gru_w_ih, w.gru_w_hh, w.gru_b_ih, w.gru_b_hh = (g.weight_ih.data_ptr(), g.weight_hh.data_ptr(),
                                                         g.bias_ih.data_ptr(), g.bias_hh.data_ptr())
        w.dyn = _mlp_struct(wm.dyn.mlp, wm.dyn.head, d.D)
        w.rew = _mlp_struct(wm.rew.mlp, wm.rew.head, S)
        w.cont = _mlp_struct(wm.cont.mlp, wm.cont.head, S)
        w.actor = _mlp_struct(self.actor.mlp, self.actor.head, S)
        w.critic = _mlp_struct(self.critic.mlp, self.critic.head, S)
        w.buckets = self.critic.buckets.data_ptr()
        w.pack, w.pack_bytes = self.pack.data_ptr(), self.pack.numel() * 4
        return w

    def repack(self) -> None:
        """Rebuild the one-hot gather tables from the current weights (needed after optimiser steps)."""
        w = self._weights()
        N.check(N.lib().drm_pack_imagine(C.byref(w), self.pack.data_ptr(), self.pack.numel() * 4,
                                         N.stream_ptr(self.device)), "drm_pack_imagine")

    def _workspace(self, B: int, H: int) -> torch.Tensor:
        if self._ws_key != (B, H):
            nbytes = N.lib().drm_imagine_workspace_bytes(C.byref(self.dims), B, H)
            self._ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=self.device)
            self._ws_key = (B, H)
        return self._ws

    def _alloc_outputs(self, B, H, want_dense_z, want_logp, want_prior_logits, want_sampled):
        d, dev = self.dims, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        out = {
            "h": torch.empty(B, H, d.D, **f32), "zidx": torch.empty(B, H, d.L, dtype=torch.uint8, device=dev),
            "actions": torch.empty(B, H, dtype=torch.int64, device=dev), "rewards": torch.empty(B, H, **f32),
            "conts": torch.empty(B, H, **f32), "values": torch.empty(B, H, **f32), "v_last": torch.empty(B, **f32),
            "lam_returns": torch.empty(B, H, **f32),
        }
        if want_dense_z:
            out["z"] = torch.empty(B, H, d.L, d.K, **f32)
        if want_logp:
            out["actor_logp"] = torch.empty(B, H, d.A, **f32)
        if want_prior_logits:
            out["prior_logits"] = torch.empty(B, H, d.L, d.K, **f32)
        if want_sampled:
            out["zidx_sampled"] = torch.empty(B, H, d.L, dtype=torch.uint8, device=dev)
            out["actions_sampled"] = torch.empty(B, H, dtype=torch.int64, device=dev)
        return out

    def _device_noise_ok(self, B: int, precision_code: int) -> bool:
        """Whether the in-kernel generator serves this shape: the step-wise launch sequence (at most 64 trajectories, or
        256 and more with fp32 / split products) at widths up to 512."""
        d = self.dims
        if os.environ.get("DRM_HOST_NOISE", "0") == "1" or d.D > 512 or d.M > 512:
            return False
        if B <= 64:
            return True
        return (B >= 256 and precision_code in (0, 2) and d.L <= 64 and os.environ.get("DRM_NO_FUSED_STEP") is None)

    def _reserve_noise(self):
        """(seed, counter) for one call: the device generator's seed and its current offset, which is advanced as a torch
        kernel that draws would advance it -- the generator's state covers the in-kernel draws too."""
        gen = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
        offset = int(gen.get_offset())
        gen.set_offset(offset + 4)
        return int(gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF, offset

    @torch.no_grad()
    def _noise_buffer(self, B: int, H: int):
        """Action noise (H,B,A) and latent noise (H,B,L,K) as views of one buffer, so that a single
        generator launch draws both (the same layout, hence the same stream, on every path)."""
        d = self.dims
        n_a, n_z = H * B * d.A, H * B * d.L * d.K
        pad_a = (n_a + 63) // 64 * 64
        flat = torch.empty(pad_a + n_z, device=self.device)
        return flat, flat[:n_a].view(H, B, d.A), flat[pad_a:].view(H, B, d.L, d.K)

    def run(self, start: WorldModelState, horizon: int, gamma: float, lam: float,
            noise: Optional[Dict[str, torch.Tensor]] = None, forced_actions: Optional[torch.Tensor] = None,
            forced_zidx: Optional[torch.Tensor] = None, want_dense_z: bool = True, want_logp: bool = True,
            want_prior_logits: bool = False, want_sampled: bool = False, start_zidx: Optional[torch.Tensor] = None,
            repack: bool = True, persistent: bool = False, precision: str = "f32",
            noalloc: bool = False, device_noise: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """Run the block.  With ``persistent=True`` all inputs are staged into, and all outputs
        returned as, buffers that live as long as the engine (overwritten by the next call with the
        same shape); every device address is then the same from call to call and the native side
        replays the block as one recorded hipGraph.  ``precision="f32"`` (default): fp32 results to the rollout's
        tolerance; from 256 trajectories on the matrix products are formed at fp32 accuracy on the bf16
        matrix pipe (exact three-way split of both operands, the faster accurate form there), otherwise with fp32 matrix
        instructions.  ``"f32_mfma"`` / ``"f32_split"`` ask for one of the two forms explicitly (``DRM_F32_MFMA=1`` makes
        ``"f32"`` mean ``"f32_mfma"`` everywhere).  ``precision="bf16"``: the matrix products of the large-row path take
        bf16 inputs (fp32 accumulation, everything else fp32).
        ``noalloc=True``: the library's entry point that takes no workspace (it uses a static device array of its own,
        for at most a few dozen trajectories) and never records a graph.
        Noise: ``noise`` given -> those Exp(1) tensors (parity tests).  Otherwise either one ``exponential_`` launch fills
        noise tensors, or the draws take their numbers from the counter-based generator inside the kernels, keyed by
        torch's device seed and addressed by a block of the device generator's offset reserved per call (so
        ``torch.manual_seed`` and a restored generator state reproduce a run either way).  ``device_noise=None`` picks
        the faster of the two: in-kernel from 256 trajectories on, the launch below (``True`` / ``DRM_DEVICE_NOISE=1``:
        in-kernel wherever the step-wise launch sequence runs; ``False`` / ``DRM_HOST_NOISE=1``: always the launch)."""
        d, dev = self.dims, self.device
        B, H = start.h.shape[0], int(horizon)
        if repack:
            self.repack()
        flags = (want_dense_z, want_logp, want_prior_logits, want_sampled)
        slot = self._slots.get((B, H, flags)) if persistent else None
        if slot is None:
            slot = {"out": self._alloc_outputs(B, H, *flags)}
            if persistent:
                flat, ea, ez = self._noise_buffer(B, H)
                slot.update(h0=torch.empty(B, d.D, device=dev), zidx0=torch.empty(B, d.L, dtype=torch.uint8, device=dev),
                            noise=flat, ea=ea, ez=ez)
                self._slots[(B, H, flags)] = slot
        out = slot["out"]
        h0 = start.h.detach()
        zidx0 = start_zidx if start_zidx is not None else N.onehot_to_index(start.z.detach().contiguous().float())
        prec = resolve_precision(precision, B, d.D, d.M)
        # measured on one MI355X: drawing inside the kernels wins from 256 trajectories on (B = 4096: 6.88 against 7.00 ms per
        # block, no 250 MB of noise to write and read); at 16 trajectories the generator's arithmetic sits on the critical
        # path of the draw kernels and costs what the noise launch cost (0.573 against 0.566 ms), so few rows keep the launch
        prefer = device_noise if device_noise is not None else (B >= 256 or os.environ.get("DRM_DEVICE_NOISE", "0") == "1")
        in_kernel = noise is None and prefer and self._device_noise_ok(B, prec)
        rng_seed = rng_counter = 0
        rng_state = None
        if in_kernel:
            rng_seed, rng_counter = self._reserve_noise()
        ea = ez = None
        if persistent:
            if in_kernel and "rng" not in slot:
                slot["rng"] = torch.zeros(1, dtype=torch.int64, device=dev)
            rng_state = slot["rng"] if in_kernel else None
            if h0.dtype == torch.float32 and h0.is_contiguous() and zidx0.is_contiguous():
                N.copy_pair(slot["h0"], h0, slot["zidx0"], zidx0, rng_state, rng_counter)
            else:
                slot["h0"].copy_(h0)
                slot["zidx0"].copy_(zidx0)
                if in_kernel:
                    rng_state.fill_(rng_counter - (1 << 64) if rng_counter >= (1 << 63) else rng_counter)
            h0, zidx0 = slot["h0"], slot["zidx0"]
            if in_kernel:
                rng_counter = 0          # read from the device word: the recorded launches keep their arguments
            elif noise is None:
                slot["noise"].exponential_()
                ea, ez = slot["ea"], slot["ez"]
            else:
                ea, ez = slot["ea"].copy_(noise["a"]), slot["ez"].copy_(noise["z"])
        else:
            h0 = h0.contiguous().float()
            if in_kernel:
                pass
            elif noise is None:
                flat, ea, ez = self._noise_buffer(B, H)
                flat.exponential_()
            else:
                ea, ez = noise["a"].contiguous(), noise["z"].contiguous()
        assert in_kernel or (ea.shape == (H, B, d.A) and ez.shape == (H, B, d.L, d.K))
        ws = None if noalloc else self._workspace(B, H)
        io = N.drm_imagine_io()
        io.B, io.H = B, H
        io.h0, io.zidx0, io.expo_a, io.expo_z = h0.data_ptr(), zidx0.data_ptr(), N.ptr(ea), N.ptr(ez)
        io.rng_state, io.rng_seed, io.rng_counter = N.ptr(rng_state), rng_seed, rng_counter
        io.forced_actions = N.ptr(forced_actions.contiguous() if forced_actions is not None else None)
        io.forced_zidx = N.ptr(forced_zidx.contiguous() if forced_zidx is not None else None)
        io.h_out, io.zidx_out, io.actions_out = out["h"].data_ptr(), out["zidx"].data_ptr(), out["actions"].data_ptr()
        io.z_out = N.ptr(out.get("z"))
        io.zidx_sampled_out, io.actions_sampled_out = N.ptr(out.get("zidx_sampled")), N.ptr(out.get("actions_sampled"))
        io.actor_logp_out, io.prior_logits_out = N.ptr(out.get("actor_logp")), N.ptr(out.get("prior_logits"))
        io.rewards_out, io.conts_out, io.values_out = (out["rewards"].data_ptr(), out["conts"].data_ptr(),
                                                       out["values"].data_ptr())
        io.v_last_out, io.lam_returns_out = out["v_last"].data_ptr(), out["lam_returns"].data_ptr()
        io.gamma, io.lam = float(gamma), float(lam)
        io.precision = prec
        w = self._weights()
        if noalloc:
            io.workspace, io.workspace_bytes, io.use_graph = None, 0, 0
            N.check(N.lib().drm_imagine_block_noalloc_f32(C.byref(w), C.byref(io), N.stream_ptr(dev)),
                    "drm_imagine_block_noalloc_f32")
            return out
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        io.use_graph = 1 if (persistent and forced_actions is None and forced_zidx is None) else 0
        N.check(N.lib().drm_imagine_block_f32(C.byref(w), C.byref(io), N.stream_ptr(dev)), "drm_imagine_block_f32")
        return out


def imagine_eager(world_model, actor, critic, start: WorldModelState, horizon: int, gamma: float, lam: float):
    """The same block with eager tensor ops, for models that live on a host device."""
    from .utils import symexp
    s = start
    hs, zs, acts, rews, conts, logps = [], [], [], [], [], []
    with torch.no_grad():
        for _ in range(horizon):
            dist = actor(s)
            a = dist.sample()
            s, info = world_model.step(s, a_prev_idx=a)
            hs.append(s.h); zs.append(s.z); acts.append(a); logps.append(dist.logits)
            rews.append(symexp(info["r_hat"].squeeze(-1)))
            conts.append(torch.sigmoid(info["c_hat"].squeeze(-1)))
        Hs = WorldModelState(h=torch.stack(hs, 1), z=torch.stack(zs, 1))
        R, Cn = torch.stack(rews, 1), torch.stack(conts, 1)
        V = critic.value(Hs)
        v_last = critic.value(WorldModelState(h=hs[-1], z=zs[-1]))
        ret = torch.zeros_like(V)
        nxt = v_last
        for t in reversed(range(horizon)):
            nxt = R[:, t] + gamma * (1.0 - Cn[:, t]) * ((1.0 - lam) * V[:, t] + lam * nxt)
            ret[:, t] = nxt
    return {"h": Hs.h, "z": Hs.z, "actions": torch.stack(acts, 1), "rewards": R, "conts": Cn, "values": V,
            "v_last": v_last, "lam_returns": ret, "actor_logp": torch.stack(logps, 1)}


def imagine(world_model, actor, critic, start: WorldModelState, horizon: int, gamma: float, lam: float, **kw):
    """Dispatch on the device of the start state: HIP tensors run the native block."""
    if start.h.is_cuda:
        eng = getattr(world_model, "_imagine_engine", None)
        if eng is None or eng.actor is not actor or eng.critic is not critic:
            eng = ImagineEngine(world_model, actor, critic)
            object.__setattr__(world_model, "_imagine_engine", eng)
        return eng.run(start, horizon, gamma, lam, **kw)
    return imagine_eager(world_model, actor, critic, start, horizon, gamma, lam)
