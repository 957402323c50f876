"""Training loop with the reference's entry point ``train(cfg, summary_writer=None)``
(lib/trainer.py:16-342) on top of the MI355X hot path.

Per environment step: one posterior step + actor draw (B = 1), store into the device-resident
replay ring, then as many updates as the train-ratio credit allows.  One update
(lib/trainer.py:162-247):

    replay gather (HIP)  ->  world-model loss: hoisted convs, native observe scan, batched heads
    -> backward, [data-parallel gradient all-reduce], clip, Adam
    -> imagination block (native: actor / prior step x H, heads, critic value, lambda-return)
    -> critic update (+ EMA copy)  ->  actor update

Data parallel: launch with torchrun; every rank owns a replica, its own environment and replay ring
(numpy seed offset by the rank) and a per-rank batch; gradients are averaged with one all-reduce per
model before clipping, the actor's return percentiles are taken over the returns of all ranks, and
only rank 0 logs and writes checkpoints.
"""
from __future__ import annotations

import contextlib
import gc
import os
import time
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import dist as D
from .actor import Actor
from .config import Config
from .critic import Critic
from .imagine import imagine
from .optim import FlatAdam
from .replay_buffer import ReplayBuffer
from .utils import log_episode_video, log_unimix, log_wm_imagination_video, log_wm_reconstruction_video, make_env
from .world_model import WorldModel, WorldModelState


def _adam(params, lr, eps, capturable=False):
    params = list(params)
    return torch.optim.Adam(params, lr=lr, eps=eps, fused=params[0].is_cuda, capturable=capturable and params[0].is_cuda)


def _sync_check() -> bool:
    """DRM_SYNC_CHECK: the library synchronises after every launch, which no recording allows; updates stay eager then."""
    return os.environ.get("DRM_SYNC_CHECK", "0") not in ("", "0")


@contextlib.contextmanager
def _recording(graph, stream=None):
    """torch.cuda.graph(graph) with the cyclic garbage collector held off: a collection that runs while
    the stream is recording may free pinned memory, streams or other graphs, calls the runtime refuses
    during a recording (the process then aborts in a destructor)."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        # thread-local error mode: runtime calls of other threads (a collective backend's watchdog polling its events)
        # must not invalidate a recording they have no part in
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            yield
    finally:
        if was_enabled:
            gc.enable()


def _clip_and_step(optim, params, clip: float) -> None:
    if isinstance(optim, FlatAdam):
        optim.step()          # the clip is part of the flat step
    else:
        nn.utils.clip_grad_norm_(params, clip)
        optim.step()


class _GraphedUpdate:
    """One small-model update (loss -> backward -> [gradient all-reduce] -> clip -> Adam -> post) replayed
    as two HIP graphs around the all-reduce.

    The actor and critic updates are a few hundred tiny kernels on B*H rows; run eagerly they cost the
    host's launch time (~1.7 ms each), replayed from a graph they cost their GPU time.  Inputs are
    copied into fixed buffers; the loss and every intermediate live in the graph's memory pool.
    """

    def __init__(self, loss_fn, grads: D.FlatGrads, clip: float, optim: torch.optim.Optimizer, post=None):
        self.loss_fn, self.grads, self.clip, self.optim, self.post = loss_fn, grads, clip, optim, post
        self.static = None
        self.g_loss = self.g_step = None
        self.loss_out = None

    def matches(self, inputs) -> bool:
        """Whether ``inputs`` have the shapes and types the recording was made with (always true before it is made)."""
        return self.static is None or (len(self.static) == len(inputs) and all(
            d.shape == s.shape and d.dtype == s.dtype for d, s in zip(self.static, inputs)))

    def run(self, *inputs: torch.Tensor) -> torch.Tensor:
        if self.g_loss is None:
            self._capture(inputs)
        else:
            for dst, src in zip(self.static, inputs):
                dst.copy_(src)
        self.g_loss.replay()
        self.grads.sync()
        self.g_step.replay()
        return self.loss_out

    def _capture(self, inputs) -> None:
        self.static = [t.detach().clone() for t in inputs]
        params = [p for p in self.grads.module.parameters() if p.requires_grad]
        # the parameters' gradient-accumulation nodes were made on the main stream by the eager updates;
        # autograd notes the change of stream once per node, which is expected here
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        # a caller that runs its updates on a stream of its own gets the recording on that same stream: autograd's
        # gradient-accumulation nodes then belong to the recording stream and the graph stays one chain
        own = torch.cuda.current_stream() != torch.cuda.default_stream()
        side = torch.cuda.current_stream() if own else torch.cuda.Stream()
        self._capture_stream = side if own else None
        side.wait_stream(torch.cuda.current_stream())
        rng = torch.cuda.get_rng_state()
        with torch.cuda.stream(side):   # settle allocator / autograd state on a side stream: loss + backward only
            for _ in range(2):
                self.grads.zero()
                self.loss_fn(*self.static).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        torch.cuda.set_rng_state(rng)   # the rehearsal drew noise; the recorded update must continue the eager sequence
        self.g_loss = torch.cuda.CUDAGraph()
        dot = os.environ.get("DRM_GRAPH_DOT")           # diagnostic: node list of the recorded update as a DOT file
        if dot:
            self.g_loss.enable_debug_mode()
        with _recording(self.g_loss, self._capture_stream):
            self.grads.zero()
            self.loss_out = self.loss_fn(*self.static)
            self.loss_out.backward()
        if dot:
            self.g_loss.debug_dump(f"{dot}.{len(self.static)}in.dot")
        self._capture_step(params)

    def _capture_step(self, params) -> None:
        self.g_step = torch.cuda.CUDAGraph()
        with _recording(self.g_step, getattr(self, "_capture_stream", None)):
            _clip_and_step(self.optim, params, self.clip)
            if self.post is not None:
                self.post()

    def step_only(self) -> None:
        """clip -> Adam -> post for gradients that are already in the flat buffer (and reduced)."""
        if self.g_step is None:
            self._capture_step([p for p in self.grads.params])
        self.g_step.replay()


class ActingPolicy:
    """The policy side of the environment loop (lib/trainer.py:126-151): frame -> posterior state ->
    action, with the recurrent state, the previous action and the continue flag kept on the device.

    On the GPU the whole step (u8 -> f32, encoder, fused scan step, actor, action draw, state update) is
    recorded once into a HIP graph after a few eager steps and replayed per environment step; per step
    the host copies one 12 KB frame in and reads one index back.  The graph reads the parameters in
    place, so optimiser steps between replays are seen (DRM_NO_ACT_GRAPH=1 keeps the step eager).
    """

    def __init__(self, world_model: WorldModel, actor: Actor, device, obs_shape):
        self.wm, self.actor = world_model, actor
        self.device = torch.device(device)
        on_gpu = self.device.type == "cuda"
        self.use_graph = on_gpu and os.environ.get("DRM_NO_ACT_GRAPH", "0") != "1" and not _sync_check()
        self.frame = torch.zeros(1, *obs_shape, dtype=torch.uint8, device=self.device)
        self.stage = torch.zeros(1, *obs_shape, dtype=torch.uint8, pin_memory=True) if on_gpu else None
        self.h = torch.zeros(1, world_model.hidden_size, device=self.device)
        self.z = torch.zeros(1, world_model.num_latents, world_model.classes_per_latent, device=self.device)
        self.last_action = torch.zeros(1, dtype=torch.long, device=self.device)
        self.last_cont = torch.ones(1, 1, dtype=torch.float32, device=self.device)
        self.action = torch.zeros(1, dtype=torch.long, device=self.device)
        self.graph = None
        self.eager_steps = 0
        self.reset()

    @property
    def state(self) -> WorldModelState:
        return WorldModelState(h=self.h, z=self.z)

    @torch.no_grad()
    def reset(self) -> None:
        """Start of an episode (lib/trainer.py:145-148): fresh state, previous action 0, continue 1."""
        s = self.wm.init_state(1, self.device)
        self.h.copy_(s.h)
        self.z.copy_(s.z)
        self.last_action.zero_()
        self.last_cont.fill_(1.0)

    @torch.no_grad()
    def _step(self) -> None:
        obs = self.frame.to(torch.float32) / 255.0
        s = self.wm.act_step(self.state, self.last_action, obs, self.last_cont)
        x = torch.cat([s.h, s.z.reshape(1, -1)], dim=-1)
        logits = self.actor.head(self.actor.mlp(x))
        # Categorical(logits=log_unimix(...)).sample() of lib/trainer.py:130, i.e. multinomial's
        # argmax(probs / Exp(1)), without the distribution object's host-side argument checks
        probs = torch.softmax(log_unimix(logits, self.actor.unimix_eps, dim=-1), dim=-1)
        self.action.copy_(torch.argmax(probs / torch.empty_like(probs).exponential_(), dim=-1))
        self.h.copy_(s.h)
        self.z.copy_(s.z)

    @torch.no_grad()
    def act(self, obs) -> int:
        """Action index for the frame ``obs`` (uint8 array of the observation shape)."""
        frame = torch.as_tensor(np.asarray(obs), dtype=torch.uint8)
        if self.stage is not None:
            self.stage.copy_(frame.reshape(self.stage.shape))
            self.frame.copy_(self.stage, non_blocking=True)
        else:
            self.frame.copy_(frame.reshape(self.frame.shape))
        if self.use_graph and self.eager_steps >= 3:
            if self.graph is None:   # recording launches nothing: the step itself is the replay below
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with _recording(self.graph):
                    self._step()
            self.graph.replay()
        else:
            self._step()
            self.eager_steps += 1
        return int(self.action.item())

    def feedback(self, action: int, cont: bool) -> None:
        """What the environment did with the step: the action actually taken and whether the episode goes on."""
        self.last_action.fill_(int(action))
        self.last_cont.fill_(float(cont))
        if not cont:
            self.reset()


class Learner:
    """Models, optimisers and the update step; shared by ``train`` and the benchmark."""

    def __init__(self, cfg: Config, obs_shape, act_size: int):
        self.cfg = cfg
        dev = cfg.device
        self.world_model = WorldModel(obs_shape=obs_shape, action_size=act_size, num_latents=cfg.num_latents,
                                      classes_per_latent=cfg.classes_per_latent, hidden_size=cfg.hidden_size,
                                      base_cnn_channels=cfg.base_cnn_channels, mlp_hidden_units=cfg.mlp_hidden_units,
                                      free_bits=cfg.free_bits, beta_pred=cfg.beta_pred, beta_dyn=cfg.beta_dyn,
                                      beta_rep=cfg.beta_rep, unimix_eps=cfg.unimix_eps).to(dev).train()
        state_size = cfg.hidden_size + cfg.num_latents * cfg.classes_per_latent
        self.actor = Actor(state_size=state_size, action_size=act_size, mlp_hidden_units=cfg.mlp_hidden_units, mlp_layers=2,
                           entropy_scale=cfg.actor_entropy_scale, ret_norm_limit=cfg.actor_ret_norm_limit,
                           ret_norm_decay=cfg.actor_ret_norm_decay, unimix_eps=cfg.unimix_eps).to(dev).train()
        self.critic = Critic(state_size=state_size, mlp_hidden_units=cfg.mlp_hidden_units, mlp_layers=2,
                             num_buckets=cfg.critic_num_buckets, bucket_min=cfg.critic_bucket_min,
                             bucket_max=cfg.critic_bucket_max, ema_decay=cfg.critic_ema_decay,
                             ema_regularizer=cfg.critic_ema_regularizer).to(dev).train()
        for m in (self.world_model, self.actor, self.critic):
            D.broadcast_module(m)
        self.actor.ret_scale.gather = D.gather_flat if D.world_size() > 1 else None
        on_gpu = torch.device(dev).type == "cuda"
        # GPU: parameters and gradients of each model in flat buffers, optimiser step = the library's flat
        # clip + Adam (DRM_TORCH_ADAM=1: torch's fused Adam on the same parameters)
        flat_adam = on_gpu and os.environ.get("DRM_TORCH_ADAM", "0") != "1"
        self.g_wm, self.g_actor, self.g_critic = (D.FlatGrads(m, static=flat_adam)
                                                  for m in (self.world_model, self.actor, self.critic))
        if flat_adam:
            self.optim_world_model = FlatAdam(self.g_wm, cfg.world_model_lr, cfg.world_model_adam_eps, cfg.world_model_grad_clip)
            self.optim_actor = FlatAdam(self.g_actor, cfg.actor_critic_lr, cfg.actor_critic_adam_eps, cfg.actor_critic_grad_clip)
            self.optim_critic = FlatAdam(self.g_critic, cfg.actor_critic_lr, cfg.actor_critic_adam_eps, cfg.actor_critic_grad_clip)
        else:
            self.optim_world_model = _adam(self.world_model.parameters(), cfg.world_model_lr, cfg.world_model_adam_eps,
                                           capturable=on_gpu)
            self.optim_actor = _adam(self.actor.parameters(), cfg.actor_critic_lr, cfg.actor_critic_adam_eps, capturable=on_gpu)
            self.optim_critic = _adam(self.critic.parameters(), cfg.actor_critic_lr, cfg.actor_critic_adam_eps, capturable=on_gpu)
        # actor / critic updates replayed from HIP graphs after a few eager ones (DRM_NO_UPDATE_GRAPHS=1: always eager)
        self.graph_updates = on_gpu and os.environ.get("DRM_NO_UPDATE_GRAPHS", "0") != "1" and not _sync_check()
        self.eager_updates = 0
        self._wm_graph = _GraphedUpdate(None, self.g_wm, cfg.world_model_grad_clip, self.optim_world_model)
        # Every update runs on one stream of the learner's own (DRM_ONE_STREAM=0: on the caller's): the recordings below
        # are then made on the stream the eager warm-up updates ran on, autograd's gradient-accumulation nodes belong to
        # it, and each recorded update is a single chain of nodes that the host launches in ~0.2 ms (recorded from the
        # default stream's context the same update becomes a fork-join graph that takes the host 7.8 ms to launch).
        self._stream = (torch.cuda.Stream(device=dev) if self.graph_updates and os.environ.get("DRM_ONE_STREAM", "1") == "1"
                        else None)
        # The world-model loss + backward recorded too (DRM_WM_GRAPH=0: eager): ~800 launches whose enqueueing costs the
        # host ~10 ms per update when issued one by one; replayed, the host is free for the environment loop and the
        # update takes its device time.
        self.wm_graph = self.graph_updates and self._stream is not None and os.environ.get("DRM_WM_GRAPH", "1") == "1"
        self._wm_aux = None

        def _wm_loss(obs, actions, rewards, continues):
            loss, info = self.world_model.loss(obs, actions, rewards, continues)
            self._wm_aux = info          # after the recording: the graph's own output tensors
            return loss

        self._wm_full = _GraphedUpdate(_wm_loss, self.g_wm, cfg.world_model_grad_clip, self.optim_world_model)
        self._critic_graph = _GraphedUpdate(
            lambda h, z, ret: self.critic.loss(model_states=WorldModelState(h=h, z=z), returns=ret),
            self.g_critic, cfg.actor_critic_grad_clip, self.optim_critic, post=self.critic.update_slow)
        self._actor_graph = _GraphedUpdate(
            lambda h, z, act, ret, val, scale: self.actor.loss_with_scale(WorldModelState(h=h, z=z), act, ret, val, scale),
            self.g_actor, cfg.actor_critic_grad_clip, self.optim_actor)

    def make_stream_current(self) -> None:
        """Make the learner's stream the calling thread's current stream (ordered after what the previous current
        stream has queued).  A loop that samples, updates and acts on this one stream saves the two cross-stream
        hand-overs per update that ``update`` otherwise needs (measured: 10.9 against 11.7 ms per update)."""
        if self._stream is not None and torch.cuda.current_stream() != self._stream:
            self._stream.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(self._stream)

    def update(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """One training update (lib/trainer.py:162-247).  On the GPU the work is issued on the learner's own stream,
        ordered after what the caller's stream has queued so far; the caller's stream waits for it before going on
        (no hand-over when the caller already works on that stream, see ``make_stream_current``)."""
        if self._stream is None:
            return self._update(batch)
        cur = torch.cuda.current_stream()
        if cur == self._stream:
            return self._update(batch)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            out = self._update(batch)
        cur.wait_stream(self._stream)
        return out

    def _update(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        cfg = self.cfg
        obs = batch.get("observations_f32")
        if obs is None:
            obs = batch["observations"].float() / 255.0
        # ---- world model ----
        wm_recorded = False
        wm_inputs = (obs, batch["actions"], batch["rewards"], batch["continues"])
        if self.wm_graph and self.eager_updates >= 3 and self._wm_full.matches(wm_inputs):
            try:
                wm_loss = self._wm_full.run(*wm_inputs).detach().clone()
                wm_info = self._wm_aux
                wm_recorded = True
            except RuntimeError as exc:
                if self._wm_full.g_step is not None:      # a replay failed: not something eager execution would fix
                    raise
                print(f"[dreamer_amd] recording the world-model update failed ({exc}); it stays eager", flush=True)
                self.wm_graph = False
                self._wm_full.g_loss = self._wm_full.g_step = None
        if not wm_recorded:
            wm_loss, wm_info = self.world_model.loss(obs, batch["actions"], batch["rewards"], batch["continues"])
            self.g_wm.zero()
            wm_loss.backward()
            self.g_wm.sync()
        if wm_recorded:
            pass                                    # clip + step were part of the recorded update
        elif isinstance(self.optim_world_model, FlatAdam):
            self.optim_world_model.step()          # two launches: nothing to gain from a recorded graph
        elif self.graph_updates and self.eager_updates >= 3:
            self._wm_graph.step_only()
        else:
            nn.utils.clip_grad_norm_(self.world_model.parameters(), cfg.world_model_grad_clip)
            self.optim_world_model.step()
        # ---- imagination from the last posterior states, with the updated world model ----
        start: WorldModelState = wm_info["state"]
        im = imagine(self.world_model, self.actor, self.critic,
                     WorldModelState(h=start.h.detach(), z=start.z.detach()), cfg.imagination_horizon, cfg.gamma, cfg.lam,
                     **({"persistent": True} if self.graph_updates else {}))
        Hs = WorldModelState(h=im["h"], z=im["z"])
        returns, values, actions = im["lam_returns"], im["values"], im["actions"]
        if self.graph_updates and self.eager_updates >= 3:
            critic_loss = self._critic_graph.run(Hs.h, Hs.z, returns)
            scale = self.actor.ret_scale.update_get_S(returns.detach())   # rank-global percentiles (may all-gather)
            actor_loss = self._actor_graph.run(Hs.h, Hs.z, actions, returns, values, scale.reshape(()))
            return {"world_model_loss": wm_loss.detach(), "actor_loss": actor_loss.detach().clone(),
                    "critic_loss": critic_loss.detach().clone(), "pred_loss": wm_info["pred_loss"].detach(),
                    "dyn_loss": wm_info["dyn_loss"].detach(), "rep_loss": wm_info["rep_loss"].detach(), "imagination": im}
        self.eager_updates += 1
        # ---- critic ----
        critic_loss = self.critic.loss(model_states=Hs, returns=returns)
        self.g_critic.zero()
        critic_loss.backward()
        self.g_critic.sync()
        _clip_and_step(self.optim_critic, self.critic.parameters(), cfg.actor_critic_grad_clip)
        self.critic.update_slow()
        # ---- actor ----
        actor_loss = self.actor.loss(model_states=Hs, actions=actions, returns=returns, values=values)
        self.g_actor.zero()
        actor_loss.backward()
        self.g_actor.sync()
        _clip_and_step(self.optim_actor, self.actor.parameters(), cfg.actor_critic_grad_clip)
        return {"world_model_loss": wm_loss.detach(), "actor_loss": actor_loss.detach(), "critic_loss": critic_loss.detach(),
                "pred_loss": wm_info["pred_loss"].detach(), "dyn_loss": wm_info["dyn_loss"].detach(),
                "rep_loss": wm_info["rep_loss"].detach(), "imagination": im}

    def state_dicts(self):
        return {"world_model": self.world_model.state_dict(), "actor": self.actor.state_dict(),
                "critic": self.critic.state_dict(), "optim_world_model": self.optim_world_model.state_dict(),
                "optim_actor": self.optim_actor.state_dict(), "optim_critic": self.optim_critic.state_dict()}

    def load_state_dicts(self, sd) -> None:
        """Restore models and optimisers from a checkpoint of the layout ``state_dicts`` writes (the reference's,
        lib/trainer.py:326-335).  Parameters are copied in place: the flat parameter buffers, the recorded
        graphs and the packed gather tables of the engines (rebuilt from the weights on every call) stay valid."""
        for name in ("world_model", "actor", "critic"):
            getattr(self, name).load_state_dict(sd[name])
        for name in ("optim_world_model", "optim_actor", "optim_critic"):
            getattr(self, name).load_state_dict(sd[name])


def _log(writer, learner: Learner, out, step: int, ups: float):
    cfg = learner.cfg
    im = out["imagination"]
    ret, V, R, C = im["lam_returns"], im["values"], im["rewards"], im["conts"]
    logp = im["actor_logp"]
    ent = -(logp.exp() * logp).sum(-1).mean().item()
    scale = (learner.actor.ret_scale.p95 - learner.actor.ret_scale.p05).detach().clamp(min=cfg.actor_ret_norm_limit)
    adv = (ret - V).reshape(-1)
    s = writer.add_scalar
    s("policy/ret_scale", scale.item(), step)
    s("policy/adv_mean", adv.mean().item(), step)
    s("policy/adv_std", adv.std(unbiased=False).item(), step)
    s("policy/adv_mean_scaled", (adv / scale).mean().item(), step)
    s("policy/adv_std_scaled", (adv / scale).std(unbiased=False).item(), step)
    s("policy/imagination_entropy", ent, step)
    writer.add_histogram("policy/imagination_probs", logp[:, -1].exp(), step)
    s("value/mae_vs_returns", adv.abs().mean().item(), step)
    s("value/bias_vs_returns", adv.mean().item(), step)
    s("value/mean", V.mean().item(), step)
    s("value/std", V.std(unbiased=False).item(), step)
    s("value/lambda_ret_mean", ret.mean().item(), step)
    s("value/lambda_ret_std", ret.std(unbiased=False).item(), step)
    s("world_model/pred_loss", out["pred_loss"].item(), step)
    s("world_model/dyn_loss", out["dyn_loss"].item(), step)
    s("world_model/rep_loss", out["rep_loss"].item(), step)
    s("world_model/rew_mean", R.mean().item(), step)
    s("world_model/rew_std", R.std(unbiased=False).item(), step)
    s("world_model/cont_mean", C.mean().item(), step)
    s("train/world_model_loss", out["world_model_loss"].item(), step)
    s("train/actor_loss", out["actor_loss"].item(), step)
    s("train/critic_loss", out["critic_loss"].item(), step)
    s("perf/updates_per_second", ups, step)
    print(f"Update {step}, World Model Loss: {out['world_model_loss'].item():.3f}, Actor Loss: {out['actor_loss'].item():.3f}, "
          f"Critic Loss: {out['critic_loss'].item():.3f}, Updates/sec: {ups:.3f}")


def _rng_snapshot():
    """Host and device generator states as tensors (loadable with ``weights_only=True``)."""
    kind, keys, pos, has_gauss, cached = np.random.get_state()
    snap = {"numpy_keys": torch.from_numpy(np.asarray(keys, dtype=np.int64)),
            "numpy_pos": torch.tensor([int(pos), int(has_gauss)], dtype=torch.int64),
            "numpy_gauss": torch.tensor([float(cached)], dtype=torch.float64), "torch_cpu": torch.get_rng_state()}
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        snap["torch_cuda"] = torch.cuda.get_rng_state()
    return snap


def _rng_restore(snap) -> None:
    pos, has_gauss = (int(v) for v in snap["numpy_pos"])
    np.random.set_state(("MT19937", snap["numpy_keys"].cpu().numpy().astype(np.uint32), pos, has_gauss,
                         float(snap["numpy_gauss"][0])))
    torch.set_rng_state(snap["torch_cpu"].cpu())
    if "torch_cuda" in snap and torch.cuda.is_available():
        torch.cuda.set_rng_state(snap["torch_cuda"].cpu())


def _rank_file(directory: str, stem: str, rank: int) -> str:
    """Per-rank file names: rank 0 keeps the plain name (a single-process run reads and writes what it always did)."""
    return os.path.join(directory, f"{stem}.pth" if rank == 0 else f"{stem}.rank{rank}.pth")


def _replace(tmp: str, final: str) -> None:
    os.replace(tmp, final)


def save_checkpoint(directory: str, learner: "Learner", counters: Dict[str, float], buffer: Optional[ReplayBuffer] = None,
                    rank: int = 0) -> None:
    """Rank 0: ``ckpt.pth`` with the reference's six entries (lib/trainer.py:326-335) plus ``trainer_state`` (update /
    step counters, train-ratio credit, generator states; tensors and plain numbers only), written to a temporary name
    and moved into place.  Every other rank of a data-parallel run writes ``rng.rank<r>.pth`` with its own generator
    states (the ranks draw different batches and actions; models and optimisers are identical, so rank 0's copy serves
    all).  With ``buffer`` each rank's replay ring goes to ``replay.pth`` / ``replay.rank<r>.pth``, stamped with the
    counters of this checkpoint; without it a ring file left by an earlier checkpoint is removed, so that it cannot be
    paired with these counters later."""
    os.makedirs(directory, exist_ok=True)
    stamp = (int(counters.get("updates_done", 0)), int(counters.get("policy_steps", 0)))
    replay = _rank_file(directory, "replay", rank)
    if buffer is not None:
        tmp = replay + ".tmp"
        buffer.save(tmp, stamp=stamp)
        _replace(tmp, replay)       # the ring first: a crash before the checkpoint lands leaves a ring whose stamp no
    elif os.path.exists(replay):    # checkpoint carries, and the loader skips it
        os.remove(replay)
    if rank == 0:
        sd = learner.state_dicts()
        sd["trainer_state"] = {"counters": {k: float(v) for k, v in counters.items()}, "rng": _rng_snapshot()}
        tmp = os.path.join(directory, "ckpt.pth.tmp")
        torch.save(sd, tmp)
        _replace(tmp, os.path.join(directory, "ckpt.pth"))
    else:
        tmp = _rank_file(directory, "rng", rank) + ".tmp"
        torch.save({"rng": _rng_snapshot(), "stamp": torch.tensor(stamp, dtype=torch.int64)}, tmp)
        _replace(tmp, _rank_file(directory, "rng", rank))


def load_checkpoint(path: str, learner: "Learner", buffer: Optional[ReplayBuffer] = None, rank: int = 0,
                    verbose: bool = False) -> Dict[str, float]:
    """Restore a learner (and, when this rank's stamped replay file lies beside the checkpoint, the replay ring) from
    ``path`` (a directory or the ``ckpt.pth`` file).  Files are read with ``weights_only=True``: nothing in them is
    executed.  A checkpoint written by the reference (no ``trainer_state``) restores models and optimisers and
    returns {}.  Generator states: rank 0 takes the checkpoint's; another rank takes its own ``rng.rank<r>.pth`` when
    that file carries the checkpoint's stamp, and otherwise reseeds from the checkpoint's state offset by its rank --
    never rank 0's state itself, which would make every rank draw the same batches and actions."""
    ckpt = os.path.join(path, "ckpt.pth") if os.path.isdir(path) else path
    directory = os.path.dirname(ckpt)
    sd = torch.load(ckpt, map_location=learner.cfg.device, weights_only=True)
    learner.load_state_dicts(sd)
    extra = sd.get("trainer_state")
    counters = dict(extra["counters"]) if extra else {}
    stamp = (int(counters.get("updates_done", 0)), int(counters.get("policy_steps", 0))) if extra else None
    replay = _rank_file(directory, "replay", rank)
    if buffer is not None and os.path.exists(replay):
        taken = buffer.load(replay, expect_stamp=stamp)
        if not taken and verbose:
            print(f"{replay} belongs to another checkpoint (stamp differs from {stamp}): replay ring not restored")
    if not extra:
        return {}
    if rank == 0:
        _rng_restore(extra["rng"])
        return counters
    own = _rank_file(directory, "rng", rank)
    if os.path.exists(own):
        mine = torch.load(own, map_location="cpu", weights_only=True)
        if [int(v) for v in mine["stamp"]] == list(stamp):
            _rng_restore(mine["rng"])
            return counters
    _rng_restore(extra["rng"])
    _reseed_for_rank(rank)
    return counters


def _reseed_for_rank(rank: int) -> None:
    """Host and device generators moved to a stream of this rank's own, derived from their current state."""
    if rank == 0:
        return
    base = int(np.random.randint(0, 2 ** 31 - 1))
    np.random.seed((base + 7919 * rank) % (2 ** 31 - 1))
    torch.manual_seed((base + 104729 * rank) % (2 ** 63 - 1))


def train(cfg: Config, summary_writer=None, env=None, verbose: bool = True, resume: Optional[str] = None,
          save_replay: bool = False):
    """The reference's entry point (lib/trainer.py:16).  ``resume``: a checkpoint directory or file to continue from
    (the reference only saves); ``save_replay``: write the replay ring beside every checkpoint so that a resumed
    run continues with the data it had."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        cfg.device = D.init_from_env(cfg.device)
    is_main = D.rank() == 0
    env = env if env is not None else make_env(cfg.env_id)
    act_space = env.action_space
    assert hasattr(act_space, "n"), "Only discrete action space is supported."
    obs_shape, act_size = env.observation_space.shape, int(act_space.n)
    if is_main and verbose:
        print(f"Observation shape: {obs_shape}, Action size: {act_size}")
    learner = Learner(cfg, obs_shape, act_size)
    learner.make_stream_current()      # sampling, updates and acting share one stream from here on
    wm, actor = learner.world_model, learner.actor
    if is_main and verbose:
        count = lambda m: sum(p.numel() for p in m.parameters())
        print(f"World Model parameters: {count(wm):,}\nActor parameters: {count(actor):,}\n"
              f"Critic parameters: {count(learner.critic):,}")
    buffer = ReplayBuffer(capacity=cfg.replay_capacity, obs_shape=obs_shape, seq_len=cfg.batch_length, device=cfg.device)
    if D.world_size() > 1:
        np.random.seed((np.random.randint(0, 2 ** 31 - 1) + 7919 * D.rank()) % (2 ** 31 - 1))
    writer = summary_writer if is_main else None

    time_counter = time.time()
    update_credit, credits_per_update = 0.0, cfg.batch_size * cfg.batch_length
    updates_done = update_counter = policy_steps = 0
    policy = ActingPolicy(wm, actor, cfg.device, obs_shape)
    current_obs = None
    if resume is not None:
        c = load_checkpoint(resume, learner, buffer, rank=D.rank(), verbose=verbose)
        updates_done, policy_steps = int(c.get("updates_done", 0)), int(c.get("policy_steps", 0))
        update_credit = float(c.get("update_credit", 0.0))
        policy.reset()    # the environment starts a new episode; the recurrent state does too
        if is_main and verbose:
            print(f"Resumed from {resume}: {updates_done} updates, {policy_steps} env steps, {len(buffer)} stored transitions")
    for _ in range(max(cfg.num_iterations - policy_steps, 0)):
        if current_obs is None:
            current_obs, _ = env.reset()
        action = policy.act(current_obs)
        if updates_done == 0:
            action = env.action_space.sample()
        next_obs, reward, terminated, truncated, _ = env.step(action)
        cont = not (terminated or truncated)
        buffer.store(current_obs, action, float(reward), cont)
        current_obs = next_obs
        policy.feedback(action, cont)
        if not cont:
            current_obs, _ = env.reset()
        policy_steps += 1
        update_credit += cfg.train_ratio

        while len(buffer) > cfg.batch_length and update_credit >= credits_per_update:
            if is_main and verbose:
                print(f"Update {updates_done}, total env steps {policy_steps}, buffer size {len(buffer)}, "
                      f"update credit {update_credit:.1f}")
            batch = buffer.sample(cfg.batch_size, want_f32=buffer.device.type == "cuda")
            out = learner.update(batch)
            if writer is not None and updates_done % cfg.log_interval == 0 and updates_done > 0:
                elapsed = time.time() - time_counter
                time_counter = time.time()
                with torch.no_grad():
                    _log(writer, learner, out, updates_done, update_counter / max(elapsed, 1e-9))
                update_counter = 0
            if updates_done % cfg.video_interval == 0 and updates_done > 0 and writer is not None:
                wm.eval(); actor.eval()
                log_episode_video(cfg, writer, env, wm, actor, updates_done)
                log_wm_reconstruction_video(cfg, writer, env, wm, actor, updates_done)
                log_wm_imagination_video(cfg, writer, env, wm, actor, updates_done)
                wm.train(); actor.train()
            update_credit -= credits_per_update
            updates_done += 1
            update_counter += 1
            if (updates_done - 1) % cfg.save_interval == 0 and updates_done > 1 and cfg.checkpoint_dir is not None:
                # after update number `updates_done - 1` (the reference's save point, lib/trainer.py:326-335), with
                # the counters as they stand once that update is accounted for; rank 0 writes the models, every rank
                # its own generator states and (with save_replay) its own ring
                save_checkpoint(cfg.checkpoint_dir, learner, {"updates_done": updates_done, "policy_steps": policy_steps,
                                                              "update_credit": update_credit},
                                buffer if save_replay else None, rank=D.rank())
                if is_main:
                    print(f"Saved model checkpoint to {cfg.checkpoint_dir}")
    env.close()
    return learner
