"""Adam over the flat parameter / gradient buffers of ``dist.FlatGrads`` (GPU only).

Replaces ``clip_grad_norm_`` + ``torch.optim.Adam(fused=True).step()`` of the reference's update
(lib/trainer.py:95-103, :170-173, :231-235, :244-247) by two HBM-bound launches of the native library;
the arithmetic is torch's (global-norm clip coefficient min(1, c / (norm + 1e-6)); Adam without weight
decay, bias corrections from the step count).  ``state_dict`` has torch's layout, so checkpoints keep the
reference's format.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from .dist import FlatGrads


class FlatAdam:
    def __init__(self, grads: FlatGrads, lr: float, eps: float, clip: float, betas=(0.9, 0.999)):
        if grads.flat_params is None:
            raise ValueError("FlatAdam needs FlatGrads(module, static=True)")
        self.grads, self.lr, self.eps, self.clip, self.betas = grads, float(lr), float(eps), float(clip), betas
        flat = grads.flat
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._ws = torch.empty(N.lib().drm_adam_workspace_bytes() // 4, dtype=torch.float32, device=flat.device)

    def step(self) -> None:
        """Clip by global norm, then one Adam step; gradients stay unscaled in the buffer."""
        g = self.grads
        N.check(N.lib().drm_adam_flat_step(g.flat_params.data_ptr(), g.flat.data_ptr(), self.exp_avg.data_ptr(),
                                           self.exp_avg_sq.data_ptr(), g.flat.numel(), self.lr, self.betas[0], self.betas[1],
                                           self.eps, self.clip, self.step_count.data_ptr(), self._ws.data_ptr(),
                                           self.grad_norm.data_ptr(), N.stream_ptr(g.flat.device)), "drm_adam_flat_step")

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.grads.zero()

    def _param_ids(self):
        """Index of each flat-buffer parameter in ``list(module.parameters())``: the ids torch.optim.Adam gives the
        parameters when it is built from ``module.parameters()`` as the reference does (lib/trainer.py:95-103), frozen
        ones (the critic's ``slow.*`` copy, lib/critic.py:51-53) included in the count."""
        everything = list(self.grads.module.parameters())
        where = {id(p): i for i, p in enumerate(everything)}
        return len(everything), [where[id(p)] for p in self.grads.params]

    def state_dict(self):
        """torch.optim.Adam's layout: one param group listing every parameter of the module, and state for the
        parameters that have received a gradient at least once (torch creates a parameter's state at its first step
        with a gradient; ``seq.norm.*`` and frozen copies never get one)."""
        g = self.grads
        n_all, ids = self._param_ids()
        seen = g.seen if g.seen is not None else range(len(g.params))
        state = {}
        for slot in sorted(seen):
            p, off = g.params[slot], g.offsets[slot]
            sl = slice(off, off + p.numel())
            state[ids[slot]] = {"step": self.step_count.reshape(()).clone(), "exp_avg": g.slot_view(self.exp_avg, slot).clone(),
                                "exp_avg_sq": g.slot_view(self.exp_avg_sq, slot).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": True, "differentiable": False, "fused": True,
                 "decoupled_weight_decay": False, "params": list(range(n_all))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd) -> None:
        """Accepts what ``state_dict`` writes and what ``torch.optim.Adam(module.parameters()).state_dict()`` writes."""
        g = self.grads
        n_all, ids = self._param_ids()
        listed = sum(len(grp["params"]) for grp in sd["param_groups"])
        if listed != n_all:
            raise ValueError(f"optimizer state lists {listed} parameters, the module has {n_all}")
        if g.seen is None:
            g.seen = set()      # from here on: the slots with state in the file, plus whatever receives a gradient later
        for slot, pid in enumerate(ids):
            st = sd["state"].get(pid)
            if st is None:
                continue
            g.slot_view(self.exp_avg, slot).copy_(st["exp_avg"])
            g.slot_view(self.exp_avg_sq, slot).copy_(st["exp_avg_sq"])
            self.step_count.fill_(float(st["step"]))
            g.seen.add(slot)
