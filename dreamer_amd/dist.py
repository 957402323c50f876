"""Data-parallel plumbing: one process per GPU, gradients averaged with ONE collective per model.

Trajectories are independent and every loss is a batch mean, so with equal per-rank batches the
mean of the ranks' gradients equals the gradient of the global batch.  Each model's gradients live
in one flat fp32 buffer (parameter ``.grad`` tensors are views into it), so the all-reduce is a
single RCCL call on a contiguous buffer instead of ~100 small ones, and no copy in or out is needed.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(device=None):
    """Join the job a launcher (torchrun) described in the environment: WORLD_SIZE > 1 and no process
    group yet -> one is created (RCCL for a GPU device, gloo for CPU).  Returns the device this rank
    should use (``cuda:LOCAL_RANK`` when the configured device is a GPU)."""
    import os
    n = int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dev.type == "cuda":
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
    if n > 1 and dist.is_available() and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if dev.type == "cuda":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    return dev


def _via_host(t: torch.Tensor) -> bool:
    """gloo moves device tensors through the host itself, and not reliably for ranks that share a GPU
    (the one-GPU rehearsal of the multi-process path): stage such tensors explicitly."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_sum(t: torch.Tensor) -> None:
    if _via_host(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


class FlatGrads:
    """Flat gradient buffer of a module.

    Built after the first backward pass from the parameters that actually received a gradient
    (``seq.norm.*`` and the critic's frozen ``slow.*`` copy never do; they are left out on every rank
    alike).  Afterwards ``zero()`` replaces ``optimizer.zero_grad()`` and backward accumulates
    straight into the buffer.
    """

    def __init__(self, module: torch.nn.Module, static: bool = False):
        self.module = module
        self.flat: Optional[torch.Tensor] = None
        self.flat_params: Optional[torch.Tensor] = None
        self.params: List[torch.nn.Parameter] = []
        self.offsets: List[int] = []
        self.static = False
        self.seen = None          # static mode, once gradients were gathered: slots that have received one at least once
        self.channels_last_kernels = os.environ.get("DRM_TOWER_CHANNELS_LAST", "1") == "1" and any(
            p.is_cuda for p in module.parameters())
        if static:
            self._build_static()

    def _build(self) -> None:
        self.params = [p for p in self.module.parameters() if p.grad is not None]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        o = 0
        for p in self.params:
            view = self.flat[o:o + p.numel()].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            self.offsets.append(o)
            o += p.numel()

    def _build_static(self) -> None:
        """Flat buffers for the gradients AND the parameters of every trainable parameter, laid out
        before the first step (each tensor starts on a 256-byte boundary): what a flat optimiser step
        needs.  In this mode backward writes ordinary ``.grad`` tensors (``zero()`` drops them, so autograd
        hands over each gradient without an accumulation kernel per parameter) and ``sync()`` gathers them
        into the flat buffer with one multi-tensor copy.  Parameters that receive no gradient keep a
        zero slot, which leaves them unchanged."""
        self.static = True
        self.params = [p for p in self.module.parameters() if p.requires_grad]
        ref = self.params[0]
        o = 0
        for p in self.params:
            self.offsets.append(o)
            o += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(o, dtype=ref.dtype, device=ref.device)
        self.flat_params = torch.zeros(o, dtype=ref.dtype, device=ref.device)

        self.views = [self.slot_view(self.flat, i) for i in range(len(self.params))]
        self._missing = None
        with torch.no_grad():
            for i, p in enumerate(self.params):
                pv = self.slot_view(self.flat_params, i)
                pv.copy_(p.data)
                p.data = pv

    def slot_view(self, buf: torch.Tensor, i: int) -> torch.Tensor:
        """Parameter ``i``'s slot of a flat buffer laid out like ``flat``, with the parameter's logical shape.
        Convolution kernels keep their logical (O, I, kh, kw) shape; with channels-last towers their slot is laid
        out (O, kh, kw, I), the format the convolution backend then takes without transposing them on every call."""
        p, off = self.params[i], self.offsets[i]
        if p.dim() == 4 and self.channels_last_kernels:
            return buf[off:off + p.numel()].view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
        return buf[off:off + p.numel()].view_as(p)

    def zero(self) -> None:
        if self.flat is None or self.static:
            for p in self.module.parameters():
                p.grad = None
        else:
            self.flat.zero_()

    def _gather(self) -> None:
        have = [i for i, p in enumerate(self.params) if p.grad is not None]
        missing = frozenset(range(len(self.params))) - frozenset(have)
        if missing != self._missing:   # slots of parameters without a gradient hold zeros
            for i in missing:
                self.views[i].zero_()
            self._missing = missing
            self.seen = (self.seen if self.seen is not None else set()) | set(have)
        with torch.no_grad():
            torch._foreach_copy_([self.views[i] for i in have], [self.params[i].grad for i in have])

    def sync(self) -> None:
        """Average the gradients over the data-parallel ranks (no-op for a single process)."""
        if self.flat is None:
            self._build()
        if self.static:
            self._gather()
        if world_size() > 1:
            all_reduce_sum(self.flat)
            self.flat.div_(world_size())


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Give every rank rank-``src``'s parameters and buffers (identical replicas at start): one
    collective per dtype on a flat copy."""
    if world_size() == 1:
        return
    with torch.no_grad():
        by_dtype = {}
        for t in list(module.parameters()) + list(module.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t.data)
        for tensors in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in tensors])
            if _via_host(flat):
                h = flat.cpu()
                dist.broadcast(h, src=src)
                flat.copy_(h)
            else:
                dist.broadcast(flat, src=src)
            o = 0
            for t in tensors:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()


def gather_flat(x: torch.Tensor) -> torch.Tensor:
    """Concatenation over ranks of a flat tensor of equal length on every rank (rank order)."""
    if world_size() == 1:
        return x
    if _via_host(x):
        h = x.cpu().contiguous()
        out = torch.empty(world_size() * h.numel(), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h)
        return out.to(x.device)
    out = torch.empty(world_size() * x.numel(), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous())
    return out
