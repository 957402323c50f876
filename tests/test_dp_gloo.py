"""Data-parallel semantics on CPU with the gloo backend (world size 2): averaged per-rank
gradients equal the gradient of the concatenated batch, the return percentiles are global, and
replicas start identical."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import DIMS, build_models, observe_inputs, rel_err


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from dreamer_amd import dist as D
        from dreamer_amd.actor import EMAPercentileScale
        d, salt, B = DIMS["small"], 5, 4
        # replicas: rank 1 starts from different weights and must end up with rank 0's
        wm, actor, critic = build_models(d, salt + 10 * rank)
        for m in (wm, actor, critic):
            D.broadcast_module(m)
        ref_wm, _, _ = build_models(d, salt)
        for (k, p), (_, q) in zip(wm.named_parameters(), ref_wm.named_parameters()):
            assert torch.equal(p, q), k
        obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
        obs = obs.float() / 255.0
        rows = slice(rank * B // world, (rank + 1) * B // world)
        local_noise = {"init": noise["init"][rows], "reset": noise["reset"][:, rows], "z": noise["z"][:, rows]}
        loss, _ = wm.loss(obs[rows], actions[rows], rewards[rows], continues[rows], noise=local_noise)
        flat = D.FlatGrads(wm)
        flat.zero()
        loss.backward()
        flat.sync()
        # single-process reference over the whole batch
        full, _ = ref_wm.loss(obs, actions, rewards, continues, noise=noise)
        full.backward()
        for (k, p), (_, q) in zip(wm.named_parameters(), ref_wm.named_parameters()):
            if q.grad is None:
                assert p.grad is None, k
            else:
                assert rel_err(p.grad, q.grad) < 1e-5, (k, rel_err(p.grad, q.grad))
        assert flat.flat.numel() == sum(q.numel() for q in ref_wm.parameters() if q.grad is not None)
        # second update re-uses the flat buffer: zero() must clear what backward accumulates into
        flat.zero()
        assert all(float(p.grad.abs().max()) == 0.0 for p in flat.params)
        # global return percentiles
        ret = torch.arange(40, dtype=torch.float32).reshape(4, 10) * (1.0 + rank)
        scale = EMAPercentileScale(0.99, 1.0).train()
        scale.gather = D.gather_flat
        s_local = scale.update_get_S(ret)
        both = torch.cat([torch.arange(40, dtype=torch.float32), torch.arange(40, dtype=torch.float32) * 2.0])
        single = EMAPercentileScale(0.99, 1.0).train()
        s_ref = single.update_get_S(both)
        assert torch.equal(s_local, s_ref) and torch.equal(scale.p05, single.p05) and torch.equal(scale.p95, single.p95)
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process():
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)
