"""Two data-parallel ranks: the flat-gradient gather + all-reduce + flat Adam step of each rank equals a single
process on the concatenated batch.  On one GPU the ranks share the device and the collectives run over gloo (device
tensors staged through the host); with two or more GPUs visible the same check runs with one rank per GPU over
RCCL (torch's ``nccl`` backend), the path ``bench.py --gpus N`` and ``train`` take under torchrun."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import DIMS, build_models, observe_inputs, rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dreamer_amd import dist as D
        from dreamer_amd.optim import FlatAdam
        d, salt, B = DIMS["small"], 5, 4
        wm, _, _ = build_models(d, salt + 10 * rank)
        wm = wm.to(dev)
        D.broadcast_module(wm)
        ref_wm, _, _ = build_models(d, salt)
        ref_wm = ref_wm.to(dev)
        for (k, p), (_, q) in zip(wm.named_parameters(), ref_wm.named_parameters()):
            assert torch.equal(p, q), k
        flat = D.FlatGrads(wm, static=True)
        opt = FlatAdam(flat, lr=3e-4, eps=1e-8, clip=1000.0)
        opt_ref = torch.optim.Adam(ref_wm.parameters(), lr=3e-4, eps=1e-8, fused=True)
        obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
        obs = (obs.float() / 255.0).to(dev)
        actions, rewards, continues = actions.to(dev), rewards.to(dev), continues.to(dev)
        noise = {k: v.to(dev) for k, v in noise.items()}
        rows = slice(rank * B // world, (rank + 1) * B // world)
        local_noise = {"init": noise["init"][rows], "reset": noise["reset"][:, rows].contiguous(), "z": noise["z"][:, rows].contiguous()}
        for step in range(2):
            flat.zero()
            loss, _ = wm.loss(obs[rows], actions[rows], rewards[rows], continues[rows], noise=local_noise)
            loss.backward()
            flat.sync()
            opt_ref.zero_grad(set_to_none=True)
            full, _ = ref_wm.loss(obs, actions, rewards, continues, noise=noise)
            full.backward()
            for (k, q), view in zip(((k, q) for k, q in ref_wm.named_parameters() if q.requires_grad), flat.views):
                if q.grad is None:
                    assert float(view.abs().max()) == 0.0, k
                else:
                    assert rel_err(view.cpu(), q.grad.cpu()) < 2e-5, (step, k)
            torch.nn.utils.clip_grad_norm_(ref_wm.parameters(), 1000.0)
            opt_ref.step()
            opt.step()
            for (k, p), (_, q) in zip(wm.named_parameters(), ref_wm.named_parameters()):
                assert rel_err(p.detach().cpu(), q.detach().cpu()) < 1e-5, (step, k)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_process():
    mp.spawn(_worker, args=(2, _free_port(), "gloo"), nprocs=2, join=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one rank per GPU over RCCL")
def test_two_ranks_over_rccl_equal_single_process():
    mp.spawn(_worker, args=(2, _free_port(), "nccl"), nprocs=2, join=True)
