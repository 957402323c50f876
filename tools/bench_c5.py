"""Diagnostic: the scaled model (recurrent width 4096, 32 x 32 categorical latent) on one GPU: world-model loss + backward
over a 16 x 64 batch and the imagination block, fp32 products against bf16 matrix inputs."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd.actor import Actor
from dreamer_amd.critic import Critic
from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.world_model import WorldModel, WorldModelState

dev = torch.device("cuda:0")
D, L, K, M, A = 4096, 32, 32, 512, 6
torch.manual_seed(0)
wm = WorldModel((3, 64, 64), A, num_latents=L, classes_per_latent=K, hidden_size=D, mlp_hidden_units=M).to(dev).train()
actor = Actor(D + L * K, A, M, 2).to(dev)
critic = Critic(D + L * K, M, 2).to(dev)
B, T = 16, 64
g = torch.Generator().manual_seed(1)
obs = torch.rand(B, T, 3, 64, 64, generator=g).to(dev)
actions = torch.randint(0, A, (B, T), generator=g).to(dev)
rewards = torch.randn(B, T, generator=g).to(dev)
continues = (torch.rand(B, T, generator=g) > 0.02).to(dev)
out = {}


def timeit(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def update():
    wm.zero_grad(set_to_none=True)
    loss, _ = wm.loss(obs, actions, rewards, continues)
    loss.backward()


ONLY = os.environ.get("C5_ONLY", "")          # "imagine16": just the few-row imagination block (for kernel traces)
if ONLY == "ops_bf16":        # which framework ops (by input shape) launch what inside the bf16 update
    from torch.profiler import profile, ProfilerActivity
    wm.observe_precision = "bf16"
    for _ in range(3):
        update()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(2):
            update()
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:70]:
        print(f"{e.self_device_time_total / 2:8.1f} us/update {e.count / 2:6.1f} x {e.self_device_time_total / e.count:7.1f} us  {e.key[:44]:44s} {str(e.input_shapes)[:110]}")
    sys.exit(0)
for prec in (("f32",) if ONLY == "wm" else ("bf16",) if ONLY == "wm_bf16" else () if ONLY else ("f32", "bf16")):
    wm.observe_precision = prec
    out[f"wm_loss_backward_ms_16x64_{prec}"] = timeit(update, 5)
    print(prec, "world-model loss + backward: %.2f ms" % out[f"wm_loss_backward_ms_16x64_{prec}"], flush=True)
if ONLY in ("wm", "wm_bf16"):
    print(json.dumps(out))
    sys.exit(0)
eng = ImagineEngine(wm, actor, critic)
if ONLY == "split1024":       # the split form on the wide model: time and distance from the fp32 matrix instructions
    Bi = 1024
    h0 = torch.randn(Bi, D, generator=g).tanh().to(dev)
    z0 = torch.nn.functional.one_hot(torch.randint(0, K, (Bi, L), generator=g), K).float().to(dev)
    st = WorldModelState(h=h0, z=z0)
    ea = torch.empty(15, Bi, A, device=dev).exponential_(); ez = torch.empty(15, Bi, L, K, device=dev).exponential_()
    res = {}
    for prec in ("f32_mfma", "f32_split"):
        o = eng.run(st, 15, 0.997, 0.95, repack=False, precision=prec, noise={"a": ea, "z": ez})
        res[prec] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
        ms = timeit(lambda: eng.run(st, 15, 0.997, 0.95, repack=False, persistent=True, precision=prec), 5)
        print(prec, "B=1024: %.2f ms per block" % ms, flush=True)
    same = (res["f32_mfma"]["actions"] == res["f32_split"]["actions"]).all(dim=1)
    print("trajectories with identical actions: %d of %d" % (int(same.sum()), Bi))
    for k in ("h", "lam_returns", "values", "rewards"):
        a, b = res["f32_mfma"][k][same], res["f32_split"][k][same]
        print(k, "max |diff| on those: %.3e (scale %.3e)" % (float((a - b).abs().max()), float(a.abs().max())))
    sys.exit(0)
for Bi in ((16,) if ONLY == "imagine16" else (16, 1024)):
    h0 = torch.randn(Bi, D, generator=g).tanh().to(dev)
    z0 = torch.nn.functional.one_hot(torch.randint(0, K, (Bi, L), generator=g), K).float().to(dev)
    st = WorldModelState(h=h0, z=z0)
    for prec in ("f32", "bf16"):
        ms = timeit(lambda: eng.run(st, 15, 0.997, 0.95, repack=False, persistent=True, precision=prec), 5)
        out[f"imagine_ms_B{Bi}_{prec}"] = ms
        print(f"imagine B={Bi} {prec}: {ms:.2f} ms per block = {Bi * 15 / ms * 1e3:.0f} steps/s", flush=True)
print(json.dumps(out))
