"""Diagnostic: parts of the world-model update, eager against replayed from a recorded graph (device time per pass)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from dreamer_amd.config import Config
from dreamer_amd.trainer import Learner

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
S = torch.cuda.Stream()
torch.cuda.set_stream(S)
cfg = Config(device=dev)
torch.manual_seed(0)
learner = Learner(cfg, (3, 64, 64), 6)
wm = learner.world_model
B, T, L, K, D = 16, 64, cfg.num_latents, cfg.classes_per_latent, cfg.hidden_size
obs = torch.rand(B, T, 3, 64, 64, device=dev)
actions = torch.randint(0, 6, (B, T), device=dev)
rewards = torch.randn(B, T, device=dev)
conts = torch.rand(B, T, device=dev) > 0.01
hf = torch.randn(B * T, D, device=dev, requires_grad=True)
zf = F.one_hot(torch.randint(0, K, (B * T, L), device=dev), K).float().requires_grad_(True)
frames = obs.reshape(B * T, 3, 64, 64)


def zero():
    for p in wm.parameters():
        p.grad = None


def enc():
    zero(); wm.enc.conv(frames).square().mean().backward()


def dec():
    zero(); wm.dec(hf, zf).square().mean().backward()


def heads():
    zero(); (wm.rew(hf, zf).mean() + wm.cont(hf, zf).mean() + wm.dyn(hf).mean()).backward()


def scan():
    zero()
    h, z, post, last = wm.observe(obs, actions, conts)
    (h.mean() + z.mean() + post.mean()).backward()


def full():
    zero(); wm.loss(obs, actions, rewards, conts)[0].backward()


for name, fn in (("encoder tower fwd+bwd", enc), ("decoder fwd+bwd", dec), ("reward/continue/dyn heads fwd+bwd", heads),
                 ("observe (encoder + scan) fwd+bwd", scan), ("whole loss fwd+bwd", full)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / 10 * 1e3
    g = torch.cuda.CUDAGraph()
    gc.collect(); gc.disable()
    with torch.cuda.graph(g, stream=S):
        fn()
    gc.enable()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        g.replay()
    host = (time.perf_counter() - t) / 10 * 1e3
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t) / 10 * 1e3
    print(f"{name:36s} eager {eager:7.3f} ms | graph replay {rep:7.3f} ms (host {host:6.3f} ms)", flush=True)
