"""Diagnostic: wall time of the phases of one training update on the GPU (synchronising between phases)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner
from dreamer_amd.imagine import imagine
from dreamer_amd.world_model import WorldModelState
import torch.nn as nn
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = Config(device=dev)
L = Learner(cfg, (3, 64, 64), 6)
cap = 4096
rng = np.random.default_rng(0)
buf = ReplayBuffer(cap, (3, 64, 64), 64, dev)
buf.store_many(rng.integers(0, 256, size=(cap, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=cap),
               rng.standard_normal(cap).astype(np.float32), rng.random(cap) > 0.01)
for _ in range(8):
    L.update(buf.sample(16, want_f32=True))
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
N = 10
wm = L.world_model
for _ in range(N):
    t = time.perf_counter()
    batch = buf.sample(16, want_f32=True); t = tick("gather", t)
    obs = batch["observations_f32"]
    B, T = 16, 64
    # world-model loss, piecewise (same code as WorldModel.loss)
    noise = {"reset": torch.empty(T, B, 32, 32, device=dev).exponential_(), "z": torch.empty(T, B, 32, 32, device=dev).exponential_(),
             "init": torch.empty(B, 32, 32, device=dev).exponential_()}
    t = tick("noise", t)
    total, info = wm.loss(obs, batch["actions"], batch["rewards"], batch["continues"], noise=noise); t = tick("wm.loss fwd", t)
    L.g_wm.zero(); total.backward(); t = tick("wm backward", t)
    L.g_wm.sync(); nn.utils.clip_grad_norm_(wm.parameters(), 1000.0); L.optim_world_model.step(); t = tick("wm clip+adam", t)
    st = info["state"]
    im = imagine(wm, L.actor, L.critic, WorldModelState(h=st.h.detach(), z=st.z.detach()), 15, 0.997, 0.95); t = tick("imagine (with repack)", t)
    Hs = WorldModelState(h=im["h"], z=im["z"])
    cl = L.critic.loss(Hs, im["lam_returns"]); L.g_critic.zero(); cl.backward(); L.g_critic.sync()
    nn.utils.clip_grad_norm_(L.critic.parameters(), 100.0); L.optim_critic.step(); L.critic.update_slow(); t = tick("critic update", t)
    al = L.actor.loss(Hs, im["actions"], im["lam_returns"], im["values"]); L.g_actor.zero(); al.backward(); L.g_actor.sync()
    nn.utils.clip_grad_norm_(L.actor.parameters(), 100.0); L.optim_actor.step(); t = tick("actor update", t)
for k, v in acc.items():
    print(f"{k:24s} {v / N * 1e3:8.2f} ms")
print("sum", sum(acc.values()) / N * 1e3)
# finer: pieces of wm.loss forward
import torch.nn.functional as F
acc.clear()
for _ in range(N):
    obs = buf.sample(16, want_f32=True)["observations_f32"]
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        feat = wm.enc.conv(obs.reshape(B * T, 3, 64, 64)); t = tick("enc.conv fwd (1024 frames)", t)
        hf = torch.randn(B * T, 512, device=dev); zf = torch.zeros(B * T, 32, 32, device=dev)
        x = wm.dec(hf, zf); t = tick("dec fwd (1024 frames)", t)
        p = wm.dyn(hf); r = wm.rew(hf, zf); c = wm.cont(hf, zf); t = tick("dyn+rew+cont fwd", t)
for k, v in acc.items():
    print(f"{k:28s} {v / N * 1e3:8.2f} ms")
