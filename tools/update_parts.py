"""Diagnostic: device time of the parts of one recorded training update (events on the learner's stream)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner
from dreamer_amd.imagine import imagine
from dreamer_amd.world_model import WorldModelState

dev = torch.device("cuda:0")
cfg = Config(device=dev)
L = Learner(cfg, (3, 64, 64), 6)
cap = 4096
rng = np.random.default_rng(0)
buf = ReplayBuffer(cap, (3, 64, 64), 64, dev)
buf.store_many(rng.integers(0, 256, size=(cap, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=cap),
               rng.standard_normal(cap).astype(np.float32), rng.random(cap) > 0.01)
L.make_stream_current()
for _ in range(8):
    L.update(buf.sample(16, want_f32=True))
torch.cuda.synchronize()
names = ["world model (graph + optimiser)", "imagination", "critic update", "return scale", "actor update"]
tot = [0.0] * len(names)
N = 20
for _ in range(N):
    b = buf.sample(16, want_f32=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ev[0].record()
    L._wm_full.run(b["observations_f32"], b["actions"], b["rewards"], b["continues"])
    info = L._wm_aux
    ev[1].record()
    st = info["state"]
    im = imagine(L.world_model, L.actor, L.critic, WorldModelState(h=st.h.detach(), z=st.z.detach()), cfg.imagination_horizon,
                 cfg.gamma, cfg.lam, persistent=True)
    ev[2].record()
    L._critic_graph.run(im["h"], im["z"], im["lam_returns"])
    ev[3].record()
    scale = L.actor.ret_scale.update_get_S(im["lam_returns"].detach())
    ev[4].record()
    L._actor_graph.run(im["h"], im["z"], im["actions"], im["lam_returns"], im["values"], scale.reshape(()))
    ev[5].record()
    torch.cuda.synchronize()
    for i in range(len(names)):
        tot[i] += ev[i].elapsed_time(ev[i + 1])
for n, t in zip(names, tot):
    print(f"{n:34s} {t / N:7.3f} ms")
print(f"{'sum':34s} {sum(tot) / N:7.3f} ms")
