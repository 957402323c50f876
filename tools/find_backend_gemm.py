import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner
dev = torch.device("cuda:0")
cfg = Config(device=dev)
L = Learner(cfg, (3, 64, 64), 6)
rng = np.random.default_rng(0)
buf = ReplayBuffer(4096, (3, 64, 64), 64, dev)
buf.store_many(rng.integers(0, 256, size=(4096, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=4096),
               rng.standard_normal(4096).astype(np.float32), rng.random(4096) > 0.01)
for _ in range(5):
    L.update(buf.sample(16, want_f32=True))
torch.cuda.synchronize()
import traceback
orig = {}
def wrap(name):
    f = getattr(torch, name)
    orig[name] = f
    def g(*a, **k):
        big = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() >= 1 << 20]
        if big:
            print(name, [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)], "from", "".join(traceback.format_stack(limit=4)[:-1]).replace("\n", " | ")[:400])
        return f(*a, **k)
    setattr(torch, name, g)
for n in ("mm", "addmm", "matmul"):
    wrap(n)
import torch.nn.functional as F
oF = F.linear
def lin(x, w, b=None):
    if x.is_cuda and (x.numel() >= 1 << 20 or w.numel() >= 1 << 20):
        print("F.linear", tuple(x.shape), tuple(w.shape), "from", "".join(traceback.format_stack(limit=4)[:-1]).replace("\n", " | ")[:400])
    return oF(x, w, b)
F.linear = lin
ot = torch.Tensor.__matmul__
def mm(a, b):
    if a.is_cuda and (a.numel() >= 1 << 20 or b.numel() >= 1 << 20):
        print("@", tuple(a.shape), tuple(b.shape), "from", "".join(traceback.format_stack(limit=4)[:-1]).replace("\n", " | ")[:400])
    return ot(a, b)
torch.Tensor.__matmul__ = mm
L.update(buf.sample(16, want_f32=True))
torch.cuda.synchronize()
print("done")
