"""cProfile of the host side of Learner.update (where the launch-issuing thread spends its time)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = Config(device=dev)
torch.manual_seed(0)
learner = Learner(cfg, (3, 64, 64), 6)
rng = np.random.default_rng(0)
cap = 4096
buf = ReplayBuffer(cap, (3, 64, 64), cfg.batch_length, dev)
buf.store_many(rng.integers(0, 256, size=(cap, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=cap),
               rng.standard_normal(cap).astype(np.float32), rng.random(cap) > 0.01)
for _ in range(8):
    learner.update(buf.sample(cfg.batch_size, want_f32=True))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    learner.update(buf.sample(cfg.batch_size, want_f32=True))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
