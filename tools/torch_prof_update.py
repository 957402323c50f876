"""Diagnostic: which torch ops launch the big reduction / elementwise kernels of one training update."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = Config(device=dev)
L = Learner(cfg, (3, 64, 64), 6)
rng = np.random.default_rng(0)
buf = ReplayBuffer(4096, (3, 64, 64), 64, dev)
buf.store_many(rng.integers(0, 256, size=(4096, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=4096),
               rng.standard_normal(4096).astype(np.float32), rng.random(4096) > 0.01)
for _ in range(8):
    L.update(buf.sample(16, want_f32=True))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        L.update(buf.sample(16, want_f32=True))
    torch.cuda.synchronize()
if os.environ.get("COMPACT") == "1":     # one line per (op, input shapes): device time per update, launches per update, average
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:int(os.environ.get("ROWS", "40"))]:
        print(f"{e.self_device_time_total / 3:8.1f} us/update {e.count / 3:6.1f} x {e.self_device_time_total / e.count:7.1f} us  {e.key[:44]:44s} {str(e.input_shapes)[:110]}")
else:
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get("ROWS", "40")), max_name_column_width=50, max_shapes_column_width=70))
