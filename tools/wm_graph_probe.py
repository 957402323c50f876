"""Diagnostic: the same updates with the world-model pass eager and recorded (DRM_WM_GRAPH=1): losses side by side, time per update."""
import os, sys, time
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
np.random.seed(0)
learner = Learner(cfg, (3, 64, 64), 6)
rng = np.random.default_rng(0)
cap = 4096
buf = ReplayBuffer(cap, (3, 64, 64), cfg.batch_length, dev)
buf.store_many(rng.integers(0, 256, size=(cap, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=cap),
               rng.standard_normal(cap).astype(np.float32), rng.random(cap) > 0.01)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
if os.environ.get("PROBE_ON_LEARNER_STREAM", "0") == "1" and learner._stream is not None:   # the caller works on the learner's stream
    learner.make_stream_current()
for i in range(n):
    out = learner.update(buf.sample(cfg.batch_size, want_f32=True))
    torch.cuda.synchronize()
    print(i, "wm %.6f pred %.6f dyn %.6f rep %.6f critic %.6f actor %.6f" % tuple(
        float(out[k]) for k in ("world_model_loss", "pred_loss", "dyn_loss", "rep_loss", "critic_loss", "actor_loss")), flush=True)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20):
    learner.update(buf.sample(cfg.batch_size, want_f32=True))
torch.cuda.synchronize()
print("ms per update %.3f" % ((time.perf_counter() - t) / 20 * 1e3), flush=True)
if learner.wm_graph:
    g = learner._wm_full
    # each of these recordings starts from its own inputs, so replaying one alone is safe
    for name, fn in (("wm loss+backward graph", g.g_loss.replay), ("wm clip+step graph", g.g_step.replay),
                     ("critic loss graph", learner._critic_graph.g_loss.replay), ("actor loss graph", learner._actor_graph.g_loss.replay)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fn()
        host = (time.perf_counter() - t) / 20 * 1e3
        torch.cuda.synchronize()
        print("%-26s %.3f ms per replay (host enqueue %.3f ms)" % (name, (time.perf_counter() - t) / 20 * 1e3, host), flush=True)
