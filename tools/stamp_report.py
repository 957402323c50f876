"""Diagnostic: phase timing inside the small-row linear kernel (needs the DRM_STAMPS library)."""
import ctypes as C, os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DRM_LIB"] = os.path.join(ROOT, "dreamer_amd", "libdreamer_amd_stamps.so")
sys.path.insert(0, ROOT)
import bench
from dreamer_amd import _native as N
from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.world_model import WorldModelState
dev = torch.device("cuda:0")
cfg, wm, actor, critic = bench.build_models(dev)
eng = ImagineEngine(wm, actor, critic)
B, H = 16, 15
h0, z0 = bench.start_states(B, cfg, dev)
st = WorldModelState(h=h0, z=z0)
buf = torch.zeros(1 + 8 * 100000, dtype=torch.int64, device=dev)
fn = N.lib().drm_debug_set_stamp_buffer
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
for _ in range(10):
    eng.run(st, H, 0.997, 0.95, repack=False, persistent=True)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
for _ in range(5):
    eng.run(st, H, 0.997, 0.95, repack=False, persistent=True)
torch.cuda.synchronize()
n = int(buf[0].item())
rec = buf[1:1 + 8 * n].view(n, 8).cpu().numpy()
rec = rec[rec[:, 2].argsort()]
print("records", n)
stats = collections.defaultdict(list); xccs = collections.defaultdict(list)
prev_end = None
for r in rec:
    kid, ns, xcc = int(r[0]), int(r[1]) & 0xff, (int(r[1]) >> 8) & 0xf
    xccs[kid].append(xcc)
    t = r[2:2 + ns] * 10.0  # ns
    gap = (t[0] - prev_end) if prev_end is not None else 0.0
    stats[kid].append((gap, *(t[i + 1] - t[i] for i in range(ns - 1))))
    prev_end = t[-1]
for kid, v in sorted(stats.items()):
    import numpy as np
    a = np.array(v)
    print(f"kernel {kid:4d} n={len(v):4d}  gap_from_prev_stamp_end {np.median(a[:,0]):8.0f} ns | phases: "
          + " ".join(f"{np.median(a[:, i]):7.0f}" for i in range(1, a.shape[1])))
for kid, v in sorted(xccs.items()):
    print("kernel", kid, "xcc of block 0 over launches:", v[:48])

# raw time line around the last block's head launches (kernel ids 5000 / 5001 = first / last workgroup)
idx = [i for i, r in enumerate(rec) if int(r[0]) in (5000, 5001)]
if idx:
    lo, hi = max(idx[-6] - 3, 0), min(idx[-1] + 4, len(rec))
    base = rec[lo][2]
    for r in rec[lo:hi]:
        ns = int(r[1]) & 0xff
        t = (r[2:2 + ns] - base) * 10.0
        print(f"kid {int(r[0]):5d} xcc {(int(r[1]) >> 8) & 0xf}  stamps (ns from first shown): " + " ".join(f"{x:8.0f}" for x in t))
