"""Diagnostic: shapes of the products that one training update sends to the library's plain fp32 product kernel
(N.gemm), with the time of each shape measured alone."""
import os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd import _native as N
from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner

seen = collections.Counter()
orig = N.gemm


def spy(a, b, ta, tb, bias=None, **kw):
    seen[(tuple(a.shape), tuple(b.shape), ta, tb, bias is not None)] += 1
    return orig(a, b, ta, tb, bias=bias, **kw)


N.gemm = spy
dev = torch.device("cuda:0")
cfg = Config(device=dev)
L = Learner(cfg, (3, 64, 64), 6)
cap = 2048
rng = np.random.default_rng(0)
buf = ReplayBuffer(cap, (3, 64, 64), 64, dev)
buf.store_many(rng.integers(0, 256, size=(cap, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=cap),
               rng.standard_normal(cap).astype(np.float32), rng.random(cap) > 0.01)
L.make_stream_current()
L.update(buf.sample(16, want_f32=True))
torch.cuda.synchronize()
first = dict(seen)
N.gemm = orig


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tot = 0.0
for (sa, sb, ta, tb, hb), n in sorted(first.items(), key=lambda kv: -kv[1]):
    a = torch.randn(*sa, device=dev); b = torch.randn(*sb, device=dev)
    m = sa[1] if ta else sa[0]; k = sa[0] if ta else sa[1]; nn = sb[0] if not tb else sb[1]
    if not ta and not tb:
        nn = sb[0]
    us = t(lambda: orig(a, b, ta, tb))
    gf = 2.0 * m * nn * k / 1e9
    tot += us * n
    print(f"{n:3d} x  a{sa} b{sb} ta={int(ta)} tb={int(tb)} bias={int(hb)}  {us:7.1f} us  {gf / us * 1e3:6.1f} TF   ({us * n:7.1f} us in the update)")
print(f"total {tot / 1e3:.3f} ms over the calls of the first update (recorded or eager)")
