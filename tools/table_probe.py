"""Diagnostic: first head layer over cat[h, z] for 61440 states: one-hot table rows in the epilogue against a dense
product over the 1024 one-hot columns."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for M in (61440, 4096):
    D, L, K, Nn = 512, 32, 32, 512
    x = torch.randn(M, D, device=dev); w = torch.randn(Nn, D + L * K, device=dev) * 0.03; b = torch.randn(Nn, device=dev)
    idx = torch.randint(0, K, (M, L), device=dev, dtype=torch.uint8)
    table = w[:, D:].t().contiguous()
    wh = w[:, :D]
    z = torch.nn.functional.one_hot(idx.long(), K).float().reshape(M, L * K)
    xz = torch.cat([x, z], 1)
    for split in (False, True):
        t0 = timeit(lambda: N.fused_linear(x, wh, b, split=split))
        t1 = timeit(lambda: N.fused_linear(x, wh, b, table=table, zidx=idx, split=split))
        t2 = timeit(lambda: N.fused_linear(xz, w, b, split=split))
        print(f"M={M} split={split}: h part only {t0*1e6:7.1f} us | + table rows {t1*1e6:7.1f} us | dense K=1536 {t2*1e6:7.1f} us")
