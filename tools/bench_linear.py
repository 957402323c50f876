"""Diagnostic: time the fused linear block against torch.matmul for a few shapes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd import _native as N
dev = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (M, Nn, K) in [(4096, 512, 512), (4096, 1536, 512), (4096, 2048, 512), (61440, 512, 512), (61440, 1536, 512), (16384, 512, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5; b = torch.randn(Nn, device=dev)
    t1 = timeit(lambda: N.fused_linear(x, w, b))
    t2 = timeit(lambda: torch.addmm(b, x, w.t()))
    t3 = timeit(lambda: N.fused_linear(x, w, b, split=True))
    fl = 2.0 * M * Nn * K
    ref = (x.double() @ w.double().t() + b.double())
    e1 = float((N.fused_linear(x, w, b).double() - ref).abs().max() / ref.abs().max())
    e3 = float((N.fused_linear(x, w, b, split=True).double() - ref).abs().max() / ref.abs().max())
    print(f"    split (3 x bf16, 6 products): {t3*1e6:8.1f} us {fl/t3/1e12:6.1f} TF | error vs fp64: fp32 MFMA {e1:.2e}, split {e3:.2e}")
    err = float((N.fused_linear(x, w, b) - torch.addmm(b, x, w.t())).abs().max())
    print(f"M={M:6d} N={Nn:5d} K={K}: ours {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF | torch {t2*1e6:8.1f} us {fl/t2/1e12:6.1f} TF | maxdiff {err:.2e}")
