"""Diagnostic: few-row products over a long inner dimension (the memory-bound products of wide models): time and the
rate at which the weights are streamed, fp32 and bf16 matrix inputs, against torch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd import _native as N
dev = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (M, Nn, K) in [(16, 12288, 4096), (16, 13312, 4096), (16, 5120, 12288), (16, 1024, 4096), (32, 12288, 4096), (64, 12288, 4096), (6, 12288, 4096)]:
    # eight weight sets used in turn, so that nothing is served from the 256 MB cache
    ws = [torch.randn(Nn, K, device=dev) / K ** 0.5 for _ in range(8)]
    x = torch.randn(M, K, device=dev); b = torch.randn(Nn, device=dev)
    i = [0]
    def ours():
        i[0] = (i[0] + 1) % 8
        return N.fused_linear(x, ws[i[0]], b)
    def ref():
        i[0] = (i[0] + 1) % 8
        return torch.addmm(b, x, ws[i[0]].t())
    t1, t2 = timeit(ours), timeit(ref)
    by = 4.0 * Nn * K
    err = float((N.fused_linear(x, ws[0], b) - torch.addmm(b, x, ws[0].t())).abs().max())
    print(f"M={M:3d} N={Nn:6d} K={K:6d}: ours {t1*1e6:7.1f} us {by/t1/1e12:5.2f} TB/s | torch {t2*1e6:7.1f} us {by/t2/1e12:5.2f} TB/s | maxdiff {err:.1e}", flush=True)
