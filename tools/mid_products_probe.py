"""Diagnostic: the default model's mid-sized update products (1024 rows) on the plain fp32 kernel against the split form."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N
dev = "cuda:0"
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
# (rows, out, k): the product as x (rows, k) @ w (out, k).T
for M, Nn, K in ((1024, 1024, 4096), (1024, 1536, 4096), (1024, 512, 1536), (1024, 1536, 512), (512, 1536, 1024), (1024, 512, 512), (1024, 512, 1024), (1024, 1024, 512), (512, 512, 1024)):
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5
    gf = 2.0 * M * Nn * K / 1e9
    a = t(lambda: N.gemm(x, w, False, False)); b = t(lambda: N.fused_linear(x, w, split=True))
    ref = x.double() @ w.double().t()
    e = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"{M} x {Nn} x {K} ({gf:.1f} GFLOP): fp32 {a:7.1f} us {gf / a * 1e3:6.1f} TF err {e(N.gemm(x, w, False, False)):.1e} | split {b:7.1f} us {gf / b * 1e3:6.1f} TF err "
          f"{e(N.fused_linear(x, w, split=True)):.1e}", flush=True)
