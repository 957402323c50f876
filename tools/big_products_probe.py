"""Diagnostic: the scaled model's large update products on the fp32 kernel, the split form and the bf16 form."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N
dev = "cuda:0"
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for M, Nn, K in ((12288, 4096, 1024), (1024, 4096, 5120), (1024, 5120, 4096), (4096, 5120, 1024), (512, 5120, 1024)):
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5
    gf = 2.0 * M * Nn * K / 1e9
    a = t(lambda: N.gemm(x, w, False, False)); b = t(lambda: N.fused_linear(x, w, split=True)); c = t(lambda: N.fused_linear(x, w, bf16=True))
    ref = x.double() @ w.double().t()
    e = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"{M} x {Nn} x {K} ({gf:.0f} GFLOP): fp32 {a:7.1f} us {gf / a * 1e3:6.1f} TF err {e(N.gemm(x, w, False, False)):.1e} | split {b:7.1f} us {gf / b * 1e3:6.1f} TF err "
          f"{e(N.fused_linear(x, w, split=True)):.1e} | bf16 {c:7.1f} us {gf / c * 1e3:6.1f} TF err {e(N.fused_linear(x, w, bf16=True)):.1e}", flush=True)
