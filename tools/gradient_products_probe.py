"""Diagnostic: the large Linear gradients -- operands read as they lie (drm_gemm_split_f32) against transposed copies + the
forward split kernel."""
import os, sys
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
for M, Nn, K, ta in ((4096, 1536, 1024, True), (1024, 4096, 1024, True), (1024, 4096, 1024, False), (4096, 5120, 1024, True), (1024, 5120, 4096, False)):
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn(K, Nn, device=dev)
    gf = 2.0 * M * Nn * K / 1e9
    t1 = t(lambda: N.gemm_split(a, b, ta))
    if ta:
        t2 = t(lambda: N.fused_linear(a.t().contiguous(), b.t().contiguous(), split=True))
        t3 = t(lambda: (a.t().contiguous(), b.t().contiguous()))
    else:
        t2 = t(lambda: N.fused_linear(a, b.t().contiguous(), split=True))
        t3 = t(lambda: b.t().contiguous())
    print(f"{'A^T B' if ta else 'A B  '} {M} x {Nn} x {K} ({gf:.1f} GFLOP): as they lie {t1:6.1f} us ({gf / t1 * 1e3:5.1f} TF) | copies + forward kernel {t2:6.1f} us (copies {t3:5.1f})", flush=True)
