"""Diagnostic: the 240-row products of the actor / critic updates, backend against the library's kernel."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N
dev = "cuda:0"
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for rows, kin, nout in ((240, 1536, 512), (240, 512, 512), (240, 512, 255), (240, 512, 6)):
    x = torch.randn(rows, kin, device=dev); w = torch.randn(nout, kin, device=dev) / kin ** 0.5; dy = torch.randn(rows, nout, device=dev)
    print(f"rows {rows} in {kin} out {nout}: forward torch {t(lambda: x @ w.t()):6.1f} native {t(lambda: N.gemm(x, w, False, False)) if N.gemm_ok(x, w, False, False) else -1:6.1f} | "
          f"dgrad torch {t(lambda: dy @ w):6.1f} native {t(lambda: N.gemm(dy, w, False, True)) if N.gemm_ok(dy, w, False, True) else -1:6.1f} | "
          f"wgrad torch {t(lambda: dy.t() @ x):6.1f} native {t(lambda: N.gemm(dy, x, True, True)) if N.gemm_ok(dy, x, True, True) else -1:6.1f}", flush=True)
