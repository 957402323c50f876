"""Diagnostic: the default model's mid-sized forward products inside a recorded graph (device time per launch, no host
share): the plain fp32 kernel (with its summing kernel) against the split form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N
dev = "cuda:0"
def graph_time(fn, inner=20, reps=20):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / inner * 1e3
for M, Nn, K in ((1024, 512, 512), (1024, 512, 1536), (1024, 1536, 512), (1024, 1024, 512), (1024, 512, 1024), (1024, 1024, 4096), (240, 255, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5; b = torch.randn(Nn, device=dev)
    gf = 2.0 * M * Nn * K / 1e9
    t1 = graph_time(lambda: N.gemm(x, w, False, False, bias=b))
    try:
        t2 = graph_time(lambda: N.fused_linear(x, w, b, split=True))
    except Exception as e:
        t2 = float("nan")
    print(f"{M} x {Nn} x {K} ({gf:.2f} GFLOP): plain fp32 kernel {t1:6.1f} us ({gf / t1 * 1e3:5.1f} TF) | split form {t2:6.1f} us", flush=True)
