"""Diagnostic: a few launches of the split GEMM (and the fp32 MFMA form) at two shapes, for counter passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N
dev = "cuda:0"
for (M, Nn, K) in [(61440, 1536, 512), (4096, 512, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5; b = torch.randn(Nn, device=dev)
    for _ in range(10):
        N.fused_linear(x, w, b, split=True)
    for _ in range(10):
        N.fused_linear(x, w, b)
torch.cuda.synchronize()
