"""Diagnostic: the tower convolutions with bf16 inputs (precision 1) at the training shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreamer_amd import _native as N
dev = torch.device("cuda:0"); NI = 1024
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
cl = lambda x: x.contiguous(memory_format=torch.channels_last)
for cs, cb, hs in ((64, 32, 16), (128, 64, 8), (256, 128, 4)):
    small = cl(torch.randn(NI, cs, hs, hs, device=dev)); big = cl(torch.randn(NI, cb, 2 * hs, 2 * hs, device=dev))
    kern = cl(torch.randn(cs, cb, 4, 4, device=dev) * 0.05)
    gf = 2.0 * NI * hs * hs * cs * cb * 16 / 1e9
    d = t(lambda: N.conv4s2_down(big, kern, bf16=True)); u = t(lambda: N.conv4s2_up(small, kern, bf16=True)); w = t(lambda: N.conv4s2_wgrad(small, big, bf16=True))
    print(f"cs={cs:3d} cb={cb:3d} hs={hs:2d} ({gf:.1f} GFLOP), bf16 inputs: down {d:6.1f} us {gf / d * 1e3:6.1f} TF | up {u:6.1f} us {gf / u * 1e3:6.1f} TF | wgrad {w:6.1f} us {gf / w * 1e3:6.1f} TF", flush=True)
