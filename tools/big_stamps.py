"""Diagnostic: phase timing inside the large-row fp32 product (needs the DRM_STAMPS library: make stamps)."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DRM_LIB"] = os.path.join(ROOT, "dreamer_amd", "libdreamer_amd_stamps.so")
sys.path.insert(0, ROOT)
from dreamer_amd import _native as N
dev = "cuda:0"
buf = torch.zeros(1 + 8 * 100000, dtype=torch.int64, device=dev)
fn = N.lib().drm_debug_set_stamp_buffer
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
for (M, Nn, K) in [(4096, 512, 512), (61440, 1536, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) / K ** 0.5; b = torch.randn(Nn, device=dev)
    for _ in range(5):
        N.fused_linear(x, w, b)
    torch.cuda.synchronize()
    buf.zero_()
    assert fn(buf.data_ptr()) == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        N.fused_linear(x, w, b)
    ev[1].record()
    torch.cuda.synchronize()
    assert fn(0) == 0
    n = int(buf[0].item())
    rec = buf[1:1 + 8 * n].view(n, 8).cpu().numpy()
    rec = rec[rec[:, 2].argsort()]
    print(f"M={M} N={Nn} K={K}: {ev[0].elapsed_time(ev[1]) * 100:.1f} us per launch; {n} records")
    per = n // 10
    last = rec[-per:]
    for r in last[:: max(1, per // 24)]:
        ns = int(r[1]) & 0xff
        hw = (int(r[1]) >> 16) & 0xffffffff
        t = r[2:2 + ns] * 10.0
        print(f"   block y {int(r[1]) >> 48:4d} xcc {(int(r[1]) >> 8) & 0xf} wave slot {hw & 0xf} simd {(hw >> 4) & 3} cu {(hw >> 8) & 0xf} se {(hw >> 13) & 7} "
              f"start {t[0] - last[0][2] * 10.0:9.0f} ns | phases (ns): " + " ".join(f"{t[i + 1] - t[i]:7.0f}" for i in range(ns - 1)))
