"""Sanity check of the flush form of the guard allocator (GUARD_FLUSH=1): tensors of many sizes are allocated, filled,
read back and freed through torch; no access is made past any end, so the run must finish without a fault."""
import os
import sys

import torch

so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libguard_alloc.so")
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free"))
tot = 0.0
for n in (1, 3, 4, 5, 17, 255, 256, 1000, 4099, 1 << 16, (1 << 20) + 3, 5_000_001):
    a = torch.arange(n, dtype=torch.float32, device="cuda:0")
    b = (a * 2).to(torch.uint8).float() + a
    assert a.data_ptr() % 16 == 0
    tot += float(b.sum())
    x = torch.randn(n % 97 + 1, 64, device="cuda:0")
    tot += float((x @ x.t()).trace())
torch.cuda.synchronize()
print("flush sanity ok", tot)
sys.exit(0)
