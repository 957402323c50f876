"""Diagnostic: device time per node of a replayed graph of 200 identical small operations, by kind of operation."""
import time
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
S = torch.cuda.Stream()
torch.cuda.set_stream(S)
a = torch.randn(1 << 16, device=dev); b = torch.randn(1 << 16, device=dev)
big_a = torch.randn(1 << 22, device=dev); big_b = torch.randn(1 << 22, device=dev)
x = torch.randn(64, 32, 16, 16, device=dev); w = torch.randn(32, 32, 4, 4, device=dev) * 0.05
m1 = torch.randn(1024, 512, device=dev); m2 = torch.randn(512, 512, device=dev)
ops = {
    "elementwise add_ (256 KB)": lambda: a.add_(1.0),
    "copy_ contiguous (256 KB)": lambda: a.copy_(b),
    "copy_ contiguous (16 MB)": lambda: big_a.copy_(big_b),
    "zero_ (256 KB)": lambda: a.zero_(),
    "conv2d 64x32x16x16 k4 s2": lambda: F.conv2d(x, w, None, 2, 1),
    "mm 1024x512x512": lambda: torch.mm(m1, m2),
    "sum reduce (256 KB)": lambda: a.sum(),
    "randn-like exponential_ (256 KB)": lambda: a.exponential_(),
}
N = 200
for name, op in ops.items():
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N):
        op()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / N * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=S):
        for _ in range(N):
            op()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t) / 5 / N * 1e6
    print(f"{name:36s} eager {eager:7.2f} us per op | graph replay {rep:7.2f} us per node", flush=True)
