"""Per-layer forward / backward times of the tower convolutions at the training shape (1024 frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
N = 1024
CL = len(sys.argv) > 1 and sys.argv[1] == "channels_last"   # NHWC memory format for inputs, weights and gradients


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


layers = [("enc1", False, 3, 32, 64), ("enc2", False, 32, 64, 32), ("enc3", False, 64, 128, 16), ("enc4", False, 128, 256, 8),
          ("dec1", True, 256, 128, 4), ("dec2", True, 128, 64, 8), ("dec3", True, 64, 32, 16), ("dec4", True, 32, 3, 32)]
for name, tr, cin, cout, hin in layers:
    x = torch.randn(N, cin, hin, hin, device=dev, requires_grad=True)
    w = torch.randn((cin, cout, 4, 4) if tr else (cout, cin, 4, 4), device=dev, requires_grad=True) * 0.05
    w = w.detach().requires_grad_(True)
    if CL:
        x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = w.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    f = (lambda: F.conv_transpose2d(x, w, None, 2, 1)) if tr else (lambda: F.conv2d(x, w, None, 2, 1))
    y = f()
    g = torch.randn_like(y)
    if CL:
        g = g.contiguous(memory_format=torch.channels_last)
    t_f = timed(f)
    t_fb = timed(lambda: torch.autograd.grad(f(), (x, w), g))
    t_fw = timed(lambda: torch.autograd.grad(f(), (w,), g))
    out_mb = y.numel() * 4 / 1e6
    in_mb = x.numel() * 4 / 1e6
    gf = 2.0 * N * (y.shape[2] ** 2 if not tr else hin ** 2) * cin * cout * 16 / 1e9
    print(f"{name}: fwd {t_f:7.1f} us  fwd+bwd(x,w) {t_fb:7.1f} us  fwd+bwd(w) {t_fw:7.1f} us   in {in_mb:6.1f} MB out {out_mb:6.1f} MB  {gf:6.1f} GFLOP")
