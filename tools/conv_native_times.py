"""Diagnostic: the library's stride-2 4x4 convolution kernels at the training shape (1024 frames), per layer shape and pass,
beside the backend's routine for the same pass (channels-last)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dreamer_amd import _native as N

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
NI = 1024


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for cs, cb, hs in ((64, 32, 16), (128, 64, 8), (256, 128, 4)):
    small = cl(torch.randn(NI, cs, hs, hs, device=dev))
    big = cl(torch.randn(NI, cb, 2 * hs, 2 * hs, device=dev))
    kern = cl(torch.randn(cs, cb, 4, 4, device=dev) * 0.05)
    gf = 2.0 * NI * hs * hs * cs * cb * 16 / 1e9
    t_down = timed(lambda: N.conv4s2_down(big, kern))
    t_up = timed(lambda: N.conv4s2_up(small, kern))
    t_wg = timed(lambda: N.conv4s2_wgrad(small, big))
    no_backend = os.environ.get("CONV_TIMES_NO_BACKEND") == "1"      # counter passes: the library's kernels only
    b_down = 0.0 if no_backend else timed(lambda: F.conv2d(big, kern, None, 2, 1))
    b_up = 0.0 if no_backend else timed(lambda: F.conv_transpose2d(small, kern, None, 2, 1))
    print(f"cs={cs:3d} cb={cb:3d} hs={hs:2d} ({gf:.1f} GFLOP): down {t_down:6.1f} us {gf / t_down * 1e3:6.1f} TF (backend {b_down:6.1f}) | "
          f"up {t_up:6.1f} us {gf / t_up * 1e3:6.1f} TF (backend {b_up:6.1f}) | wgrad {t_wg:6.1f} us {gf / t_wg * 1e3:6.1f} TF", flush=True)
