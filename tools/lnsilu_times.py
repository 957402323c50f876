"""Times of the towers' LayerNorm + SiLU kernels at the training shapes, plain and channels-last."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N

dev = "cuda:0"


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for c, hw in ((32, 32), (64, 16), (128, 8), (256, 4)):
    n = 1024
    g = torch.randn(c, hw, hw, device=dev); b = torch.randn(c, hw, hw, device=dev); cb = torch.randn(c, device=dev)
    for cl in (False, True):
        x = torch.randn(n, c, hw, hw, device=dev)
        dy = torch.randn(n, c, hw, hw, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); dy = dy.contiguous(memory_format=torch.channels_last)
        y, mean, rstd = N.ln_silu_forward(x, g, b, 1e-5, cb)
        tf = t(lambda: N.ln_silu_forward(x, g, b, 1e-5, cb))
        tb = t(lambda: N.ln_silu_backward(x, dy, g, b, mean, rstd, cb))
        mb = x.numel() * 4 / 1e6
        if os.environ.get("LN_REFERENCE_OPS") and not cl:
            yy = torch.empty_like(x)
            print(f"    same size: copy {t(lambda: yy.copy_(x)):6.1f} us, silu {t(lambda: torch.nn.functional.silu(x)):6.1f} us, "
                  f"x * y {t(lambda: torch.mul(x, dy)):6.1f} us")
        if os.environ.get("LN_NO_BIAS"):
            tf0 = t(lambda: N.ln_silu_forward(x, g, b, 1e-5, None))
            print(f"    without the channel bias: forward {tf0:6.1f} us")
        print(f"C={c:3d} {hw:2d}x{hw:<2d} {'channels-last' if cl else 'plain        '} forward {tf:6.1f} us ({2 * mb / tf:5.2f} TB/s)  "
              f"backward {tb:6.1f} us ({5 * mb / tb:5.2f} TB/s over 5 tensor passes)")
