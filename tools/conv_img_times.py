"""Times of the image-side tower kernels at the training shape (1024 frames) beside the backend's routines."""
import torch
import torch.nn.functional as F

from dreamer_amd import _native as N

torch.backends.cudnn.benchmark = True
dev = "cuda:0"
n, cs, cimg, h, w = 1024, 32, 3, 32, 32
big = torch.randn(n, cimg, 2 * h, 2 * w, device=dev)
small = torch.randn(n, cs, h, w, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(cs, cimg, 4, 4, device=dev) * 0.1
bias = torch.randn(cimg, device=dev)
big_cl = big.contiguous(memory_format=torch.channels_last)


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


rows = [
    ("down  (first encoder layer forward)", lambda: N.conv_img_down(big, wt), lambda: F.conv2d(big_cl, wt, None, 2, 1)),
    ("up    (last decoder layer forward)", lambda: N.conv_img_up(small, wt, bias), lambda: F.conv_transpose2d(small, wt, bias, 2, 1)),
    ("wgrad (either layer)", lambda: N.conv_img_wgrad(small, big),
     lambda: torch.ops.aten.convolution_backward(small, big_cl, wt, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])),
]
for name, mine, theirs in rows:
    print(f"{name:40s} library {t(mine):8.1f} us   backend {t(theirs):8.1f} us")
