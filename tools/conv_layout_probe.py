"""Times the conv towers (encoder + decoder, forward + backward, 1024 frames) in NCHW and channels-last
layouts with torch's own LayerNorm/SiLU, to see which layout MIOpen serves faster."""
import os, sys, time
os.environ["DRM_NO_LN_FUSION"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreamer_amd.world_model import WorldModel

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
torch.manual_seed(0)
wm = WorldModel((3, 64, 64), 6).to(dev)
x = torch.rand(1024, 3, 64, 64, device=dev)
h = torch.randn(1024, 512, device=dev)
z = torch.nn.functional.one_hot(torch.randint(0, 32, (1024, 32), device=dev), 32).float()


def run(cl):
    enc, dec = wm.enc.conv, wm.dec
    if cl:
        enc.to(memory_format=torch.channels_last); dec.to(memory_format=torch.channels_last)
    xin = x.contiguous(memory_format=torch.channels_last) if cl else x
    def once():
        f = enc(xin)
        y = dec(h, z)
        (f.sum() + y.sum()).backward()
    for _ in range(4):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3

print("NCHW ms", run(False))
print("channels_last ms", run(True))
