"""Diagnostic: the imagination block at the default widths over a sweep of trajectory counts, fp32 MFMA products against
the split form (three bf16 planes per operand, six partial products on the bf16 pipe)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamer_amd.actor import Actor
from dreamer_amd.critic import Critic
from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.world_model import WorldModel, WorldModelState

dev = torch.device("cuda:0")
D, L, K, M, A = 512, 32, 32, 512, 6
torch.manual_seed(0)
wm = WorldModel((3, 64, 64), A, num_latents=L, classes_per_latent=K, hidden_size=D, mlp_hidden_units=M).to(dev)
actor = Actor(D + L * K, A, M, 2).to(dev)
critic = Critic(D + L * K, M, 2).to(dev)
eng = ImagineEngine(wm, actor, critic)
g = torch.Generator().manual_seed(1)
for B in (256, 512, 1024, 2048, 4096):
    h0 = torch.randn(B, D, generator=g).tanh().to(dev)
    z0 = torch.nn.functional.one_hot(torch.randint(0, K, (B, L), generator=g), K).float().to(dev)
    st = WorldModelState(h=h0, z=z0)
    line = f"B={B:5d}:"
    for prec in ("f32", "f32_split", "bf16"):
        fn = lambda: eng.run(st, 15, 0.997, 0.95, repack=False, persistent=True, precision=prec)
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 20
        for _ in range(n): fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
        line += f"  {prec} {ms:7.3f} ms"
    print(line, flush=True)
