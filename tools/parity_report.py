"""Print the relative errors of the native imagination block against the golden vectors (GPU)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DIMS, build_models, expo, load, rel_err, start_state
from dreamer_amd.imagine import ImagineEngine
from dreamer_amd.world_model import WorldModelState
DEV = "cuda:0"
for tag in ("small", "default"):
    g = load(f"imagine_{tag}.npz"); d, salt = DIMS[tag], int(g["salt"])
    wm, actor, critic = build_models(d, salt)
    eng = ImagineEngine(wm.to(DEV), actor.to(DEV), critic.to(DEV))
    h0, z0, _ = start_state(d, salt)
    noise = {"a": torch.from_numpy(g["noise_a"]).to(DEV), "z": expo("imagine_z", (d["H"], d["B"], d["L"], d["K"]), salt).to(DEV)}
    out = eng.run(WorldModelState(h=h0.to(DEV), z=z0.to(DEV)), d["H"], float(g["gamma"]), float(g["lam"]), noise=noise)
    torch.cuda.synchronize()
    same = np.array_equal(out["actions"].cpu().numpy(), g["actions"]) and np.array_equal(out["zidx"].cpu().numpy(), g["zidx"])
    print(tag, "indices equal:", same, {k: f"{rel_err(out[k].cpu(), g[k]):.2e}" for k in ("h", "rewards", "conts", "values", "lam_returns", "actor_logp")})
