"""Diagnostic: per-parameter gradient error of the GPU world-model loss vs the host eager path."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DIMS, build_models, observe_inputs, rel_err
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
d, salt, B = DIMS[tag], 21 if tag == "default" else 11, 2 if tag == "default" else 4
wm_c, _, _ = build_models(d, salt); wm_g, _, _ = build_models(d, salt); wm_g = wm_g.to("cuda:0")
obs, actions, rewards, continues, noise = observe_inputs(d, salt, B)
lc, ic = wm_c.loss(obs.float() / 255.0, actions, rewards, continues, noise=noise)
lg, ig = wm_g.loss(obs.cuda().float() / 255.0, actions.cuda(), rewards.cuda(), continues.cuda(), noise={k: v.cuda() for k, v in noise.items()})
print("loss", float(lc), float(lg))
for k in ("h", "post_logits", "prior_logits"):
    print(k, rel_err(ig[k].detach().cpu(), ic[k].detach()))
lc.backward(); lg.backward()
pc, pg = dict(wm_c.named_parameters()), dict(wm_g.named_parameters())
for n, p in pc.items():
    if p.grad is not None:
        print(f"{n:28s} {rel_err(pg[n].grad.cpu(), p.grad):.2e}  norm {float(p.grad.norm()):.4e}")
if "--eager-gpu" in sys.argv:
    import dreamer_amd.observe as ob
    wm_e, _, _ = build_models(d, salt); wm_e = wm_e.to("cuda:0")
    ob_scan = ob.observe_scan
    ob.observe_scan = lambda wm, *a: ob._scan_eager(wm, *a)
    import dreamer_amd.world_model as W
    le, ie = wm_e.loss(obs.cuda().float() / 255.0, actions.cuda(), rewards.cuda(), continues.cuda(), noise={k: v.cuda() for k, v in noise.items()})
    le.backward()
    pe = dict(wm_e.named_parameters())
    print("---- eager torch ops on the GPU vs CPU")
    for n, p in pc.items():
        if p.grad is not None:
            print(f"{n:28s} {rel_err(pe[n].grad.cpu(), p.grad):.2e}")
