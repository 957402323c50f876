#!/usr/bin/env python3
"""Benchmark of the Dreamer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload imagine16|imagine4096]

One "step" is one imagination block (lib/trainer.py:176-224 of the reference: actor -> action
draw -> prior RSSM step, H = 15 times; reward / continue heads; critic value; lambda-return) for
B start states, including the generation of its Exp(1) noise.  Weights are fixed (random init of
the default architecture, heads made non-zero), start states synthetic.  Default workload:
the reference's default configuration (B = 16 trajectories per GPU).  With N > 1 every rank runs
its own B trajectories (trajectories are independent: no collective on the data path) and the
reported value is the whole-job rate (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec
MFMA_F32_TFLOPS = 157.3   # dense fp32 matrix peak
MFMA_BF16_TFLOPS = 2500.0 # dense bf16 matrix peak
# algorithmic work per imagined latent step (default dims, A = 6), dense form:
# actor 2,103,296 + seq.linear 1,114,112 + GRU 3,145,728 + dyn 2,621,440 + rew/cont 2 x 2,622,464
# + critic value 2,358,272
FLOP_PER_LATENT_STEP = 16_587_776


def build_models(device, actions=6, seed=0):
    from dreamer_amd.actor import Actor
    from dreamer_amd.config import Config
    from dreamer_amd.critic import Critic
    from dreamer_amd.world_model import WorldModel
    cfg = Config()
    torch.manual_seed(seed)
    wm = WorldModel((3, 64, 64), actions, cfg.num_latents, cfg.classes_per_latent, cfg.hidden_size,
                    cfg.base_cnn_channels, cfg.mlp_hidden_units, cfg.free_bits, cfg.beta_pred, cfg.beta_dyn,
                    cfg.beta_rep, cfg.unimix_eps)
    S = cfg.hidden_size + cfg.num_latents * cfg.classes_per_latent
    actor = Actor(S, actions, cfg.mlp_hidden_units, 2, cfg.actor_entropy_scale, cfg.actor_ret_norm_limit,
                  cfg.actor_ret_norm_decay, cfg.unimix_eps)
    critic = Critic(S, cfg.mlp_hidden_units, 2, cfg.critic_num_buckets, cfg.critic_bucket_min, cfg.critic_bucket_max,
                    cfg.critic_ema_decay, cfg.critic_ema_regularizer)
    with torch.no_grad():  # zero-initialised heads would make R = 0, C = 0.5 and a uniform policy
        for head in (wm.rew.head, wm.cont.head, actor.head, critic.head):
            torch.nn.init.normal_(head.weight, std=0.05)
            torch.nn.init.normal_(head.bias, std=0.05)
    return cfg, wm.to(device), actor.to(device), critic.to(device)


def start_states(B, cfg, device, seed=1):
    g = torch.Generator().manual_seed(seed)
    h0 = torch.tanh(torch.randn(B, cfg.hidden_size, generator=g))
    idx = torch.randint(0, cfg.classes_per_latent, (B, cfg.num_latents), generator=g)
    z0 = torch.nn.functional.one_hot(idx, cfg.classes_per_latent).float()
    return h0.to(device), z0.to(device)


# kernels (rocprofv3 names) behind each timing class of the library
CLASS_KERNELS = {"linear": ("k_skinny", "k_linear16<1, 0", "k_linear16m"), "k_linear_big": ("k_linear_big",)}
MFMA_CLASSES = ("k_linear_big",)   # classes whose launches take the precision of the block


def pmc_traffic(workload, cls):
    """HBM-side bytes per launch of a kernel class from the committed rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 16-byte-per-lane reads at half their
    size on gfx950 and is doubled), and the file they were read from.  (None, None) when no pass of this
    workload is on file."""
    for rel in ("profiles/r3_pmc.json", "profiles/r2_pmc.json", "profiles/r1_pmc.json"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            break
    else:
        return None, None
    if cls not in CLASS_KERNELS:
        return None, None
    try:
        table = json.load(open(path)).get(workload, {})
    except (OSError, ValueError):
        return None, None
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    n = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    for kernel, ctrs in table.items():
        if not any(kernel.startswith("drm::" + k) or ("drm::" + k) in kernel for k in CLASS_KERNELS[cls]):
            continue
        for c in tot:
            if c in ctrs:
                tot[c] += ctrs[c]["total"]
                n[c] += ctrs[c]["dispatches"]
    if n["FETCH_SIZE"] == 0 or n["WRITE_SIZE"] == 0:
        return None, None
    return (1024.0 * (2.0 * tot["FETCH_SIZE"] / n["FETCH_SIZE"] + tot["WRITE_SIZE"] / n["WRITE_SIZE"]),
            rel + " (committed rocprofv3 --pmc passes of this workload, not collected in this run)")


def cpu_baseline(cfg, wm, actor, critic, B, H, budget_s=15.0):
    """The oracle (CPU restatement of the reference, decoder call included as the reference makes it)
    timed on the host cores for the same B x H block."""
    import oracle as O
    ww = {k: v.detach().cpu() for k, v in wm.state_dict().items()}
    wa = {k: v.detach().cpu() for k, v in actor.state_dict().items()}
    wc = {k: v.detach().cpu() for k, v in critic.state_dict().items()}
    h0, z0 = start_states(B, cfg, "cpu")
    A, L, K = wm.action_size, cfg.num_latents, cfg.classes_per_latent

    def once():
        ea = torch.empty(H, B, A).exponential_()
        ez = torch.empty(H, B, L, K).exponential_()
        with torch.no_grad():
            O.imagine_block(ww, wa, wc, h0, z0, ea, ez, cfg.gamma, cfg.lam, cfg.unimix_eps, run_decoder=True)

    once()
    t0 = time.perf_counter()
    once()
    per = time.perf_counter() - t0
    reps = max(3, min(200, int(budget_s / max(per, 1e-3))))
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        once()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B * H / med, "unit": "imagined latent steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} repetitions of the B={B} x H={H} imagination block (median), decoder call included as in the "
                      f"reference; {os.cpu_count()} logical CPUs visible"}


def measure_train(device, world, rank, steps, warmup, fence):
    """Full training updates (lib/trainer.py:162-247): replay gather, world-model loss + backward,
    gradient all-reduce, clip, Adam, imagination block, critic and actor updates; per-GPU batch
    16 x 64 from a synthetic HBM-resident replay ring."""
    import numpy as np
    from dreamer_amd.config import Config
    from dreamer_amd.replay_buffer import ReplayBuffer
    from dreamer_amd.trainer import Learner
    cfg = Config(device=device)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True   # as the reference's main.py:11 sets it (MIOpen picks its conv kernels by search)
    learner = Learner(cfg, (3, 64, 64), 6)
    learner.make_stream_current()      # as dreamer_amd.trainer.train does: sampling, updates and acting on one stream
    with torch.no_grad():
        for head in (learner.world_model.rew.head, learner.world_model.cont.head, learner.actor.head, learner.critic.head):
            torch.nn.init.normal_(head.weight, std=0.05)
    cap = 8192
    rng = np.random.default_rng(1234 + rank)
    buf = ReplayBuffer(cap, (3, 64, 64), cfg.batch_length, device)
    buf.store_many(rng.integers(0, 256, size=(cap, 3, 64, 64), dtype=np.uint8), rng.integers(0, 6, size=cap),
                   rng.standard_normal(cap).astype(np.float32), rng.random(cap) > 0.01)
    np.random.seed(42 + rank)

    def step():
        return learner.update(buf.sample(cfg.batch_size, want_f32=True))

    for i in range(warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            progress("first training update done")
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    host_dt = time.perf_counter() - t0     # launches queued; the GPU may still be working
    fence()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out["world_model_loss"]).all()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ups = steps / dt
    collective = None
    if world > 1:
        # the gradient all-reduce on its own: the world model's flat buffer, ten calls, device-timed on this rank
        flat = learner.g_wm.flat
        from dreamer_amd import dist as DD
        for _ in range(3):
            DD.all_reduce_sum(flat)
        fence()
        t1 = time.perf_counter()
        for _ in range(10):
            DD.all_reduce_sum(flat)
        fence()
        ar = (time.perf_counter() - t1) / 10
        nbytes = flat.numel() * 4
        collective = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""), "ranks": dist.get_world_size(),
                      "world_model_allreduce_bytes": nbytes, "world_model_allreduce_ms": ar * 1e3,
                      "algbw_GB/s": nbytes / ar / 1e9, "busbw_GB/s": nbytes / ar / 1e9 * 2 * (world - 1) / world}
        progress(f"all-reduce of {nbytes / 1e6:.1f} MB over {world} ranks ({collective['backend']}): {ar * 1e3:.3f} ms")
    acting = None if os.environ.get("DRM_BENCH_NO_ACTING", "0") == "1" else measure_acting(learner, device, rng)
    return {"acting": acting, "collective": collective, "host_ms_per_update": host_dt / steps * 1e3, "updates_per_sec": ups, "ms_per_update": dt / steps * 1e3, "n_gpus": world, "per_gpu_batch": "16 x 64",
            "replayed_steps_per_sec": ups * world * cfg.batch_size * cfg.batch_length, "steps": steps, "warmup": warmup,
            "grad_allreduce_bytes_per_update": 4 * sum(g.flat.numel() for g in (learner.g_wm, learner.g_actor, learner.g_critic))}


def measure_acting(learner, device, rng, steps=300):
    """Policy latency per environment step (lib/trainer.py:126-151): one 12 KB frame host -> device, the
    B = 1 posterior step, the actor, and the action back on the host.  `policy` is what `train` runs
    (ActingPolicy: the step replayed from a recorded graph); `act_step` the same arithmetic launched
    eagerly; `full_step` evaluates the whole `WorldModel.step` (prior, reward, continue and decoder heads
    too) as the reference's loop does."""
    from dreamer_amd.trainer import ActingPolicy
    wm, actor = learner.world_model, learner.actor
    frames = rng.integers(0, 256, size=(8, 3, 64, 64), dtype="uint8")
    out = {}
    policy = ActingPolicy(wm, actor, device, (3, 64, 64))
    for i in range(steps + 20):
        if i == 20:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        policy.feedback(policy.act(frames[i % 8]), True)
    torch.cuda.synchronize()
    out["ms_per_env_step_policy"] = (time.perf_counter() - t0) / steps * 1e3
    out["policy_graph"] = policy.graph is not None
    for name in ("act_step", "full_step"):
        state = wm.init_state(1, device)
        last_action = torch.zeros(1, dtype=torch.long, device=device)
        last_cont = torch.ones(1, dtype=torch.float32, device=device)
        for i in range(steps + 20):
            if i == 20:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            with torch.no_grad():
                obs_t = torch.as_tensor(frames[i % 8], dtype=torch.float32, device=device).unsqueeze(0) / 255.0
                if name == "act_step":
                    state = wm.act_step(state, last_action, obs_t, last_cont)
                else:
                    state, _ = wm.step(state, last_action, obs_t, last_cont)
                action = actor(state).sample().item()
            last_action.fill_(action)
        torch.cuda.synchronize()
        out["ms_per_env_step_" + name] = (time.perf_counter() - t0) / steps * 1e3
    return out


def measure_scaled_model(device, steps=4):
    """The scaled model on one GPU: D = 4096, 32 x 32 categorical latent, 64 x 64 frames; random weights and inputs."""
    from dreamer_amd.actor import Actor
    from dreamer_amd.critic import Critic
    from dreamer_amd.imagine import ImagineEngine
    from dreamer_amd.world_model import WorldModel, WorldModelState
    D, L, K, M, A, B, T, H = 4096, 32, 32, 512, 6, 16, 64, 15
    torch.manual_seed(0)
    wm = WorldModel((3, 64, 64), A, num_latents=L, classes_per_latent=K, hidden_size=D, mlp_hidden_units=M).to(device).train()
    actor = Actor(D + L * K, A, M, 2).to(device)
    critic = Critic(D + L * K, M, 2).to(device)
    g = torch.Generator().manual_seed(1)
    obs = torch.rand(B, T, 3, 64, 64, generator=g).to(device)
    actions = torch.randint(0, A, (B, T), generator=g).to(device)
    rewards = torch.randn(B, T, generator=g).to(device)
    continues = (torch.rand(B, T, generator=g) > 0.02).to(device)

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps * 1e3

    def update():
        wm.zero_grad(set_to_none=True)
        loss, _ = wm.loss(obs, actions, rewards, continues)
        loss.backward()

    out = {"config": f"D={D}, {L}x{K} latent, mlp {M}, A={A}; world-model loss + backward over {B} x {T} frames; "
                     f"imagination block H={H}"}
    for prec in ("f32", "bf16"):
        wm.observe_precision = prec
        out[f"wm_loss_backward_ms_{prec}"] = timeit(update)
    eng = ImagineEngine(wm, actor, critic)
    for Bi in (16, 1024):
        h0 = torch.randn(Bi, D, generator=g).tanh().to(device)
        z0 = torch.nn.functional.one_hot(torch.randint(0, K, (Bi, L), generator=g), K).float().to(device)
        st = WorldModelState(h=h0, z=z0)
        for prec in ("f32", "bf16"):
            ms = timeit(lambda: eng.run(st, H, 0.997, 0.95, repack=False, persistent=True, precision=prec))
            out[f"imagine_B{Bi}_{prec}_ms_per_block"] = ms
            out[f"imagine_B{Bi}_{prec}_steps_per_s"] = Bi * H / ms * 1e3
    return out


def progress(msg):
    """One line on stderr per phase (stdout carries only the JSON line)."""
    print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="imagine16", choices=["imagine16", "imagine4096", "imagine4096_bf16", "imagine4096_split"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-kernels", action="store_true",
                    help="skip the stand-alone bandwidth figures of the sampling / lambda-return / gather kernels")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the large-batch (B = 4096) and the rebuilt-tables measurements beside the default workload")
    ap.add_argument("--train-steps", type=int, default=20, help="full training updates to time as well (0 = skip)")
    ap.add_argument("--train-limit-s", type=float, default=420.0, help="give up on the training updates after this long")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hook for a one-GPU box: every rank on device 0, collectives over gloo (exercises the
    # multi-process control flow; the measured numbers mean nothing then)
    rehearsal = os.environ.get("DRM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from dreamer_amd import _native as N
    from dreamer_amd.imagine import ImagineEngine
    from dreamer_amd.world_model import WorldModelState

    N.check(N.lib().drm_arch_ok(), "drm_arch_ok")
    cfg, wm, actor, critic = build_models(device)
    H = cfg.imagination_horizon
    eng = ImagineEngine(wm, actor, critic)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def measure_imagine(workload, steps, warmup, profile_steps, repack=False):
        """Times `steps` imagination blocks of one workload (after `warmup` untimed ones, barrier + device
        synchronisation on both sides, maximum over ranks) and, on rank 0, the per-kernel-class launch durations of
        a short separate pass with HIP events on the launch stream.  `repack`: the one-hot gather tables are rebuilt
        from the weights before every block, as a training update must after its optimiser step."""
        B = 16 if workload == "imagine16" else 4096
        precision = "bf16" if workload.endswith("_bf16") else "f32_split" if workload.endswith("_split") else "f32"
        h0, z0 = start_states(B, cfg, device, seed=1 + rank)
        start = WorldModelState(h=h0, z=z0)
        zidx0 = N.onehot_to_index(z0)

        def step():
            # runs the block including its Exp(1) noise: one generator launch for both tensors at few trajectories, drawn
            # inside the kernels (counter-based generator) from 256 trajectories on in the fp32 / split forms
            # "f32" here names the fp32 matrix-instruction form explicitly (the engine's own default at 256 rows and more
            # is the split form, measured as "f32_split")
            return eng.run(start, H, cfg.gamma, cfg.lam, start_zidx=zidx0, repack=repack, persistent=True,
                           precision="f32_mfma" if precision == "f32" else precision)

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        fence()
        elapsed = time.perf_counter() - t0
        assert torch.isfinite(out["lam_returns"]).all()
        progress(f"{workload}{' (tables rebuilt per block)' if repack else ''}: {elapsed / steps * 1e3:.3f} ms per block on this rank")
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        res = {"B": B, "precision": precision, "elapsed": elapsed, "steps": steps, "warmup": warmup,
               "ms_per_step": elapsed / steps * 1e3, "value": world * B * H * steps / elapsed, "roofline": None}
        if rank == 0 and profile_steps > 0:
            res["roofline"] = kernel_roofline(step, workload, precision, profile_steps, res["ms_per_step"])
        return res

    def kernel_roofline(step, workload, precision, profile_steps, step_ms):
        # The profiled pass launches kernel by kernel with an event after each.  The host needs longer
        # per launch than a small kernel runs, so queue some tens of ms of unrelated work first: the
        # launches then wait in the stream and run back to back, as they do in the graph replay.
        pad = torch.randn(8192, 8192, device=device)
        torch.cuda.synchronize()
        N.prof_enable(True)
        for _ in range(4):
            pad = torch.mm(pad, pad).clamp_(-1.0, 1.0)
        for _ in range(profile_steps):
            step()
        torch.cuda.synchronize()
        summ = N.prof_summary()
        N.prof_enable(False)
        del pad
        # An event between two kernels costs the stream a few microseconds of its own.  The timed region
        # ran the same launches without events, so the difference between the two totals, spread
        # over the launches, is that cost; it is taken off every launch.
        n_launch = sum(r["launches"] for r in summ.values())
        ev_ms_per_step = sum(r["ms"] for r in summ.values()) / profile_steps
        event_cost_ms = max(0.0, (ev_ms_per_step - step_ms) / (n_launch / profile_steps))
        for r in summ.values():
            r["ms_net"] = max(r["ms"] - event_cost_ms * r["launches"], 0.05 * r["ms"])
        name, rec = max(summ.items(), key=lambda kv: kv[1]["ms_net"])
        per_launch_s = rec["ms_net"] * 1e-3 / rec["launches"]
        ai = rec["flops"] / max(rec["bytes"], 1.0)
        # f32_split: an fp32-accurate product costs six bf16 MFMA products (exact three-way operand split),
        # so the ceiling for fp32 work on that pipe is the bf16 peak / 6
        mfma_peak = MFMA_F32_TFLOPS
        if name in MFMA_CLASSES:
            mfma_peak = {"bf16": MFMA_BF16_TFLOPS, "f32_split": MFMA_BF16_TFLOPS / 6.0}.get(precision, MFMA_F32_TFLOPS)
        if ai >= mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
            ach = rec["flops"] / rec["launches"] / per_launch_s / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": mfma_peak, "unit": "TFLOP/s", "frac": ach / mfma_peak}
            if precision == "f32_split":
                roofline["frac_of_f32_mfma_peak"] = ach / MFMA_F32_TFLOPS
        else:
            ach = rec["bytes"] / rec["launches"] / per_launch_s / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        traffic, source = pmc_traffic(workload, name)
        roofline.update(traffic=traffic, traffic_source=source, kernel=name,
                        algorithmic_bytes_per_launch=rec["bytes"] / rec["launches"],
                        algorithmic_flops_per_launch=rec["flops"] / rec["launches"],
                        launches_per_step=rec["launches"] / profile_steps, avg_launch_us=per_launch_s * 1e6,
                        event_cost_us_removed_per_launch=event_cost_ms * 1e3,
                        share_of_kernel_time=rec["ms_net"] / sum(r["ms_net"] for r in summ.values()),
                        kernels={k: {"us_per_step": v["ms_net"] * 1e3 / profile_steps,
                                     "launches_per_step": v["launches"] / profile_steps} for k, v in summ.items()})
        return roofline

    head = measure_imagine(args.workload, args.steps, args.warmup, args.profile_steps)
    B, precision, elapsed, roofline = head["B"], head["precision"], head["elapsed"], head["roofline"]

    # beside the headline (default workload only): the same block with the gather tables rebuilt before every block,
    # and the large-batch configuration (4096 start states: the MFMA-bound case) in both fp32 forms
    extras = {}
    if args.workload == "imagine16" and not args.no_extras:
        r = measure_imagine("imagine16", args.steps, min(args.warmup, 10), 0, repack=True)
        extras["with_table_repack"] = {"value": r["value"], "ms_per_step": r["ms_per_step"],
                                       "note": "one-hot gather tables rebuilt from the weights before every block (seven "
                                               "launches), as after an optimiser step in training"}
        c3 = {}
        for wl in ("imagine4096", "imagine4096_split"):
            r = measure_imagine(wl, 20, 5, 3)
            c3[r["precision"]] = {"value": r["value"], "unit": "imagined latent steps/s", "ms_per_step": r["ms_per_step"],
                                  "steps": r["steps"], "warmup": r["warmup"], "B_per_gpu": r["B"], "H": H,
                                  "algorithmic_tflops": r["value"] * FLOP_PER_LATENT_STEP / 1e12, "roofline": r["roofline"]}
        c3["note"] = ("f32: fp32 matrix instructions (v_mfma_f32_32x32x2_f32); f32_split: the same fp32-accurate products from an "
                      "exact 3-way bf16 split on the bf16 matrix pipe, the engine's default form at 256 trajectories and more")
        extras["c3"] = c3
        torch.cuda.empty_cache()

    # the three HBM-bound kernels of the path at device-filling sizes (they are single 6-us launches inside the
    # block above): achieved GB/s against the 8 TB/s peak, algorithmic bytes / event-timed launch
    stream_kernels = None
    if rank == 0 and world == 1 and not args.no_stream_kernels:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_stream
            stream_kernels = {"peak_GB/s": bench_stream.PEAK_GBS,
                              "rows": bench_stream.run(large_only=True, device=str(device), verbose=False)}
        except Exception as e:
            stream_kernels = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    # the scaled model (recurrent width 4096, 32 x 32 latent): world-model loss + backward over the 16 x 64 batch and
    # the imagination block, fp32 products against bf16 matrix inputs read from bf16 weight images (one GPU, rank 0)
    if rank == 0 and world == 1 and args.workload == "imagine16" and not args.no_extras:
        try:
            extras["c5"] = measure_scaled_model(device)
            progress("scaled model (D=4096): " + ", ".join(f"{k} {v:.2f}" for k, v in extras["c5"].items() if isinstance(v, float)))
        except Exception as e:
            extras["c5"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    def emit(train):
        if rank != 0:
            return
        value = world * B * H * args.steps / elapsed
        line = {
            "metric": "imagined latent steps/sec", "value": value, "unit": "imagined latent steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f32_split": "f32 (matrix products from an exact 3-way bf16 split, six partial products, f32 accumulation)",
                      "bf16": "bf16 matrix inputs, f32 accumulation and elementwise"}[precision], "data": "synthetic",
            "config": {"workload": f"imagination block, default config (D=512, 32x32 latent, mlp 512, A=6), B={B} start "
                                   f"states per GPU x H={H}, {precision}, noise generation included",
                       "B_per_gpu": B, "H": H},
            "algorithmic_tflops": value * FLOP_PER_LATENT_STEP / 1e12,
            "roofline": roofline,
            "stream_kernels": stream_kernels,
            "train": train,
        }
        line.update(extras)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, wm, actor, critic, 16, H)
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)

    train = None
    if args.train_steps > 0:
        progress("training updates: building the learner, warm-up (includes the convolution kernel search)")
        # The headline measurement is complete at this point.  If the training part stalls (a rank lost in a
        # collective), every rank gives up after the limit and rank 0 still prints the line.
        import threading

        def give_up():
            progress(f"training updates: no result after {args.train_limit_s} s, reporting without them")
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)   # where each rank is stuck
            emit({"error": f"timeout after {args.train_limit_s} s"})
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(1)       # the headline line is out; a stalled training section must not look like success

        guard = threading.Timer(args.train_limit_s, give_up)
        guard.daemon = True
        guard.start()
        try:
            train = measure_train(device, world, rank, args.train_steps, 6, fence)
            progress(f"training updates: {train['ms_per_update']:.2f} ms per update")
        except Exception as e:  # the headline line above is already measured: report the failure beside it
            train = {"error": f"{type(e).__name__}: {e}"}
            progress(f"training updates failed: {train['error']}")
        guard.cancel()

    emit(train)
    failed = isinstance(train, dict) and "error" in train
    sys.stdout.flush()
    sys.stderr.flush()
    if world > 1:
        # every number is out; leave without the group's teardown, which would wait for a rank that failed
        os._exit(1 if failed else 0)
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
