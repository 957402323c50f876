"""The training loop end to end on a host device with the synthetic environment (CPU only)."""
import numpy as np
import torch

from dreamer_amd.config import Config
from dreamer_amd.trainer import train
from dreamer_amd.utils import SyntheticPixelEnv


class _Writer:
    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, step):
        self.scalars.setdefault(tag, []).append((step, value))

    def add_histogram(self, *a, **k):
        pass


def _tiny_cfg(iters):
    Config.num_latents, Config.classes_per_latent = 4, 32
    cfg = Config(env_id="synthetic", device=torch.device("cpu"), num_iterations=iters, replay_capacity=64, batch_size=2,
                 batch_length=6, train_ratio=12.0, imagination_horizon=3, hidden_size=32, base_cnn_channels=2,
                 mlp_hidden_units=32, log_interval=2, save_interval=10 ** 9, video_interval=10 ** 9, checkpoint_dir=None)
    return cfg


def test_train_runs_updates_and_logs():
    try:
        torch.manual_seed(0); np.random.seed(0)
        cfg = _tiny_cfg(12)
        w = _Writer()
        learner = train(cfg, summary_writer=w, env=SyntheticPixelEnv(16, n_actions=3, episode_len=5), verbose=False)
        # gate: len(buffer) > batch_length (6) -> first updates at env step 7; credit 7*12/12 = 7 updates there, then 1 per step
        assert "train/world_model_loss" in w.scalars and "perf/updates_per_second" in w.scalars
        assert all(np.isfinite(v) for _, v in w.scalars["train/world_model_loss"])
        assert learner.actor.ret_scale.p95.item() != 0.0
        sd = learner.state_dicts()
        assert set(sd) == {"world_model", "actor", "critic", "optim_world_model", "optim_actor", "optim_critic"}
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32


def test_resume_restores_models_optimisers_replay_and_counters(tmp_path):
    """A run that saved a checkpoint (ckpt.pth with the reference's six entries + trainer_state, replay.pth beside it)
    is continued by a second ``train(resume=...)``: same parameters, optimiser moments, replay contents, generator
    states and counters as at the save point; files load with ``weights_only=True``."""
    from dreamer_amd.replay_buffer import ReplayBuffer
    from dreamer_amd.trainer import Learner, load_checkpoint
    try:
        torch.manual_seed(0); np.random.seed(0)
        cfg = _tiny_cfg(14)
        cfg.checkpoint_dir, cfg.save_interval = str(tmp_path), 4
        first = train(cfg, env=SyntheticPixelEnv(16, n_actions=3, episode_len=5), verbose=False, save_replay=True)
        sd = torch.load(tmp_path / "ckpt.pth", weights_only=True)
        assert set(sd) == {"world_model", "actor", "critic", "optim_world_model", "optim_actor", "optim_critic", "trainer_state"}
        counters = sd["trainer_state"]["counters"]
        assert counters["updates_done"] == 13 and counters["policy_steps"] == 13   # saved after update number 12
        # a fresh learner + ring restored from the files equals the saved state
        learner = Learner(cfg, (3, 16, 16), 3)
        ring = ReplayBuffer(cfg.replay_capacity, (3, 16, 16), cfg.batch_length, cfg.device)
        got = load_checkpoint(str(tmp_path), learner, ring)
        assert got == counters
        for k, v in learner.world_model.state_dict().items():
            assert torch.equal(v, sd["world_model"][k]), k
        assert torch.equal(learner.actor.ret_scale.p95, sd["actor"]["ret_scale.p95"])
        st = learner.optim_critic.state_dict()
        assert sorted(st["state"]) == sorted(sd["optim_critic"]["state"])
        assert len(ring) == 13 and ring.idx == 13
        assert torch.equal(ring.observations[:13], torch.load(tmp_path / "replay.pth", weights_only=True)["observations"])
        after = np.random.randint(0, 1 << 30)
        load_checkpoint(str(tmp_path), learner, ring)
        assert np.random.randint(0, 1 << 30) == after      # the numpy generator continues from its saved state
        # and train() picks the run up where the file left it: iterations 14..20
        cfg2 = _tiny_cfg(20)
        cfg2.checkpoint_dir, cfg2.save_interval = str(tmp_path), 10 ** 9
        w = _Writer()
        second = train(cfg2, summary_writer=w, env=SyntheticPixelEnv(16, n_actions=3, episode_len=5), verbose=False,
                       resume=str(tmp_path))
        steps = [s for s, _ in w.scalars["train/world_model_loss"]]
        assert min(steps) >= 13 and max(steps) == 18
        assert float(second.optim_world_model.state_dict()["state"][0]["step"]) == 20.0
        assert first is not second
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32


def test_lib_alias_package_matches_reference_imports():
    from lib.config import Config as C2
    from lib.trainer import train as t2
    from lib.world_model import WorldModel, WorldModelState  # noqa: F401
    from lib.replay_buffer import ReplayBuffer  # noqa: F401
    from lib.actor import Actor  # noqa: F401
    from lib.critic import Critic  # noqa: F401
    from lib.utils import symexp, make_env  # noqa: F401
    assert C2 is Config and t2 is train


def test_replay_file_of_another_checkpoint_is_not_paired_with_newer_counters(tmp_path):
    """The ring file carries the counters of the checkpoint it was written with; a checkpoint saved later without the
    ring removes it, a stamp that differs is skipped, and a ring written for another sequence length is refused."""
    import pytest

    from dreamer_amd.replay_buffer import ReplayBuffer
    from dreamer_amd.trainer import Learner, load_checkpoint, save_checkpoint
    try:
        cfg = _tiny_cfg(1)
        learner = Learner(cfg, (3, 16, 16), 3)
        ring = ReplayBuffer(16, (3, 16, 16), 6, cfg.device)
        for i in range(9):
            ring.store(np.full((3, 16, 16), i, np.uint8), i % 3, float(i), True)
        save_checkpoint(str(tmp_path), learner, {"updates_done": 5, "policy_steps": 9, "update_credit": 0.0}, ring)
        fresh = ReplayBuffer(16, (3, 16, 16), 6, cfg.device)
        load_checkpoint(str(tmp_path), learner, fresh)
        assert len(fresh) == 9
        # a ring stamped for other counters (as a crash between the two renames leaves it) is skipped
        ring.save(str(tmp_path / "replay.pth"), stamp=(4, 8))
        fresh = ReplayBuffer(16, (3, 16, 16), 6, cfg.device)
        load_checkpoint(str(tmp_path), learner, fresh)
        assert len(fresh) == 0
        # saving without the ring removes the file of the earlier checkpoint
        save_checkpoint(str(tmp_path), learner, {"updates_done": 6, "policy_steps": 10, "update_credit": 0.0}, None)
        assert not (tmp_path / "replay.pth").exists()
        # sequence length is part of the file
        ring.save(str(tmp_path / "other.pth"))
        with pytest.raises(ValueError, match="sequences of 6"):
            ReplayBuffer(16, (3, 16, 16), 5, cfg.device).load(str(tmp_path / "other.pth"))
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32


def _resume_worker(rank, world, port, directory, out):
    import os

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        torch.manual_seed(0); np.random.seed(0)
        cfg = _tiny_cfg(14)
        cfg.checkpoint_dir, cfg.save_interval = directory, 4
        train(cfg, env=SyntheticPixelEnv(16, n_actions=3, episode_len=5, seed=rank), verbose=False, save_replay=True)
        # the resumed ranks: generator states and rings are their own, not rank 0's
        torch.manual_seed(123); np.random.seed(123)        # whatever state the new process starts with
        from dreamer_amd.replay_buffer import ReplayBuffer
        from dreamer_amd.trainer import Learner, load_checkpoint
        learner = Learner(cfg, (3, 16, 16), 3)
        ring = ReplayBuffer(cfg.replay_capacity, (3, 16, 16), cfg.batch_length, cfg.device)
        load_checkpoint(directory, learner, ring, rank=rank)
        starts = ring.draw_starts(8)
        torch.save({"starts": torch.from_numpy(starts), "obs": ring.observations[:len(ring)].clone(),
                    "torch_draw": torch.rand(4)}, os.path.join(out, f"r{rank}.pt"))
        # without its own generator file a rank still ends up on a stream of its own
        if rank == 1:
            os.remove(os.path.join(directory, "rng.rank1.pth"))
        dist.barrier()
        load_checkpoint(directory, learner, ring, rank=rank)
        torch.save({"starts": torch.from_numpy(ring.draw_starts(8)), "torch_draw": torch.rand(4)},
                   os.path.join(out, f"r{rank}_reseeded.pt"))
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32
        if dist.is_initialized():
            dist.destroy_process_group()


def test_resume_under_data_parallelism_keeps_the_ranks_apart(tmp_path):
    """Two ranks (gloo) train, checkpoint and resume: each rank comes back with its own generator states and its own
    replay ring, so the ranks keep drawing different batches (restoring rank 0's states on every rank would average
    N copies of one gradient)."""
    import torch.multiprocessing as mp

    from test_dp_gloo import _free_port
    ck, out = tmp_path / "ck", tmp_path / "out"
    ck.mkdir(); out.mkdir()
    mp.spawn(_resume_worker, args=(2, _free_port(), str(ck), str(out)), nprocs=2, join=True)
    for f in ("ckpt.pth", "replay.pth", "replay.rank1.pth"):
        assert (ck / f).exists(), f
    a, b = torch.load(out / "r0.pt", weights_only=True), torch.load(out / "r1.pt", weights_only=True)
    assert not torch.equal(a["starts"], b["starts"]) and not torch.equal(a["torch_draw"], b["torch_draw"])
    assert not torch.equal(a["obs"], b["obs"])              # each rank's own environment frames
    a2, b2 = torch.load(out / "r0_reseeded.pt", weights_only=True), torch.load(out / "r1_reseeded.pt", weights_only=True)
    assert torch.equal(a2["starts"], a["starts"])           # rank 0 resumes the same stream every time
    assert not torch.equal(b2["starts"], a2["starts"]) and not torch.equal(b2["torch_draw"], a2["torch_draw"])
