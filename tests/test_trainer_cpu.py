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
