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


def test_lib_alias_package_matches_reference_imports():
    from lib.config import Config as C2
    from lib.trainer import train as t2
    from lib.world_model import WorldModel, WorldModelState  # noqa: F401
    from lib.replay_buffer import ReplayBuffer  # noqa: F401
    from lib.actor import Actor  # noqa: F401
    from lib.critic import Critic  # noqa: F401
    from lib.utils import symexp, make_env  # noqa: F401
    assert C2 is Config and t2 is train
