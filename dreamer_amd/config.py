"""Hyper-parameter record with the reference's field names and defaults (lib/config.py:8-68).

``num_latents`` / ``classes_per_latent`` are deliberately un-annotated class attributes, not
dataclass fields, as in the reference (lib/config.py:23-24): ``Config(num_latents=...)`` is a
TypeError there too; set them on the class or the instance.
"""
import datetime as _dt
import os as _os
from dataclasses import dataclass

import torch


def _default_device() -> torch.device:
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


_RUN_STAMP = _dt.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")


@dataclass
class Config:
    # environment / device
    env_id: str = "ALE/Pong-v5"
    device: torch.device = _default_device()

    # replay + update schedule
    num_iterations: int = 1_000_000
    replay_capacity: int = 100_000
    batch_size: int = 16
    batch_length: int = 64
    train_ratio: int = 1024.0

    # world model
    num_latents = 32
    classes_per_latent = 32
    beta_pred: float = 1.0
    beta_dyn: float = 1.0
    beta_rep: float = 0.1
    unimix_eps: float = 0.01
    free_bits: float = 1.0
    world_model_lr: float = 3e-4
    world_model_adam_eps: float = 1e-8
    world_model_grad_clip: float = 1000.0

    # actor / critic
    imagination_horizon: int = 15
    gamma: float = 0.997
    lam: float = 0.95
    critic_num_buckets: int = 255
    critic_bucket_min: float = -20.0
    critic_bucket_max: float = 20.0
    critic_ema_decay: float = 0.98
    critic_ema_regularizer: float = 1.0
    return_normalization_limit: float = 1.0
    return_normalization_decay: float = 0.99
    actor_entropy_scale: float = 3e-4
    actor_ret_norm_limit: float = 1.0
    actor_ret_norm_decay: float = 0.99
    actor_critic_lr: float = 3e-5
    actor_critic_adam_eps: float = 1e-5
    actor_critic_grad_clip: float = 100.0

    # widths
    hidden_size: int = 512
    base_cnn_channels: int = 32
    mlp_hidden_units: int = 512

    # logging / checkpoints
    create_artifacts: bool = True
    run_dir: str = _os.path.join("runs", _RUN_STAMP)
    checkpoint_dir: str = _os.path.join(run_dir, "checkpoints")
    log_interval: int = 10
    save_interval: int = 1000
    video_interval: int = 500
    video_fps: int = 20
    video_max_frames: int = 500

    seed: int = 42
