"""Training updates on the GPU: the graph-replayed actor / critic updates against the eager ones."""
import numpy as np
import pytest
import torch

from dreamer_amd.config import Config
from dreamer_amd.replay_buffer import ReplayBuffer
from dreamer_amd.trainer import Learner

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _run(updates, graphs, monkeypatch):
    monkeypatch.setenv("DRM_NO_UPDATE_GRAPHS", "0" if graphs else "1")
    Config.num_latents, Config.classes_per_latent = 8, 32
    try:
        cfg = Config(device=DEV, batch_size=4, batch_length=6, imagination_horizon=5, hidden_size=64, base_cnn_channels=4,
                     mlp_hidden_units=64)
        torch.manual_seed(0)
        np.random.seed(0)
        learner = Learner(cfg, (3, 16, 16), 4)
        with torch.no_grad():
            for head in (learner.world_model.rew.head, learner.actor.head, learner.critic.head):
                torch.nn.init.normal_(head.weight, std=0.05)
        rng = np.random.default_rng(7)
        buf = ReplayBuffer(64, (3, 16, 16), cfg.batch_length, DEV)
        buf.store_many(rng.integers(0, 256, size=(64, 3, 16, 16), dtype=np.uint8), rng.integers(0, 4, size=64),
                       rng.standard_normal(64).astype(np.float32), rng.random(64) > 0.1)
        losses = []
        for _ in range(updates):
            out = learner.update(buf.sample(cfg.batch_size, want_f32=True))
            losses.append([float(out[k]) for k in ("world_model_loss", "critic_loss", "actor_loss")])
        return learner, np.array(losses)
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32


def test_graph_replayed_updates_match_eager(monkeypatch):
    eager, le = _run(7, False, monkeypatch)
    graphed, lg = _run(7, True, monkeypatch)
    assert graphed.eager_updates == 3 and graphed._critic_graph.g_loss is not None
    assert eager._critic_graph.g_loss is None
    # identical kernels in identical order up to the 3rd update; afterwards the same arithmetic replayed
    np.testing.assert_allclose(lg, le, rtol=2e-5, atol=1e-6)
    for name in ("actor", "critic", "world_model"):
        pe, pg = dict(getattr(eager, name).named_parameters()), dict(getattr(graphed, name).named_parameters())
        for k in pe:
            assert torch.allclose(pg[k], pe[k], rtol=1e-4, atol=1e-6), (name, k)
    assert torch.allclose(graphed.actor.ret_scale.p95, eager.actor.ret_scale.p95, rtol=1e-5)


def test_acting_policy_graph_replay_matches_eager_steps(monkeypatch):
    """The recorded policy step (frame -> posterior state -> action) against the same step launched
    eagerly: same generator state -> same actions and states, also after the parameters changed in place."""
    import numpy as np
    from dreamer_amd.config import Config
    from dreamer_amd.trainer import ActingPolicy, Learner
    cfg = Config(device=torch.device("cuda:0"))
    frames = np.random.default_rng(3).integers(0, 256, size=(12, 3, 64, 64), dtype=np.uint8)

    def run(graph):
        monkeypatch.setenv("DRM_NO_ACT_GRAPH", "0" if graph else "1")
        torch.manual_seed(5)
        learner = Learner(cfg, (3, 64, 64), 6)
        with torch.no_grad():
            torch.nn.init.normal_(learner.actor.head.weight, std=0.5)
        policy = ActingPolicy(learner.world_model, learner.actor, cfg.device, (3, 64, 64))
        torch.manual_seed(9)
        policy.reset()
        acts, hs = [], []
        for i in range(12):
            if i == 8:   # an optimiser step between two environment steps
                with torch.no_grad():
                    for p in learner.world_model.seq.parameters():
                        p.mul_(1.01)
                    for p in learner.actor.parameters():
                        p.mul_(0.97)
            a = policy.act(frames[i])
            policy.feedback(a, i != 5)   # one episode boundary
            acts.append(a)
            hs.append(policy.h.clone())
        return policy, acts, torch.stack(hs)

    pe, ae, he = run(False)
    pg, ag, hg = run(True)
    assert pe.graph is None and pg.graph is not None
    assert ae == ag
    assert float((he - hg).abs().max()) < 1e-5
    assert len(set(ae)) > 1


def test_train_loop_end_to_end_on_the_gpu():
    """`train` (lib/trainer.py:16-340) with the synthetic environment on the GPU: recorded policy step, replay
    ring in HBM, eager then graph-replayed updates, flat optimiser, episode resets, logging, checkpoint layout."""
    import numpy as np
    from dreamer_amd.config import Config
    from dreamer_amd.trainer import train
    from dreamer_amd.utils import SyntheticPixelEnv

    class Writer:
        def __init__(self):
            self.scalars = {}

        def add_scalar(self, tag, value, step):
            self.scalars.setdefault(tag, []).append((step, float(value)))

        def add_histogram(self, *a, **k):
            pass

    try:
        Config.num_latents, Config.classes_per_latent = 8, 32
        torch.manual_seed(0); np.random.seed(0)
        cfg = Config(env_id="synthetic", device=torch.device("cuda:0"), num_iterations=40, replay_capacity=64, batch_size=4,
                     batch_length=6, train_ratio=24.0, imagination_horizon=3, hidden_size=64, base_cnn_channels=4,
                     mlp_hidden_units=64, log_interval=2, save_interval=10 ** 9, video_interval=10 ** 9, checkpoint_dir=None)
        w = Writer()
        learner = train(cfg, summary_writer=w, env=SyntheticPixelEnv(16, n_actions=3, episode_len=7), verbose=False)
        torch.cuda.synchronize()
        assert learner.eager_updates == 3 and learner._critic_graph.g_loss is not None   # graph-replayed updates were reached
        losses = [v for _, v in w.scalars["train/world_model_loss"]]
        assert len(losses) >= 5 and all(np.isfinite(losses))
        assert losses[-1] < losses[0]          # it learns the synthetic frames
        sd = learner.state_dicts()
        assert set(sd) == {"world_model", "actor", "critic", "optim_world_model", "optim_actor", "optim_critic"}
        assert float(sd["optim_world_model"]["state"][0]["step"]) >= 10
        for p in learner.world_model.parameters():
            assert torch.isfinite(p).all()
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32


def test_checkpoint_resume_on_the_gpu(tmp_path):
    """Checkpoint written from the flat optimisers and the HBM replay ring, restored into a fresh learner / ring with
    ``weights_only=True`` loads: parameters (in place, inside the flat buffers), Adam moments, replay bytes, counters."""
    import numpy as np
    from dreamer_amd.config import Config
    from dreamer_amd.trainer import Learner, load_checkpoint, train
    from dreamer_amd.utils import SyntheticPixelEnv
    try:
        Config.num_latents, Config.classes_per_latent = 8, 32
        torch.manual_seed(0); np.random.seed(0)
        cfg = Config(env_id="synthetic", device=DEV, num_iterations=14, replay_capacity=64, batch_size=4,
                     batch_length=6, train_ratio=24.0, imagination_horizon=3, hidden_size=64, base_cnn_channels=4,
                     mlp_hidden_units=64, log_interval=10 ** 9, save_interval=4, video_interval=10 ** 9, checkpoint_dir=str(tmp_path))
        first = train(cfg, env=SyntheticPixelEnv(16, n_actions=3, episode_len=7), verbose=False, save_replay=True)
        sd = torch.load(tmp_path / "ckpt.pth", weights_only=True, map_location="cpu")
        learner = Learner(cfg, (3, 16, 16), 3)
        ring = ReplayBuffer(64, (3, 16, 16), cfg.batch_length, DEV)
        flat_ptr = learner.g_wm.flat_params.data_ptr()
        counters = load_checkpoint(str(tmp_path), learner, ring)
        assert counters["updates_done"] == sd["trainer_state"]["counters"]["updates_done"] >= 5
        assert learner.g_wm.flat_params.data_ptr() == flat_ptr
        for k, v in learner.world_model.state_dict().items():
            assert torch.equal(v.cpu(), sd["world_model"][k]), k
        for p in learner.world_model.parameters():   # still views of the flat buffer
            assert flat_ptr <= p.data_ptr() < flat_ptr + learner.g_wm.flat_params.numel() * 4
        mine, theirs = learner.optim_world_model.state_dict(), sd["optim_world_model"]
        assert sorted(mine["state"]) == sorted(theirs["state"])
        for i in theirs["state"]:
            assert torch.equal(mine["state"][i]["exp_avg"].cpu(), theirs["state"][i]["exp_avg"])
        n = int(counters["policy_steps"])
        assert len(ring) == n and torch.equal(ring.observations[:n], first_ring_rows(tmp_path, n))
        out = learner.update(ring.sample(cfg.batch_size, want_f32=True))     # and it trains on from there
        assert torch.isfinite(out["world_model_loss"]).all()
    finally:
        Config.num_latents, Config.classes_per_latent = 32, 32


def first_ring_rows(tmp_path, n):
    return torch.load(tmp_path / "replay.pth", weights_only=True)["observations"][:n].to(DEV)
