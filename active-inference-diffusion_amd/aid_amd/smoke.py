"""full_smoke(): one tiny DiffusionStateAgent.train_step + act on cuda:0 in production (Philox) mode."""
import numpy as np


class _Box:
    def __init__(self, shape):
        self.shape = shape


class _Env:
    observation_space, action_space = _Box((17,)), _Box((6,))


def full_smoke():
    import aid_amd
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=8, efe_horizon=2, latent_dim=17, hidden_dim=64)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=4)
    agent = aid_amd.DiffusionStateAgent(_Env(), cfg, aid_amd.TrainingConfig(buffer_size=32))
    rng = np.random.default_rng(0)
    agent.replay_buffer.add_batch(rng.standard_normal((32, 17)), rng.uniform(-1, 1, (32, 6)), rng.standard_normal(32), rng.standard_normal((32, 17)), np.zeros(32))
    m = agent.train_step()
    assert m and all(np.isfinite(v) for v in m.values()), m
    a, _ = agent.act(rng.standard_normal(17).astype(np.float32))
    assert a.shape == (6,) and np.all(np.isfinite(a))
    print("full_smoke ok: train_step metrics", {k: round(v, 5) for k, v in list(m.items())[:4]})
