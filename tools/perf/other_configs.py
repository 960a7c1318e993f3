"""Times the BASELINE.json configs that are not the bench line: cfg 3 (Hopper-v4 dims, batch 1024) and cfg 4
(pixel 84x84 frame_stack 3, DrQ-v2 encoder, batch 256).  GPU only:  python tools/perf/other_configs.py [hopper|pixel]"""
import os, sys, time, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
import aid_amd


class Box:
    def __init__(self, shape): self.shape = shape


def timed_steps(agent, warm=2, steps=5):
    for _ in range(warm):
        m = agent.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        m = agent.train_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, m


def hopper():
    O, A, B = 11, 3, 1024
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=5, latent_dim=O, hidden_dim=512)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=50)

    class Env: observation_space, action_space = Box((O,)), Box((A,))
    agent = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=8192))
    rng = np.random.default_rng(0); n = 4096
    agent.replay_buffer.add_batch(rng.standard_normal((n, O)), rng.uniform(-1, 1, (n, A)), rng.standard_normal(n), rng.standard_normal((n, O)), rng.random(n) < 0.01)
    if os.environ.get("AID_CHAIN_MAX_ROWS"): agent.active_inference.CHAIN_MAX_ROWS = int(os.environ["AID_CHAIN_MAX_ROWS"])      # A/B: chains for larger updates
    dt, m = timed_steps(agent)
    print(json.dumps({"config": "cfg3 hopper B=1024 H=5 N=50", "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "rows_per_s": B / dt,
                      "finite": bool(all(np.isfinite(v) for v in m.values()))}))


def pixel():
    L, A, B = 128, 6, 256
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=5, latent_dim=L, hidden_dim=512, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=50)

    class Env: observation_space, action_space = Box((3, 3, 84, 84)), Box((A,))
    agent = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=1024), aid_amd.PixelObservationConfig())
    g = torch.Generator(device="cuda").manual_seed(0); n = 1024
    rb = agent.replay_buffer
    rb.observations.copy_(torch.rand(n, 9 * 84 * 84, device="cuda", generator=g)); rb.next_observations.copy_(torch.rand(n, 9 * 84 * 84, device="cuda", generator=g))
    rb.actions.uniform_(-1, 1); rb.rewards.normal_(); rb.size = n
    dt, m = timed_steps(agent)
    from aid_amd import _native as nat
    x = rb.observations[:B].view(B, 9, 84, 84)
    enc = agent.encoder
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        f, ctx = enc(x, agent._encoder_noise(B), keep_for_backward=True)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 5
    d = torch.randn_like(f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        enc.backward(d, ctx)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 5
    print(json.dumps({"config": "cfg4 pixel 84x84x9 B=256 H=5 N=50 L=128", "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "encoder_fwd_ms": tf * 1e3,
                      "encoder_bwd_ms": tb * 1e3, "finite": bool(all(np.isfinite(v) for v in m.values())), "mem_GB": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    if os.environ.get("AID_ROWLN_DEBUG"):       # kernel A/B experiments (34 / 35: 16-deep 512-wide full-row kernels forced on / off)
        from aid_amd import _native as nat
        nat.lib().aid_rowln_debug(int(os.environ["AID_ROWLN_DEBUG"]))
    which = sys.argv[1:] or ["hopper", "pixel"]
    for w in which:
        {"hopper": hopper, "pixel": pixel}[w]()
