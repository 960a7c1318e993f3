"""cfg 4 (pixels, B = 256): the belief chains started early (chain a under encode(next_obs), chain b under ELBO / InfoNCE / encoder backward) against the paired loop
after both encoder passes, alternating in one process."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench, aid_amd
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=256, efe_horizon=bench.HORIZON, latent_dim=128, hidden_dim=bench.D, pixel_observation=True)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=bench.NSTEPS)
class Env4: observation_space, action_space = bench._Box((3, 3, 84, 84)), bench._Box((6,))
agent = aid_amd.DiffusionPixelAgent(Env4(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
rng = np.random.default_rng(0); n = 512
frames = rng.integers(0, 256, (n, 3, 3, 84, 84), dtype=np.uint8)
agent.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (n, 6)).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(n) < 0.01)
def run(k=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): agent.train_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for rep in range(3):
    for early in (False, True):
        agent.early_belief_chains = early
        run(3)
        print(f"early_belief_chains={early}: {run():.2f} ms per train_step", flush=True)
