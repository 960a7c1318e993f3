"""BASELINE configs[3] (pixels, B = 256, uint8 replay) alone: the program to put behind `rocprofv3 --kernel-trace --stats --`.  usage: python3 tools/perf/cfg4_profile.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench, aid_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=256, efe_horizon=bench.HORIZON, latent_dim=128, hidden_dim=bench.D, pixel_observation=True)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=bench.NSTEPS)
class Env4: observation_space, action_space = bench._Box((3, 3, 84, 84)), bench._Box((6,))
agent = aid_amd.DiffusionPixelAgent(Env4(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
rng = np.random.default_rng(0); n = 512
frames = rng.integers(0, 256, (n, 3, 3, 84, 84), dtype=np.uint8)
agent.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (n, 6)).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(n) < 0.01)
for _ in range(3): agent.train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): agent.train_step()
torch.cuda.synchronize(); print(f"cfg4: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per train_step")
if os.environ.get("AID_DUMP"):
    from aid_amd import _native as nat
    nat.prof_enable(True); nat.prof_reset(); agent.train_step(); torch.cuda.synchronize(); nat.prof_dump(os.environ["AID_DUMP"]); nat.prof_enable(False)
