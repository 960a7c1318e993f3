"""BASELINE configs[2] (Hopper state dims, B = 1024) alone: the program to put behind `rocprofv3 --kernel-trace --stats --`.  usage: python3 tools/perf/cfg3_profile.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench, aid_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
o3, a3, b3 = 11, 3, 1024
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=b3, efe_horizon=bench.HORIZON, latent_dim=o3, hidden_dim=bench.D)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=bench.NSTEPS)
class Env3: observation_space, action_space = bench._Box((o3,)), bench._Box((a3,))
agent = aid_amd.DiffusionStateAgent(Env3(), cfg, aid_amd.TrainingConfig(buffer_size=8192))
rng = np.random.default_rng(0); n = 4096
agent.replay_buffer.add_batch(rng.standard_normal((n, o3)), rng.uniform(-1, 1, (n, a3)), rng.standard_normal(n), rng.standard_normal((n, o3)), rng.random(n) < 0.01)
if os.environ.get("AID_NO_OVERLAP"): agent.overlap_streams = False
for _ in range(3): agent.train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): agent.train_step()
torch.cuda.synchronize(); print(f"cfg3: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per train_step")
