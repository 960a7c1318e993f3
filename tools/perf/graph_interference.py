"""Does a captured hipGraph (the small-batch reverse loop of act()) slow later eager train_steps in the same process?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
import bench
def steps(agent, n=4):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); agent.train_step(); torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 1))
    return ts
agent = bench.make_agent(torch.device("cuda"))
steps(agent, 3)
print("before any graph:", steps(agent))
obs = np.random.default_rng(0).standard_normal(bench.O).astype(np.float32)
ai = agent.active_inference
ai.denoise_only(torch.randn(1, bench.O, device="cuda")); ai.denoise_only(torch.randn(32, bench.O, device="cuda")); torch.cuda.synchronize()
print("after two graphed small-batch loops:", steps(agent))
print("graphs alive:", len(ai.__dict__.get("_denoise_graphs", {})))
ai.__dict__.get("_denoise_graphs", {}).clear(); torch.cuda.synchronize()
print("after dropping the graphs:", steps(agent))
ai.graph_small_batches = False
agent.act(obs); torch.cuda.synchronize()
print("after an eager act():", steps(agent))

# a second agent of another kind while graphs of the first are alive
import aid_amd
ai.graph_small_batches = True
ai.denoise_only(torch.randn(1, bench.O, device="cuda")); torch.cuda.synchronize()
class Box:
    def __init__(self, shape): self.shape = shape
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=256, efe_horizon=5, latent_dim=128, hidden_dim=512, pixel_observation=True)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=50)
class Env: observation_space, action_space = Box((3, 3, 84, 84)), Box((6,))
pa = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
rng = np.random.default_rng(0); n = 512
frames = rng.integers(0, 256, (n, 3, 3, 84, 84), dtype=np.uint8)
pa.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (n, 6)).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(n) < 0.01)
steps(pa, 3)
print("pixel agent with the state agent's graph alive:", steps(pa))
ai.__dict__.get("_denoise_graphs", {}).clear(); torch.cuda.synchronize()
print("pixel agent after dropping the graph:", steps(pa))
