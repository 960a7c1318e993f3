import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, aid_amd
import bench
mode = sys.argv[1]
a = bench.make_agent(torch.device("cuda"))
for _ in range(3): a.train_step()
ai = a.active_inference
if mode == "sb":
    bench.small_batch_loops(a)
elif mode == "eval_only":
    ai.eval(); ai.train(True)
elif mode == "eval_eager":     # eval-mode loops without graphs
    ai.eval(); ai.graph_small_batches = False
    for _ in range(5): ai.denoise_only(torch.randn(1, bench.O, device="cuda"))
    ai.train(True)
elif mode == "train_graph":
    for _ in range(5): ai.denoise_only(torch.randn(1, bench.O, device="cuda"))
torch.cuda.synchronize()
print(mode, bench.other_configs.__name__)
class Box:
    def __init__(self, shape): self.shape = shape
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=256, efe_horizon=5, latent_dim=128, hidden_dim=512, pixel_observation=True)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=50)
class Env: observation_space, action_space = Box((3, 3, 84, 84)), Box((6,))
pa = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
rng = np.random.default_rng(0); n = 512
frames = rng.integers(0, 256, (n, 3, 3, 84, 84), dtype=np.uint8)
pa.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (n, 6)).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(n) < 0.01)
ts = []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pa.train_step(); torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 1))
print(mode, "pixel steps:", ts, "training flags:", pa.active_inference.training, pa.encoder.training)
