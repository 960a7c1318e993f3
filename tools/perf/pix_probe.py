"""Does a state agent that lived earlier in the process change the pixel agent's step time?  (bench.py's extras saw 161 vs 80 ms)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, aid_amd
import bench
from aid_amd import _native as nat
class Box:
    def __init__(self, shape): self.shape = shape
def mk():
    L, A, B = 128, 6, 256
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=5, latent_dim=L, hidden_dim=512, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=50)
    class Env: observation_space, action_space = Box((3, 3, 84, 84)), Box((A,))
    agent = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
    rng = np.random.default_rng(0); n = 512
    frames = rng.integers(0, 256, (n, 3, 3, 84, 84), dtype=np.uint8)
    agent.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (n, A)).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(n) < 0.01)
    return agent
def run(tag, agent, n=5):
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); agent.train_step(); torch.cuda.synchronize()
        print(tag, i, round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "state_first"
if mode == "state_first":
    a = bench.make_agent(torch.device("cuda"))
    run("state", a, 3)
    if "graphs" in sys.argv:
        print(bench.small_batch_loops(a)["act_b1"]["loop_us"])
    if "prof" in sys.argv:
        nat.prof_enable(True); nat.prof_reset(); a.train_step(); torch.cuda.synchronize(); nat.prof_read(nat.PROF_GEMM); nat.prof_enable(False); nat.prof_reset()
    del a; torch.cuda.empty_cache()
run("pixel", mk(), 5)
