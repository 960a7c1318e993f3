"""Three 256-row belief updates: one merged 768-row reverse loop against three concurrent chains (DiffusionActiveInference._denoise_chains)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
import bench
agent = bench.make_agent(torch.device("cuda"))
ai = agent.active_inference
obs = [torch.randn(256, 17, device="cuda") for _ in range(3)]
for mode in (False, True, False, True):
    ai.concurrent_belief_loops = mode
    for _ in range(2): ai.denoise_many(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ai.denoise_many(obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"concurrent chains {mode}: {dt*1e3:8.2f} ms per 3 x 256-row 50-step update", flush=True)
from aid_amd.streams import side_stream
side = side_stream()
ai.concurrent_belief_loops = True
with torch.cuda.stream(side):
    for _ in range(2): ai.denoise_many(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ai.denoise_many(obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"concurrent chains from a side stream: {dt*1e3:8.2f} ms", flush=True)
