"""A/B in situ: the 256-row chain kernels on 32- / 64- / 96-row tiles (aid_mid_debug): whole train_step and the three chains alone.  The chains are hipGraph
replays, so each setting gets a fresh capture (graph cache cleared)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
for _ in range(3): agent.train_step()
def run(n=10):
    for _ in range(2): agent.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): agent.train_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def chains(n=5):
    b = agent.replay_buffer.sample(256)
    ai.concurrent_belief_loops = True
    for _ in range(2): ai.denoise_many([b["next_observations"], b["observations"], b["next_observations"]])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ai.denoise_many([b["next_observations"], b["observations"], b["next_observations"]])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for bmt in (1, 2, 3):
        torch.cuda.synchronize(); nat.lib().aid_mid_debug(bmt)
        for k in [k for k in vars(ai) if "graph" in k and isinstance(getattr(ai, k), dict)]: getattr(ai, k).clear()
        print(f"{32 * bmt:3d}-row tiles: train_step {run():.2f} ms   three chains alone {chains():.2f} ms", flush=True)
nat.lib().aid_mid_debug(1)
