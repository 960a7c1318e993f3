"""A/B: how many denoise steps share one hoisted conditioning pass (aid_denoise_debug): the [steps x rows, 26 D] modulation buffer of a 50-step, 256-row
loop is 681 MB - beyond the 256 MB Infinity Cache, so every adaLN of every step reads its rows from HBM; short chunks keep them cache-resident at the price
of a less efficient conditioning GEMM.  One process per setting (the chains are captured once).  usage: python3 tools/perf/chunk_steps.py <cap, 0 = all>"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
cap = int(sys.argv[1]); nat.lib().aid_denoise_debug(cap)
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
for _ in range(4): agent.train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): agent.train_step()
torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 10 * 1e3
b = agent.replay_buffer.sample(256); ai.concurrent_belief_loops = True
def three(): return ai.denoise_many([b["next_observations"], b["observations"], b["next_observations"]])
for _ in range(2): three()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): three()
torch.cuda.synchronize()
print(f"chunk cap {cap:3d}: train_step {step:.2f} ms   three chains alone {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
