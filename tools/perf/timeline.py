"""Where the two streams of a cfg-2 train_step are at which time: event marks recorded by DiffusionStateAgent (agent.timeline), ms from the step's start.
Averaged over a few steps after warm-up."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
import bench
agent = bench.make_agent(torch.device("cuda"))
for _ in range(4): agent.train_step()
acc = collections.OrderedDict(); n = 8
for _ in range(n):
    agent.timeline = []
    agent.train_step(); torch.cuda.synchronize()
    t0 = agent.timeline[0][1]
    for label, ev in agent.timeline[1:]:
        acc[label] = acc.get(label, 0.0) + t0.elapsed_time(ev) / n
agent.timeline = None
for label, ms in acc.items():
    print(f"{ms:8.2f} ms  {label}")
