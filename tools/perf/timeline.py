"""Where the two streams of a cfg-2 train_step are at which time: event marks recorded by DiffusionStateAgent (agent.timeline), ms from the step's start,
averaged over back-to-back steps (no synchronisation between them: the steady state bench.py times).  AID_NO_LOOKAHEAD=1: without the look-ahead ELBO."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
import bench
agent = bench.make_agent(torch.device("cuda"))
if os.environ.get("AID_NO_LOOKAHEAD"): agent.lookahead_elbo = False
for _ in range(4): agent.train_step()
n = 10; steps = []
for _ in range(n + 1):
    agent.timeline = []
    agent.train_step()
    steps.append(agent.timeline)
agent.timeline = None
torch.cuda.synchronize()
acc = collections.OrderedDict()
for i in range(n):
    t0, h0 = steps[i][0][1], steps[i][0][2]
    for label, ev, ht in steps[i][1:] + [("start of the next step",) + tuple(steps[i + 1][0][1:])]:
        a = acc.setdefault(label, [0.0, 0.0]); a[0] += t0.elapsed_time(ev) / n; a[1] += 1e3 * (ht - h0) / n
for label, (ms, hms) in acc.items():
    print(f"{ms:8.2f} ms on the GPU   (enqueued by the host at {hms:7.2f} ms)  {label}")
