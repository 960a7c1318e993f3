"""A/B: the estimator's full-row kernels launched with at most n workgroups along the rows (grid-stride over row tiles; aid_rowln_debug(1000 + n)), so that
they hold n CUs and leave the rest - and the workgroup dispatcher - to the chains.  usage: python3 tools/perf/rowln_persist.py n [n ...]   (0 = one per tile)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda"))
for _ in range(3): agent.train_step()
def run(n=10):
    for _ in range(2): agent.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): agent.train_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for n in [int(a) for a in sys.argv[1:]] or [0, 256, 224, 208, 192]:
        nat.lib().aid_rowln_debug(1000 + n)
        print(f"rowln workgroups <= {n:4d}: {run():.2f} ms/step", flush=True)
nat.lib().aid_rowln_debug(1000)
