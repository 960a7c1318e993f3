"""A/B: priority of the side stream (score step + reverse loop under the EFE phase): default, high, low.  Both arms in one process."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
agent = bench.make_agent(torch.device("cuda"))
for _ in range(3): agent.train_step()
def run(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): agent.train_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for rep in range(3):
    for pr in (0, -1, 1):
        try:
            agent._comm = torch.cuda.Stream(device=agent.device, priority=pr)
        except Exception as e:
            print("priority", pr, "unsupported:", e); continue
        run(2)
        print(f"side-stream priority {pr:2d}: {run():.2f} ms/step", flush=True)
