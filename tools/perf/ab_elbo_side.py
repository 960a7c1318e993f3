"""A/B: ELBO on the side stream with / without the low-LDS dispatch hint, and everything on one stream.  One process, interleaved arms."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
agent = bench.make_agent(torch.device("cuda"))
for _ in range(3): agent.train_step()
def run(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): agent.train_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for ov, low in ((True, True), (True, False), (False, False)):
        agent.overlap_streams = ov; agent.side_low_lds = low; agent._comm = None
        run(2)
        print(f"overlap={ov} side_low_lds={low}: {run():.2f} ms/step", flush=True)
