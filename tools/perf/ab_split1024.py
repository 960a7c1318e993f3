"""A/B in one process: the estimator's 1024-wide Linear -> LayerNorm -> SiLU -> skip stage as ONE exclusive full-row kernel (136 KB of LDS: nothing shares
its CU) against two 512-column blocks of the co-resident plain kernel + a separate LayerNorm pass (aid_rowln_debug(40)).  Prices what a co-resident
1024-wide kernel could gain for the step: the belief chains then share CUs with every estimator kernel.  Step = cfg-2 train_step, dead belief elided."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda"))
agent.elide_dead_next_belief = os.environ.get("ELIDE", "1") == "1"
for _ in range(4): agent.train_step()
def run(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): agent.train_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for mode, name in ((41, "fused exclusive kernel"), (40, "split: co-resident GEMM + LN pass")):
        nat.lib().aid_rowln_debug(mode)
        run(2)
        print(f"{name}: {run():.2f} ms/step", flush=True)
nat.lib().aid_rowln_debug(41)
# the main-stream phase alone (no side stream work): what the split costs when nothing overlaps
agent.overlap_streams = False
for mode, name in ((41, "fused"), (40, "split")):
    nat.lib().aid_rowln_debug(mode); run(2)
    print(f"single stream, {name}: {run(5):.2f} ms/step", flush=True)
