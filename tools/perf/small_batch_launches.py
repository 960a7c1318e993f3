"""Per-GEMM timings of ONE 50-step reverse loop at B rows (direct launches, library profile hooks): which kernel the small-batch loops spend their time in."""
import sys, os, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
ai.eval()
obs = torch.randn(B, bench.O, device="cuda")
for _ in range(2): ai.denoise_only(obs, ws=ai._ws)
torch.cuda.synchronize()
nat.prof_enable(True); nat.prof_reset(); ai.denoise_only(obs, ws=ai._ws); torch.cuda.synchronize()
path = tempfile.mktemp(suffix=".csv"); nat.prof_dump(path); nat.prof_enable(False)
os.system(f"python {R}/tools/perf/launch_table.py {path} > /tmp/lt.txt; head -14 /tmp/lt.txt")
