"""One small-batch reverse loop (eval mode, direct launches) for `rocprofv3 --kernel-trace --stats`.  usage: small_batch_prof.py ROWS [MODE]  (MODE -2: skinny / panel kernels)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
B = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else -1
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
ai.eval(); ai.graph_small_batches = False
nat.lib().aid_denoise_debug(mode)
obs = torch.randn(B, bench.O, device="cuda")
for _ in range(5): ai.denoise_only(obs)
torch.cuda.synchronize()
