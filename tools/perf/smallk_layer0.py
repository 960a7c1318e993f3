"""The estimator's first decoder layer alone: 640 000 x 17 -> 1024, LayerNorm, SiLU (lin_smallk_ln_act_kernel<4, 2>) through decode_observation's entry."""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
z = torch.randn(640000, 17, device="cuda")
for _ in range(2): nat.decode_observation(ai.decoder_table(), z, None, ai._ws)
torch.cuda.synchronize()
nat.prof_enable(False)
import subprocess
t0 = time.perf_counter()
for _ in range(5): nat.decode_observation(ai.decoder_table(), z, None, ai._ws)
torch.cuda.synchronize(); print(f"decoder forward on 640 000 rows: {1e3*(time.perf_counter()-t0)/5:.3f} ms")
