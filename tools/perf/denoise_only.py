import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
import bench
agent = bench.make_agent(torch.device("cuda"))
ai = agent.active_inference
obs = torch.randn(int(os.environ.get("B","256")), 17, device="cuda")
for _ in range(3): ai.denoise_only(obs)
torch.cuda.synchronize()
