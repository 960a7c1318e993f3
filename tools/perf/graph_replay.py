import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
import bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda"))
ai = agent.active_inference
obs = torch.randn(256, 17, device="cuda"); obs2 = torch.randn(512, 17, device="cuda")
st = torch.cuda.Stream()
for o in (obs, obs2):
    with torch.cuda.stream(st):
        for _ in range(2): ai.denoise_only(o)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): ai.denoise_only(o)
        torch.cuda.synchronize(); eager=(time.perf_counter()-t0)/5
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            ai.denoise_only(o)
        g.replay(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); gr=(time.perf_counter()-t0)/5
    print(f"B={o.shape[0]}: eager {eager*1e3:.2f} ms, graph replay {gr*1e3:.2f} ms")
# host-only cost: time the C call with GPU idle → approximate by measuring enqueue time without sync
torch.cuda.synchronize(); t0=time.perf_counter(); ai.denoise_only(obs); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
