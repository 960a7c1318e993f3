import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
from aid_amd import _native as nat
from oracle import aid_oracle as orc
from oracle.weights import draw_params, param_specs
O, A, B, N, D = 17, 6, 256, 50, 512
p = draw_params(param_specs(O, A, O, D, N), 1, orc.schedule_tables(N))
pd = {k: v.cuda().contiguous() for k, v in p.items()}
tab = nat.score_net_table(pd, "latent_score_network.", O, O, D)
coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
obs = torch.randn(B, O, device="cuda")
ws = nat.Workspace("cuda")
def run(label):
    for _ in range(2): nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, ws=ws)
    torch.cuda.synchronize(); t0 = time.time(); R = 5
    for _ in range(R): nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, ws=ws)
    torch.cuda.synchronize(); dt = (time.time() - t0) / R
    print(f"{label}: {dt*1e3:.2f} ms/loop  {N/dt:.1f} denoise steps/s")
run("default stream")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    run("side stream")
    # graph capture of one loop
    g = torch.cuda.CUDAGraph()
    zz = torch.empty(B, O, device="cuda")
    nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, ws=ws)
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g, stream=st):
            nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, ws=ws)
        for _ in range(2): g.replay()
        torch.cuda.synchronize(); t0=time.time()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); dt=(time.time()-t0)/5
        print(f"hipGraph replay: {dt*1e3:.2f} ms/loop  {N/dt:.1f} denoise steps/s")
    except Exception as e:
        print("graph capture failed:", e)
