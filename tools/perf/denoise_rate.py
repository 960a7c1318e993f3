import sys, os, time, json
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
from aid_amd import _native as nat
from oracle import aid_oracle as orc
from oracle.weights import draw_params, param_specs
O, A, B, N, D = 17, 6, int(os.environ.get("B", 256)), 50, 512
p = draw_params(param_specs(O, A, O, D, N), 1, orc.schedule_tables(N))
pd = {k: v.cuda().contiguous() for k, v in p.items()}
tab = nat.score_net_table(pd, "latent_score_network.", O, O, D)
coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
obs = torch.randn(B, O, device="cuda")
ws = nat.Workspace("cuda")
for train in (True, False):
    for _ in range(2): nat.denoise_loop(tab, coefs, obs, N, train=train, seed=1, ws=ws)
    torch.cuda.synchronize(); t0 = time.time(); R = 5
    for _ in range(R): nat.denoise_loop(tab, coefs, obs, N, train=train, seed=1, ws=ws)
    torch.cuda.synchronize(); dt = (time.time() - t0) / R
    print(f"train={train} B={B}: {dt*1e3:.2f} ms/loop  {N/dt:.1f} denoise steps/s  {46.4e6*B*N/dt/1e12:.2f} TFLOP/s algorithmic")
nat.prof_enable(True); nat.prof_reset()
nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, ws=ws); torch.cuda.synchronize()
r = nat.prof_read(0); print("gemm prof", r, "TF/s", r["flops"]/r["ms"]/1e9)
nat.prof_enable(False)
