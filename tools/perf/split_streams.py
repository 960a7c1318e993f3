"""Reverse loop at B rows as ONE chain vs S independent chains of B/S rows on S streams (rows are independent: same results up to the
GEMM kernel the dispatcher picks per row count).  Latency-bound chains overlap each other's stalls."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
from aid_amd import _native as nat
from oracle import aid_oracle as orc
from oracle.weights import draw_params, param_specs
O, A, N, D = 17, 6, 50, 512
p = draw_params(param_specs(O, A, O, D, N), 1, orc.schedule_tables(N))
pd = {k: v.cuda().contiguous() for k, v in p.items()}
tab = nat.score_net_table(pd, "latent_score_network.", O, O, D)
coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
for B in (256, 768):
    obs = torch.randn(B, O, device="cuda")
    for S in (1, 2, 3, 4):
        if B % S: continue
        streams = [torch.cuda.Stream() for _ in range(S)]
        wss = [nat.Workspace("cuda") for _ in range(S)]
        parts = list(obs.chunk(S))
        def run():
            cur = torch.cuda.current_stream()
            for st in streams: st.wait_stream(cur)
            outs = []
            for st, ws, o in zip(streams, wss, parts):
                with torch.cuda.stream(st):
                    outs.append(nat.denoise_loop(tab, coefs, o, N, train=True, seed=1, ws=ws))
            for st in streams: cur.wait_stream(st)
            return outs
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.time(); R_ = 5
        for _ in range(R_): run()
        torch.cuda.synchronize(); dt = (time.time() - t0) / R_
        print(f"B={B} as {S} x {B//S} rows: {dt*1e3:.2f} ms/loop  {N/dt:.0f} denoise steps/s", flush=True)
