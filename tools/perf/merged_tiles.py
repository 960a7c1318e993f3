"""Merged reverse loop (all belief rows of a step in ONE chain) with the 32-row chain tile kept at 512 / 768 rows - 2 / 3 workgroups of gemm_mid_kernel_o3 per CU from
one launch - against the taller tiles the dispatcher picks (one workgroup per CU, BM = 32 ceil(M / 256)) and against S concurrent 256-row chains on S streams."""
import sys, os, time, ctypes
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
from aid_amd import _native as nat
from aid_amd.streams import chain_streams
from oracle import aid_oracle as orc
from oracle.weights import draw_params, param_specs
O, A, N, D = 17, 6, 50, 512
p = draw_params(param_specs(O, A, O, D, N), 1, orc.schedule_tables(N))
pd = {k: v.cuda().contiguous() for k, v in p.items()}
tab = nat.score_net_table(pd, "latent_score_network.", O, O, D)
coefs = orc.p_sample_coefs(orc.schedule_tables(N)).cuda().contiguous()
def timeit(run, reps=6):
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): run()
    torch.cuda.synchronize(); return (time.time() - t0) / reps
ref = {}
for B in (256, 512, 768):
    obs = torch.randn(B, O, device="cuda"); ws = nat.Workspace("cuda")
    for cap in (0, -1, -2):
        if cap == -2 and B < 768: continue
        nat.lib().aid_mid_debug(0); nat.lib().aid_mid_debug(cap)
        out = nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, offset=0, ws=ws)
        if cap == 0: ref[B] = out.clone()
        else: print(f"   bit-identical to the default tiles: {bool(torch.equal(out, ref[B]))}")
        dt = timeit(lambda: nat.denoise_loop(tab, coefs, obs, N, train=True, seed=1, ws=ws))
        print(f"merged B={B} tile cap {cap}: {dt*1e3:.2f} ms/loop", flush=True)
    nat.lib().aid_mid_debug(0)
    S = B // 256
    if S > 1:
        cur = torch.cuda.current_stream()
        streams = [cur] + list(chain_streams(cur.device))[:S - 1]
        wss = [nat.Workspace("cuda") for _ in range(S)]; parts = list(obs.chunk(S))
        def run():
            for st in streams[1:]: st.wait_stream(cur)
            for st, w, o in zip(streams, wss, parts):
                with torch.cuda.stream(st): nat.denoise_loop(tab, coefs, o, N, train=True, seed=1, ws=w)
            for st in streams[1:]: cur.wait_stream(st)
        print(f"{S} concurrent chains of 256 rows (direct launches): {timeit(run)*1e3:.2f} ms", flush=True)
