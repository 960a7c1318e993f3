"""act() / collector batches of <= 16 rows: the reverse loop with the adaLN launches folded into their consumer GEMMs (gemm_w4ln, aid_denoise_debug -10) against
separate adaLN kernels (-11); eval mode, hipGraph replay as the product runs it."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
ai.eval()
for mode, name in ((-11, "separate adaLN kernels (32 launches)"), (-10, "adaLN folded into the GEMMs (20)")):
    nat.lib().aid_denoise_debug(mode)
    ai.__dict__.get("_denoise_graphs", {}).clear(); torch.cuda.synchronize()
    for B in (1, 4, 8, 16):
        obs = torch.randn(B, bench.O, device="cuda")
        for _ in range(3): ai.denoise_only(obs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ai.denoise_only(obs)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        print(f"{name:40s} B={B:3d}: 50-step loop {t*1e3:6.2f} ms = {t*1e6/50:6.1f} us per denoise step", flush=True)
nat.lib().aid_denoise_debug(-10)
