"""Reverse-loop latency at collector / act() batch sizes (eval mode, hipGraph replay as the product runs it): the <= 128-row loop on the split-K
pipeline (gemm_w4: 32 x 16 tiles, waves split K, slabs folded by the following adaLN: every layer spread over 16..128 workgroups per 32 rows) against the
weight-streaming skinny / panel kernels."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
ai.eval()
for mode, name in ((-2, "skinny / panel kernels"), (-1, "gemm_w4 pipeline")):
    nat.lib().aid_denoise_debug(mode)
    ai.__dict__.get("_denoise_graphs", {}).clear(); torch.cuda.synchronize()
    for B in (1, 4, 8, 16, 32, 64, 96, 128):
        obs = torch.randn(B, bench.O, device="cuda")
        for _ in range(3): ai.denoise_only(obs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ai.denoise_only(obs)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        print(f"{name:24s} B={B:4d}: 50-step loop {t*1e3:7.2f} ms = {t*1e6/50:6.1f} us per denoise step", flush=True)
nat.lib().aid_denoise_debug(-1)
