"""Reverse-loop latency at collector / act() batch sizes (eval mode, as the collector runs it)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
ai.eval()
for B in (1, 8, 32, 256):
    obs = torch.randn(B, bench.O, device="cuda")
    for _ in range(3): ai.denoise_only(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ai.denoise_only(obs)
    t_host = (time.perf_counter() - t0) / 10          # enqueue time
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"B={B:4d}: 50-step loop {t*1e3:7.2f} ms ({50/t:7.0f} denoise steps/s), host enqueue {t_host*1e3:6.2f} ms")
