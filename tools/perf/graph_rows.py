"""Does hipGraph replay help the reverse loop beyond 64 rows?  GRAPH_MAX_ROWS = 64 vs 1024 at B = 128, 256, 512, 768 (eval mode)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
agent = bench.make_agent(torch.device("cuda")); ai = agent.active_inference
for mode in ("eval", "train"):
    (ai.eval() if mode == "eval" else ai.train())
    for B in (128, 256, 512, 768):
        obs = torch.randn(B, bench.O, device="cuda")
        row = []
        for lim in (64, 1024, 64, 1024):
            type(ai).GRAPH_MAX_ROWS = lim
            for _ in range(3): ai.denoise_only(obs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ai.denoise_only(obs)
            torch.cuda.synchronize(); row.append(f"max_rows={lim}: {(time.perf_counter() - t0) / 10 * 1e3:6.2f} ms")
        print(mode, f"B={B}: " + "  ".join(row), flush=True)
