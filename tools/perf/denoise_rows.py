"""50-step reverse loop alone at several row counts (the merged belief updates of a train_step are 3 B rows): ms per loop and algorithmic TFLOP/s.
  python tools/perf/denoise_rows.py 768 3072 ..."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch
import bench
agent = bench.make_agent(torch.device("cuda"))
ai = agent.active_inference
for B in [int(a) for a in sys.argv[1:]] or [256, 768, 1536, 3072]:
    obs = torch.randn(B, 17, device="cuda")
    for _ in range(2): ai.denoise_only(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ai.denoise_only(obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"rows {B:5d}: {dt*1e3:8.2f} ms per 50-step loop, {bench.ALGO_FLOP_PER_DENOISE_ROW_STEP * 50 * B / dt / 1e12:6.1f} TFLOP/s", flush=True)
