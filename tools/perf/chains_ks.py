"""K slices of the 512-deep x Wov^T product in the 256-row reverse loop (aid_denoise_debug -9 default 4 / -7 two / -8 one): one, two and three concurrent 256-row
chains as hipGraph replays, in situ.  usage: python tools/perf/chains_ks.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench
from aid_amd import _native as nat
for mode, what in ((-9, "4 slices (default)"), (-7, "2 slices"), (-8, "1 slice")):
    nat.lib().aid_denoise_debug(mode)
    agent = bench.make_agent(torch.device("cuda"))
    ai = agent.active_inference
    ai.concurrent_belief_loops = True
    obs = torch.randn(256, bench.O, device="cuda")
    res = []
    for k in (1, 2, 3):
        fn = (lambda: ai._denoise_chains([obs], [None], None, bench.NSTEPS, True, [torch.cuda.current_stream()])) if k == 1 else (lambda: ai.denoise_many([obs] * k))
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"Wov product in {what:20s}: one chain {res[0]:6.2f} ms, two {res[1]:6.2f} ms, three {res[2]:6.2f} ms", flush=True)
    del agent, ai; torch.cuda.empty_cache()
nat.lib().aid_denoise_debug(-9)
