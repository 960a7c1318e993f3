"""Soak: N back-to-back cfg-2 train_steps (device Philox): metrics stay finite, memory does not grow, prefetch keeps hitting."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
agent = bench.make_agent(torch.device("cuda"))
agent.train_epistemic_every = 5
torch.cuda.synchronize(); t0 = time.perf_counter(); mem0 = None
for i in range(n):
    m = agent.train_step()
    assert all(np.isfinite(v) for v in m.values()), (i, m)
    if i == 20: mem0 = torch.cuda.memory_allocated()
    if i % 50 == 0: print(i, {k: round(m[k], 4) for k in ("elbo", "score_matching_loss", "policy_loss", "value_loss", "dynamics_loss", "grad_penalty")}, flush=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{n} steps, {dt/n*1e3:.1f} ms/step, memory growth since step 20: {(torch.cuda.memory_allocated()-mem0)/1e6:.1f} MB, peak {torch.cuda.max_memory_allocated()/1e9:.2f} GB, prefetch backoff {agent._prefetch_backoff}")
