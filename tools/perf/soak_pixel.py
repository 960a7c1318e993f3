"""Soak: N back-to-back cfg-4 (pixels, B = 256) train_steps on device Philox: metrics stay finite, memory does not grow, two replicas started from the same
state and seed stay bit-identical (the schedule - side-stream ELBO next to the EFE, early chains, score step behind chain b - has no race that a long run would show)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench, aid_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(os.environ.get("B", 256))
def make():
    torch.manual_seed(0)
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=bench.HORIZON, latent_dim=128, hidden_dim=bench.D, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=bench.NSTEPS)
    class Env4: observation_space, action_space = bench._Box((3, 3, 84, 84)), bench._Box((6,))
    agent = aid_amd.DiffusionPixelAgent(Env4(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
    rng = np.random.default_rng(0); m = 512
    frames = rng.integers(0, 256, (m, 3, 3, 84, 84), dtype=np.uint8)
    agent.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (m, 6)).astype(np.float32), rng.standard_normal(m).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(m) < 0.01)
    np.random.seed(0)
    return agent
def run(agent, steps):
    hist = []
    for i in range(steps):
        m = agent.train_step()
        assert all(np.isfinite(v) for v in m.values()), (i, m)
        hist.append(m)
    torch.cuda.synchronize()
    return hist
a = make()
t0 = time.perf_counter(); ha = run(a, 20); mem0 = torch.cuda.memory_allocated(); ha += run(a, n - 20); dt = time.perf_counter() - t0
print(f"{n} pixel steps, {dt / n * 1e3:.1f} ms/step, memory growth since step 20: {(torch.cuda.memory_allocated() - mem0) / 1e6:.1f} MB, peak {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
print({k: round(ha[-1][k], 5) for k in ("elbo", "contrastive_loss", "policy_loss", "value_loss", "dynamics_loss")})
sd_a = {k: v.clone() for k, v in {**a.active_inference.state_dict(), **{"encoder." + k: v for k, v in a.encoder.state_dict().items()}}.items()}
del a; torch.cuda.empty_cache()
k = min(n, int(os.environ.get("REPLICA_STEPS", 30)))
b = make(); hb = run(b, k)
same = all(ha[i] == hb[i] for i in range(k))
print(f"second replica, same seed: metrics of the first {k} steps bit-identical: {same}")
if not same:
    for i in range(k):
        if ha[i] != hb[i]:
            print("first difference at step", i, {kk: (ha[i][kk], hb[i][kk]) for kk in ha[i] if ha[i][kk] != hb[i][kk]}); break
