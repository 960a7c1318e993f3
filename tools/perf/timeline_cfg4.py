"""Where the streams of a cfg-4 (pixels, B = 256) train_step are at which time: event marks recorded by DiffusionPixelAgent (agent.timeline), ms from the step's
start, averaged over back-to-back steps.  EARLY=0: the paired belief loop after both encoder passes."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench, aid_amd
cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=256, efe_horizon=bench.HORIZON, latent_dim=128, hidden_dim=bench.D, pixel_observation=True)
cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=bench.NSTEPS)
class Env4: observation_space, action_space = bench._Box((3, 3, 84, 84)), bench._Box((6,))
agent = aid_amd.DiffusionPixelAgent(Env4(), cfg, aid_amd.TrainingConfig(buffer_size=512), aid_amd.PixelObservationConfig())
rng = np.random.default_rng(0); n = 512
frames = rng.integers(0, 256, (n, 3, 3, 84, 84), dtype=np.uint8)
agent.replay_buffer.add_batch(frames, rng.uniform(-1, 1, (n, 6)).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.roll(frames, -1, axis=0), rng.random(n) < 0.01)
agent.early_belief_chains = os.environ.get("EARLY", "1") == "1"
agent.elbo_on_side_stream = os.environ.get("SIDE_ELBO", "1") == "1"
agent.stagger_next_obs_chain = os.environ.get("STAGGER", "0") == "1"
for _ in range(4): agent.train_step()
n = 8; steps = []
for _ in range(n + 1):
    agent.timeline = []
    agent.train_step()
    steps.append(agent.timeline)
agent.timeline = None
torch.cuda.synchronize()
acc = collections.OrderedDict()
for i in range(n):
    t0, h0 = steps[i][0][1], steps[i][0][2]
    for label, ev, ht in steps[i][1:] + [("start of the next step",) + tuple(steps[i + 1][0][1:])]:
        a = acc.setdefault(label, [0.0, 0.0]); a[0] += t0.elapsed_time(ev) / n; a[1] += 1e3 * (ht - h0) / n
for label, (ms, hms) in acc.items():
    print(f"{ms:8.2f} ms on the GPU   (enqueued by the host at {hms:7.2f} ms)  {label}")
