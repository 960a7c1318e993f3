"""One phase of a cfg-2 train_step, in isolation, a few times: the program to put behind `rocprofv3 --kernel-trace --stats --` (or a --pmc pass)
so that a phase's kernels can be read off one summary (profiles/rNN_phase_<phase>_kernel_stats.csv; bench.py's `roofline_phases` times the same
calls with HIP events).  usage: python3 tools/perf/phase_profile.py <elbo|rollout_bptt|efe|chain1|chain1_direct|chains3|chains2|value_dynamics|optimiser>[_cold] [reps]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch

import bench
from aid_amd import _native as nat

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if what.startswith("pixel_epistemic"):        # SURVEY 8(f)-1 at the reduced size bench.py's pixel_epistemic_f1 reports (64 belief rows, S = 2)
    import aid_amd
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=64, efe_horizon=1, latent_dim=128, hidden_dim=512, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=4)
    class Env4: observation_space, action_space = bench._Box((3, 3, 84, 84)), bench._Box((6,))
    agent = aid_amd.DiffusionPixelAgent(Env4(), cfg, aid_amd.TrainingConfig(buffer_size=64), aid_amd.PixelObservationConfig())
    for _ in range(reps):
        r = bench.pixel_epistemic_probe(agent)
    print("ok", what, {k: (round(v["ms"], 2), round(v["achieved"], 1)) for k, v in r.items() if k != "config"})
    sys.exit(0)
agent = bench.make_agent(torch.device("cuda"))
ai, cfg = agent.active_inference, agent.config
B, N = cfg.batch_size, cfg.diffusion.num_diffusion_steps
for _ in range(0 if what.endswith("_cold") else 2):
    agent.train_step()
agent.invalidate_prefetch(); agent._lookahead = None
torch.cuda.synchronize()
batch = agent.replay_buffer.sample(B)
obs, actions, next_obs, dones = batch["observations"], batch["actions"], batch["next_observations"], batch["dones"]
nrew = torch.randn(B, device="cuda")
latents = torch.randn(B, bench.O, device="cuda"); next_latents = torch.randn(B, bench.O, device="cuda")
train = ai.latent_score_network.training


def efe(skip):
    agent.policy_optimizer.zero_grad()
    nat.efe_step(ai.policy_network.table(), ai.policy_network.grad_table(), ai.latent_dynamics.table(), ai.reward_table(), ai.value_network.table(),
                 ai.decoder_table(), ai.epistemic_table(), latents, ai._running_mean, cfg.efe_horizon, 10, 10, train=ai.training, skip_epistemic=skip,
                 seed=ai.noise.seed, offset=ai.noise.next_offset(), ws=ai._ws)


def elbo():
    agent.score_optimizer.zero_grad(); ai.compute_diffusion_elbo(obs, nrew, latents)


def value_dyn():
    agent.value_optimizer.zero_grad(); agent.dynamics_optimizer.zero_grad()
    nat.value_step(ai.value_network.table(), ai.value_network.grad_table(), latents, next_latents, nrew, dones, cfg.discount_factor, 0.95, 5, ws=ai._ws)
    nat.dynamics_step(ai.latent_dynamics.table(), ai.latent_dynamics.grad_table(), latents, actions, next_latents, ws=ai._ws)


ai.concurrent_belief_loops = True
fn = {"elbo": elbo, "rollout_bptt": lambda: efe(True), "efe": lambda: efe(False),
      "chain1": lambda: ai._denoise_chains([obs], [None], None, N, train, [torch.cuda.current_stream()]),
      "chain1_direct": lambda: ai.denoise_only(obs, ws=ai._ws),          # the same 256-row loop launched kernel by kernel (PMC passes: no graph replay)
      "chains3": lambda: ai.denoise_many([next_obs, obs, next_obs]),
      "chains2": lambda: ai.denoise_many([next_obs, obs]),               # the two LIVE belief updates of a step (dead next_obs belief elided)
      "merged3": lambda: (setattr(ai, "concurrent_belief_loops", False), ai.denoise_many([next_obs, obs, next_obs]))[1],      # ONE 768-row loop
      "value_dynamics": value_dyn, "optimiser": lambda: agent.score_optimizer.step(reduce=False)}[what.replace("_cold", "")]
dump = os.environ.get("AID_PHASE_DUMP")        # per-launch HIP-event timings of ONE call (tools/perf/launch_table.py reads the CSV)
if dump:
    fn(); torch.cuda.synchronize()
    nat.prof_enable(True); nat.prof_reset(); fn(); torch.cuda.synchronize(); nat.prof_dump(dump); nat.prof_enable(False); nat.prof_reset()
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("ok", what, reps, f"{1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call")
