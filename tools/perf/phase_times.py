import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R+"/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch
import bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda"))
ai, cfg = agent.active_inference, agent.config
for _ in range(3): agent.train_step()
B = cfg.batch_size
batch = agent.replay_buffer.sample(B)
obs, actions, rewards, next_obs, dones = batch["observations"], batch["actions"], batch["rewards"], batch["next_observations"], batch["dones"]
def timed(label, fn, reps=3):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    print(f"{label:34s} {dt*1e3:8.2f} ms"); return out
nrew = timed("sample+normalize", lambda: agent.reward_normalizer.update_and_normalize(agent.replay_buffer.sample(B)["rewards"]))
both = timed("denoise_pair (2B rows, 50 steps)", lambda: ai.denoise_pair(obs, next_obs))
latents = both[:B].contiguous()
timed("denoise_only (B rows, 50 steps)", lambda: ai.denoise_only(next_obs))
def elbo():
    agent.score_optimizer.zero_grad(); return ai.compute_diffusion_elbo(obs, nrew, latents)
timed("elbo step (fwd+bwd+GP)", elbo)
timed("score adamw", lambda: agent.score_optimizer.step())
def efe(skip=False):
    agent.policy_optimizer.zero_grad(); return ai.compute_expected_free_energy_diffusion(latents, horizon=5, backward=True)
timed("efe step (fwd+MINE+bwd)", efe)
def efe_fwd(): return nat.efe_step(ai.policy_network.table(), None, ai.latent_dynamics.table(), ai.reward_table(), ai.value_network.table(), ai.decoder_table(), ai.epistemic_table(), latents, ai._running_mean, 5, 10, 10, train=True, ws=ai._ws)
timed("efe fwd only (incl MINE)", efe_fwd)
def efe_noepi(): return nat.efe_step(ai.policy_network.table(), ai.policy_network.grad_table(), ai.latent_dynamics.table(), ai.reward_table(), ai.value_network.table(), ai.decoder_table(), ai.epistemic_table(), latents, ai._running_mean, 5, 10, 10, train=True, skip_epistemic=True, ws=ai._ws)
timed("efe fwd+bwd WITHOUT MINE", efe_noepi)
nl = ai.denoise_only(next_obs)
timed("value step", lambda: nat.value_step(ai.value_network.table(), ai.value_network.grad_table(), latents, nl, nrew, dones, ws=ai._ws))
timed("dynamics step", lambda: nat.dynamics_step(ai.latent_dynamics.table(), ai.latent_dynamics.grad_table(), latents, actions, nl, ws=ai._ws))
timed("full train_step", lambda: agent.train_step())
