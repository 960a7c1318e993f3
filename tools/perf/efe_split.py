"""Would the EFE rollout + BPTT (K*B = 2 560 rows, ~430 dependent launches, 4.9 ms) gain from running as two concurrent half-K chains on two streams?  Timing experiment
only (the two halves write the same policy-gradient arena: results are not used): one aid_efe_step with K = 10 against two concurrent ones with K = 5."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import numpy as np, torch, bench
from aid_amd import _native as nat
from aid_amd.streams import chain_streams
agent = bench.make_agent(torch.device("cuda")); ai, cfg = agent.active_inference, agent.config
for _ in range(3): agent.train_step()
torch.cuda.synchronize()
B = cfg.batch_size
latents = torch.randn(B, ai.latent_dim, device="cuda")
ws2 = nat.Workspace("cuda")
def efe(K, ws, skip=True):
    return nat.efe_step(ai.policy_network.table(), ai.policy_network.grad_table(), ai.latent_dynamics.table(), ai.reward_table(), ai.value_network.table(),
                        ai.decoder_table(), ai.epistemic_table(), latents, ai._running_mean, cfg.efe_horizon, K, 10, train=ai.training, skip_epistemic=skip,
                        epistemic_weight=cfg.epistemic_weight, pragmatic_weight=cfg.pragmatic_weight, consistency_weight=cfg.consistency_weight,
                        discount=cfg.discount_factor, temperature=float(cfg.preference_temperature), seed=ai.noise.seed, offset=ai.noise.next_offset(), ws=ws)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
main = torch.cuda.current_stream(); st = chain_streams("cuda")[0]
def two_halves():
    st.wait_stream(main)
    with torch.cuda.stream(st): efe(5, ws2)
    efe(5, ai._ws)
    main.wait_stream(st)
print(f"rollout + BPTT, K = 10 on one stream: {timed(lambda: efe(10, ai._ws)):.2f} ms; K = 5 alone: {timed(lambda: efe(5, ai._ws)):.2f} ms; two concurrent K = 5 halves: {timed(two_halves):.2f} ms")
