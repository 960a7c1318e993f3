"""Do the EFE phase (main stream) and a reverse-diffusion loop (side stream) actually overlap?"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
from aid_amd import _native as nat
agent = bench.make_agent(torch.device("cuda")); ai, cfg = agent.active_inference, agent.config
for _ in range(3): agent.train_step()
B = cfg.batch_size
batch = agent.replay_buffer.sample(B); obs, next_obs = batch["observations"], batch["next_observations"]
latents = ai.denoise_pair(obs, next_obs)[:B].contiguous()
side = torch.cuda.Stream(priority=-1 if "prio" in sys.argv else 0); ws2 = nat.Workspace(torch.device("cuda"))
main = torch.cuda.current_stream()
def efe(): ai.compute_expected_free_energy_diffusion(latents, horizon=cfg.efe_horizon, backward=True)
def dn(force=0):
    if force: nat.gemm_force_kernel(force)
    ai.denoise_only(next_obs, ws=ws2)
    if force: nat.gemm_force_kernel(0)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def both(force=0):
    ev = torch.cuda.Event(); ev.record(main)
    efe()
    side.wait_event(ev)
    with torch.cuda.stream(side): dn(force)
    main.wait_stream(side)
def both_rev(force=0):
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side): dn(force)
    efe()
    main.wait_stream(side)
print(f"efe alone {t(efe):.2f} ms   denoise alone {t(dn):.2f} ms   denoise(tile64) alone {t(lambda: dn(4)):.2f} ms")
print(f"both (efe first)      {t(both):.2f} ms   with side splitk {t(lambda: both(6)):.2f} ms   with side low-LDS panel {t(lambda: both(512)):.2f} ms")
print(f"denoise(splitk) alone {t(lambda: dn(6)):.2f} ms   denoise(low-LDS panel) alone {t(lambda: dn(512)):.2f} ms")
R3 = torch.cat([next_obs, obs, next_obs])
def dn3(force=0):
    if force: nat.gemm_force_kernel(force)
    ai.denoise_many([next_obs, obs, next_obs], ws=ws2)
    if force: nat.gemm_force_kernel(0)
def both3(force=0):
    ev = torch.cuda.Event(); ev.record(main)
    efe()
    side.wait_event(ev)
    with torch.cuda.stream(side): dn3(force)
    main.wait_stream(side)
print(f"768-row loop alone {t(dn3):.2f} ms (low-LDS {t(lambda: dn3(512)):.2f} ms)   with efe {t(both3):.2f} ms   low-LDS with efe {t(lambda: both3(512)):.2f} ms")
# host-side enqueue cost
t0 = time.perf_counter(); efe(); te = time.perf_counter() - t0; torch.cuda.synchronize()
t0 = time.perf_counter(); dn(); td = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"host enqueue time: efe {te*1e3:.2f} ms, denoise {td*1e3:.2f} ms")
