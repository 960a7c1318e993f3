import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R + "/active-inference-diffusion_amd"); sys.path.insert(0, R)
import torch, bench
what = sys.argv[1]
agent = bench.make_agent(torch.device("cuda")); ai, cfg = agent.active_inference, agent.config
B = cfg.batch_size
obs = torch.randn(B, bench.O, device="cuda"); lat = torch.randn(B, bench.O, device="cuda")
if what == "denoise":
    ai.denoise_only(obs)
elif what == "efe":
    ai.compute_expected_free_energy_diffusion(lat, horizon=cfg.efe_horizon, backward=True)
elif what == "elbo":
    ai.compute_diffusion_elbo(obs, torch.randn(B, device="cuda"), lat)
elif what == "step_noov":
    agent.overlap_streams = False; agent.train_step()
elif what == "step":
    agent.train_step()
torch.cuda.synchronize(); print("ok", what)
