"""GPU: rows of SURVEY 8(a) that round 1 covered only indirectly, each against the reference's own numbers (tests/golden) or the oracle:
A15 FreeEnergyComputation.compute_loss, A0' RunningMeanStd (aid_reward_normalize), A10 compute_epistemic_value as a public method."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import aid_oracle as orc
from oracle.noise import NoiseStream
from oracle.weights import draw_params, param_specs
from parity_noise import ParityNoise, dev


def _golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"state_{tag}.npz"))
    O, A, L, D, N = (int(g[k]) for k in ("O", "A", "L", "D", "N"))
    specs = param_specs(O, A, L, D, N)
    return g, draw_params(specs, int(g["weights_key"]), orc.schedule_tables(N))


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_free_energy_compute_loss_golden(golden_dir, tag):
    """core/free_energy.py:30-91 at current_time=0.5, score network in eval mode: the reference's value (golden `free_energy_t05`) and
    every entry of the info dict against the oracle."""
    from aid_amd.core.free_energy import FreeEnergyComputation
    from aid_amd.models import LatentScoreNetwork
    g, p = _golden(golden_dir, tag)
    L, D = int(g["L"]), int(g["D"])
    net = LatentScoreNetwork(latent_dim=L, observation_dim=L, hidden_dim=D).cuda()
    pre = "latent_score_network."
    net.load_state_dict({k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}, strict=True)
    net.eval()
    fe = FreeEnergyComputation().cuda()
    z, obs = dev(g["z"]), dev(g["obs"])
    loss, info = fe.compute_loss(z, obs, None, net, current_time=0.5)
    np.testing.assert_allclose(float(loss), float(g["free_energy_t05"]), rtol=1e-4)      # losses rtol 1e-4
    ref, rinfo = orc.free_energy_loss(p, torch.from_numpy(g["z"]), torch.from_numpy(g["obs"]), torch.zeros(()), 0.5)
    np.testing.assert_allclose(float(loss), float(ref), rtol=1e-4)
    for k, v in rinfo.items():
        if k in info:
            np.testing.assert_allclose(float(info[k]), float(v), rtol=1e-4, atol=1e-7, err_msg=k)
    # backward=True: dF/d(log_precision) accumulated into .grad, the direct dF/d(states) and dF/d(score) returned - against the oracle's autograd
    lp = torch.zeros((), requires_grad=True)
    zt = torch.from_numpy(g["z"]).clone().requires_grad_(True)
    ref2, _ = orc.free_energy_loss(p, zt, torch.from_numpy(g["obs"]), lp, 0.5)
    (g_lp,) = torch.autograd.grad(ref2, [lp], retain_graph=True)
    direct = 0.5 * torch.sum(zt ** 2, dim=-1).mean() + 0.5 * torch.exp(lp.detach()) * torch.sum((torch.from_numpy(g["obs"]) - zt) ** 2, dim=-1).mean()
    (g_z,) = torch.autograd.grad(direct, [zt])
    loss2, info2 = fe.compute_loss(z, obs, None, net, current_time=0.5, backward=True)
    np.testing.assert_allclose(float(fe.log_precision.grad), float(g_lp), rtol=1e-4)
    np.testing.assert_allclose(info2["d_states_direct"].cpu().numpy(), g_z.numpy(), rtol=1e-5, atol=1e-7)
    s_ref = orc.score_forward(p, torch.from_numpy(g["z"]), torch.full((z.shape[0],), 0.5), torch.from_numpy(g["obs"]))
    np.testing.assert_allclose(info2["d_score"].cpu().numpy(), (0.02 * s_ref / z.shape[0]).numpy(), rtol=1e-3, atol=1e-9)
    # ... and, through the score network's reverse pass, the FULL gradient: dF/d(states), dF/d(observations) and every score-network parameter
    pg = {k: v.clone().requires_grad_(k.startswith(pre) and orc.is_param(k) and v.dtype.is_floating_point) for k, v in p.items()}
    zt2 = torch.from_numpy(g["z"]).clone().requires_grad_(True); ot2 = torch.from_numpy(g["obs"]).clone().requires_grad_(True)
    ref3, _ = orc.free_energy_loss(pg, zt2, ot2, torch.zeros(()), 0.5)
    names = [k for k in pg if pg[k].requires_grad]
    grads = torch.autograd.grad(ref3, [zt2, ot2] + [pg[k] for k in names], allow_unused=True)
    np.testing.assert_allclose(info2["d_states"].cpu().numpy(), grads[0].numpy(), rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(info2["d_observations"].cpu().numpy(), grads[1].numpy(), rtol=1e-3, atol=1e-7)
    got = {pre + n_: q.grad for n_, q in net.named_parameters()}
    checked = 0
    for k, r in zip(names, grads[2:]):
        if r is None or float(r.abs().max()) == 0.0:
            continue
        err = float((got[k].cpu() - r).norm() / r.norm())
        assert err < 1e-3, (k, err)
        checked += 1
    assert checked > 40, checked
    for q in net.parameters():
        q.grad = None
    fe.log_precision.grad = None
    fe.compute_loss(z, obs, None, net, current_time=0.5, backward=True); fe.compute_loss(z, obs, None, net, current_time=0.5, backward=True)      # accumulates, like autograd's .grad
    np.testing.assert_allclose(float(fe.log_precision.grad), 2 * float(g_lp), rtol=1e-4)
    # update_precision (:93-101): log-precision nudged by 0.01 * clamp(err, -1, 1), clamped to [-3, 3]
    fe.update_precision(torch.tensor(5.0), torch.tensor(1.0)); np.testing.assert_allclose(float(fe.log_precision), 0.01, rtol=1e-6)
    fe.update_precision(torch.tensor(0.25), torch.tensor(0.5)); np.testing.assert_allclose(float(fe.log_precision), 0.0075, rtol=1e-5)


def test_reward_normalizer_golden_sequence(golden_dir):
    """agents/base_agent.py:24-52: four update()s of the reference's RunningMeanStd (golden `rms_seq` = its mean, var, count after each) and
    the normalised batch.  The reference takes the batch moments with numpy in float32 and merges in float64; the kernel takes them in
    float64 throughout, hence 1e-6 rather than 1e-12 on mean / var (count is exact)."""
    from aid_amd.agents.base_agent import RunningMeanStd
    m = np.load(os.path.join(golden_dir, "misc.npz"))
    rms = RunningMeanStd(shape=(), device="cuda")
    ref = orc.RunningMeanStd()
    for i in range(4):
        x = NoiseStream(90 + i).randn(16) * (i + 1) + i
        out = rms.update_and_normalize(dev(x))
        ref.update(x)
        mean, var, count = m["rms_seq"][i]
        np.testing.assert_allclose(rms.mean, mean, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(rms.var, var, rtol=1e-6)
        assert rms.count == count
        np.testing.assert_allclose(out.cpu().numpy(), ref.normalize(x), rtol=1e-5, atol=1e-6)
    # data parallel form: statistics from the gathered global batch, the local slice normalised
    rms2 = RunningMeanStd(shape=(), device="cuda"); ref2 = orc.RunningMeanStd()
    x = NoiseStream(99).randn(32)
    out = rms2.update_and_normalize(dev(x[8:16]), dev(x)); ref2.update(x)
    np.testing.assert_allclose([rms2.mean, rms2.var, rms2.count], [ref2.mean, ref2.var, ref2.count], rtol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), ref2.normalize(x[8:16]), rtol=1e-5, atol=1e-6)


class _Box:
    def __init__(self, shape):
        self.shape = shape


@pytest.mark.parametrize("B,S,D", [(8, 5, 64), (256, 10, 512)])
def test_compute_epistemic_value_vs_oracle(B, S, D):
    """core/active_inference.py:398-418 as a public method: MI lower bound, joint / marginal terms, the running-mean EMA over two
    successive calls (it is mutated under no_grad), with a row- and column-dependent log-variance."""
    import aid_amd
    O, A, N = 17, 6, 4
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=2, latent_dim=O, hidden_dim=D)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)

    class Env:
        observation_space, action_space = _Box((O,)), _Box((A,))
    agent = aid_amd.DiffusionStateAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=16))
    ai = agent.active_inference
    p = draw_params(param_specs(O, A, O, D, N), 19, orc.schedule_tables(N))
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    ns = NoiseStream(5)
    mean = ns.randn(B, O); logvar = (np.log(0.1) + 0.3 * ns.randn(B, O)).astype(np.float32)
    ai.noise = ParityNoise(321)
    ref_noise = NoiseStream(321)
    rm = float(p["epistemic_estimator.running_mean"])
    for call in range(2):
        val, met = ai.compute_epistemic_value(dev(mean), dev(logvar), num_samples=S)
        clamped, mi, joint, tlog, rm = orc.epistemic_value(p, torch.from_numpy(mean), torch.from_numpy(logvar), rm, ref_noise, S, True)
        assert val.shape == (B,) and float(val.min()) == float(val.max())
        # the MI is a difference of two O(1) batch means: absolute tolerance on the terms, SURVEY's 1e-4 relative on each of them
        np.testing.assert_allclose(met["epistemic/joint_term"], float(joint), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(met["epistemic/marginal_term"], float(tlog), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(met["epistemic/mi_estimate"], float(mi), rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(float(val[0]), float(clamped), rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(met["epistemic/running_mean"], rm, rtol=1e-5)
    np.testing.assert_allclose(float(ai.state_dict()["epistemic_estimator.running_mean"]), rm, rtol=1e-5)
    # eval mode (no dropout draws) runs too and is deterministic given the noise
    ai.eval(); ai.noise = ParityNoise(7)
    v1, _ = ai.compute_epistemic_value(dev(mean), dev(logvar), num_samples=S)
    assert np.isfinite(float(v1[0]))


def test_round3_surface_reference_goldens(golden_dir):
    """Stand-alone methods that raised until round 3, each against the reference's own output (tests/golden/r3_surface.npz):
    compute_lambda_returns(exclude_immediate_rewards=True) bit-exact, _compute_gradient_penalty as a value (train / eval, both time
    branches), generate_latent_trajectory(deterministic=True), and the state-mode update_belief_via_diffusion with store_trajectory."""
    import aid_amd
    from aid_amd import _native as nat
    g = np.load(os.path.join(golden_dir, "r3_surface.npz"))
    O, A, D, N, B = (int(x) for x in g["dims"])
    p = draw_params(param_specs(O, A, O, D, N), int(g["weights_key"]), orc.schedule_tables(N))
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=2, latent_dim=O, hidden_dim=D)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)
    ai = aid_amd.core.DiffusionActiveInference(O, A, O, cfg)
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    for tag in "abc":
        r, nv, d, ns_ = dev(g[f"lam_{tag}_r"]), dev(g[f"lam_{tag}_nv"]), dev(g[f"lam_{tag}_d"].astype(np.float32)), int(g[f"lam_{tag}_n_steps"])
        out = ai.compute_lambda_returns(r, torch.zeros_like(r), nv, d, lambda_=0.95, n_steps=ns_, exclude_immediate_rewards=True)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"lam_{tag}_excl"])
        out = ai.compute_lambda_returns(r, torch.zeros_like(r), nv, d, lambda_=0.95, n_steps=ns_)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"lam_{tag}_incl"])
    z, obs = dev(g["z"]), dev(g["obs"])
    for tag, train, tt in (("train_cont", True, dev(g["t_cont"])), ("eval_cont", False, dev(g["t_cont"])), ("eval_disc", False, torch.full((B,), 3.0, device="cuda"))):
        ai.train(train)
        mask = dev(NoiseStream(int(g["gp_key"])).dropout_mask((B, D), 0.1)) if train else None
        pen = ai._compute_gradient_penalty(z, tt, obs, drop_mask=mask)
        np.testing.assert_allclose(float(pen), float(g[f"gp_{tag}"]), rtol=1e-4, err_msg=tag)       # losses rtol 1e-4
    # the same number aid_elbo_step reports as its grad_penalty metric is covered by test_elbo_golden_metrics; here also g itself vs autograd
    pen, gvec = nat.gradient_penalty(ai.latent_score_network.table(), z, dev(g["t_cont"]), obs, True, None)
    zz = torch.from_numpy(g["z"]).requires_grad_(True)
    (g_ref,) = torch.autograd.grad(orc.score_forward(p, zz, torch.from_numpy(g["t_cont"]), torch.from_numpy(g["obs"])).sum(), zz)
    np.testing.assert_allclose(gvec.cpu().numpy(), g_ref.numpy(), rtol=2e-4, atol=2e-6)
    # deterministic trajectory, eval mode: the only draw is the initial latent
    ai.eval()
    z0 = dev(NoiseStream(int(g["det_key"])).randn(B, O))
    traj = ai.latent_diffusion.generate_latent_trajectory(ai.latent_score_network, B, obs, deterministic=True, noise={"z_init": z0})
    np.testing.assert_allclose(torch.stack(traj).cpu().numpy(), g["det_traj"], rtol=1e-4, atol=2e-6)
    # store_trajectory in state mode: N+1 iterates, last one is the belief; same draws without it -> same belief through the fused entry
    ai.train(); ai.store_trajectory = True
    gs = np.load(os.path.join(golden_dir, "state_hc.npz"))
    p2 = draw_params(param_specs(17, 6, 17, int(gs["D"]), int(gs["N"])), int(gs["weights_key"]), orc.schedule_tables(int(gs["N"])))
    cfg2 = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=int(gs["B"]), efe_horizon=2, latent_dim=17, hidden_dim=int(gs["D"]))
    cfg2.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=int(gs["N"]))
    ai2 = aid_amd.core.DiffusionActiveInference(17, 6, 17, cfg2)
    sd0 = ai2.state_dict()
    ai2.load_state_dict({**p2, **{k: p2[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    ai2.store_trajectory = True
    ai2.noise = ParityNoise(int(gs["belief_key"]))
    out = ai2.update_belief_via_diffusion(dev(gs["obs"]))
    assert len(ai2.latent_trajectory) == int(gs["N"]) + 1 == out["trajectory_length"]
    np.testing.assert_allclose(torch.stack(ai2.latent_trajectory).cpu().numpy(), gs["belief_traj"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(float(out["reconstruction_error"]), float(gs["belief_rec"]), rtol=1e-4)
    np.testing.assert_allclose(out["latent_std"].cpu().numpy(), gs["belief_std"], rtol=1e-4, atol=1e-6)


def test_pixel_update_belief_reference_golden(golden_dir):
    """Pixel-mode update_belief_via_diffusion (active_inference.py:256-312, :292-295): the belief over encoder features, its batch
    statistics, the feature-decoder reconstruction error and the whole N+1-step trajectory against the reference's own run, train and
    eval mode (round 2 returned a zero reconstruction_error here)."""
    import aid_amd
    from oracle.weights import draw_feature_decoder
    g = np.load(os.path.join(golden_dir, "r3_surface.npz"))
    L, A, D, N, B = (int(x) for x in g["pix_dims"])
    pp = draw_params(param_specs(L, A, L, D, N), int(g["pix_weights_key"]), orc.schedule_tables(N))
    pp = {k: v for k, v in pp.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}
    pp.update(draw_feature_decoder(L, D, int(g["pix_fdec_key"])))
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=1, latent_dim=L, hidden_dim=D, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)
    ai = aid_amd.core.DiffusionActiveInference(L, A, L, cfg, pixel_shape=None)
    ai.load_state_dict(pp, strict=True)
    feats = dev(g["pix_feats"])
    for tag, train in (("train", True), ("eval", False)):
        ai.train(train)
        for keep in (True, False):
            ai.store_trajectory = keep
            ai.noise = ParityNoise(int(g[f"pix_{tag}_key"]), pixel=True)
            out = ai.update_belief_via_diffusion(feats)
            np.testing.assert_allclose(out["latent"].cpu().numpy(), g[f"pix_{tag}_latent"], rtol=1e-4, atol=2e-6)
            np.testing.assert_allclose(float(out["reconstruction_error"]), float(g[f"pix_{tag}_rec"]), rtol=1e-4)
            np.testing.assert_allclose(out["latent_mean"].cpu().numpy(), g[f"pix_{tag}_mean"], rtol=1e-4, atol=2e-6)
            np.testing.assert_allclose(out["latent_std"].cpu().numpy(), g[f"pix_{tag}_std"], rtol=1e-4, atol=2e-6)
            assert out["trajectory_length"] == N + 1 and len(ai.latent_trajectory) == (N + 1 if keep else 1)
            if keep:
                np.testing.assert_allclose(torch.stack(ai.latent_trajectory).cpu().numpy(), g[f"pix_{tag}_traj"], rtol=1e-4, atol=2e-6)
