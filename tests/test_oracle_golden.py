"""CPU: the oracle (oracle/aid_oracle.py) against the committed golden vectors that
oracle/gen_golden.py produced by running the reference itself.  No reference, no GPU needed."""
import os

import numpy as np
import pytest
import torch

from oracle import aid_oracle as orc
from oracle.noise import NoiseStream
from oracle.weights import draw_params, param_specs

T = torch.from_numpy


def _load(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"state_{tag}.npz"))
    O, A, L, D, N = (int(g[k]) for k in ("O", "A", "L", "D", "N"))
    specs = param_specs(O, A, L, D, N)
    p = draw_params(specs, int(g["weights_key"]), orc.schedule_tables(N))
    return g, specs, p, orc.schedule_tables(N)


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_score_forward_all_branches(golden_dir, tag):
    g, _, p, _ = _load(golden_dir, tag)
    B = int(g["B"])
    for nm, tt in (("cont", T(g["t_cont"])), ("disc", torch.full((B,), 3.0)), ("t0", torch.zeros(B)), ("t1", torch.ones(B))):
        out = orc.score_forward(p, T(g["z"]), tt, T(g["obs"]))
        np.testing.assert_allclose(out.numpy(), g[f"score_{nm}"], rtol=1e-5, atol=1e-7)
    out = orc.score_forward(p, T(g["z"]), T(g["t_cont"]), T(g["obs"]), NoiseStream(int(g["score_train_key"])), True)
    np.testing.assert_allclose(out.numpy(), g["score_train"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_belief_update_and_trajectory(golden_dir, tag):
    g, _, p, tab = _load(golden_dir, tag)
    out = orc.update_belief_via_diffusion(p, tab, T(g["obs"]), NoiseStream(int(g["belief_key"])), True)
    np.testing.assert_allclose(out["latent"].numpy(), g["belief_latent"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["reconstruction_error"].numpy(), g["belief_rec"], rtol=1e-4)
    np.testing.assert_allclose(out["latent_std"].numpy(), g["belief_std"], rtol=1e-4, atol=1e-6)
    z, traj = orc.generate_latent_trajectory(p, tab, T(g["obs"]), NoiseStream(int(g["belief_key"])), True, keep=True)
    np.testing.assert_allclose(np.stack([x.numpy() for x in traj]), g["belief_traj"], rtol=1e-4, atol=1e-6)
    out = orc.update_belief_via_diffusion(p, tab, T(g["obs"]), NoiseStream(int(g["belief_eval_key"])), False)
    np.testing.assert_allclose(out["latent"].numpy(), g["belief_eval_latent"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_small_nets(golden_dir, tag):
    g, _, p, _ = _load(golden_dir, tag)
    B, A = int(g["B"]), int(g["A"])
    z = T(g["z"])
    a, lp, ent, _, _ = orc.policy_forward(p, z, T(NoiseStream(int(g["policy_key"])).randn(B, A)))
    np.testing.assert_allclose(a.numpy(), g["policy_action"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp.numpy(), g["policy_logp"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.numpy(), g["policy_entropy"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(orc.value_forward(p, z, torch.full((B,), 2.0)).numpy(), g["value_t2"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(orc.predict_next_latent(p, z, T(g["act"]))[0].numpy(), g["dyn_next_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(orc.reward_predict(p, z).numpy(), g["reward_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(orc.decode_observation(p, z).numpy(), g["decoder_eval"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_elbo_loss_grads_and_time_importance(golden_dir, tag):
    g, _, p, _ = _load(golden_dir, tag)
    names = orc.group_names(p, "score")
    tiw = None
    for call in (1, 2):
        for n in p:
            p[n].requires_grad_(orc.is_param(n))
        loss, info, t, per = orc.diffusion_elbo(p, T(g["obs"]), T(g["rew"]), T(g["z"]), NoiseStream(int(g[f"elbo{call}_key"])), tiw, True)
        np.testing.assert_allclose(loss.item(), g[f"elbo{call}_loss"], rtol=1e-5)
        for kk in ("reconstruction_loss", "kl_loss", "score_matching_loss", "reward_loss", "grad_penalty", "mean_time"):
            np.testing.assert_allclose(float(info[kk].detach()), g[f"elbo{call}_{kk}"], rtol=1e-4, atol=1e-6)
        grads = dict(zip(names, torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)))
        gn = np.sqrt(sum(float((v ** 2).sum()) for v in grads.values() if v is not None))
        np.testing.assert_allclose(gn, g[f"elbo{call}_gradnorm"], rtol=1e-4)
        for k in g.files:
            if k.startswith(f"elbo{call}_grad/"):
                ref = g[k]
                np.testing.assert_allclose(grads[k.split("/", 1)[1]].numpy(), ref, rtol=2e-3, atol=1e-6 * max(1.0, np.abs(ref).max()))
        tiw = orc.update_time_importance(torch.ones(100) if tiw is None else tiw, t, per)
        np.testing.assert_allclose(tiw.numpy(), g[f"elbo{call}_tiw"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_efe_value_and_policy_grads(golden_dir, tag):
    g, _, p, _ = _load(golden_dir, tag)
    for n in p:
        p[n].requires_grad_(orc.is_param(n))
    efe, info, rm = orc.expected_free_energy(p, T(g["z"]), NoiseStream(int(g["efe_key"])), 0.0, int(g["H"]), 2, 2, True)
    np.testing.assert_allclose(efe.detach().numpy(), g["efe"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rm, g["efe_running_mean"], rtol=1e-5)
    np.testing.assert_allclose(float(info["epistemic/mi_estimate"]), g["efe_mi"], rtol=1e-4, atol=1e-6)
    names = orc.group_names(p, "policy")
    grads = dict(zip(names, torch.autograd.grad(efe.mean(), [p[n] for n in names])))
    for k in g.files:
        if k.startswith("efe_grad/"):
            ref = g[k]
            np.testing.assert_allclose(grads[k.split("/", 1)[1]].numpy(), ref, rtol=2e-3, atol=1e-6 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_lambda_returns(golden_dir, tag):
    g, _, _, _ = _load(golden_dir, tag)
    np.testing.assert_allclose(orc.lambda_returns(g["rew"], g["lam_next_values"], g["lam_dones"]), g["lam_returns"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.lambda_returns(g["lam64_r"], g["lam64_v"], g["lam64_d"]), g["lam64_returns"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag", ["hc", "hop"])
def test_full_train_steps(golden_dir, tag):
    """Two complete reference train_steps (K=10, S=10): every parameter's checksum + spot tensors."""
    g, specs, p, _ = _load(golden_dir, tag)
    O, A, B, N, H = (int(g[k]) for k in ("O", "A", "B", "N", "H"))
    bs = NoiseStream(int(g["store_key"])); cap = int(g["store_cap"])
    store = {"observations": bs.randn(cap, O), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32),
             "rewards": bs.randn(cap), "next_observations": bs.randn(cap, O), "dones": bs.rand(cap) < 0.1}
    st = orc.OracleAgentState(p, N)
    for step in (1, 2):
        ns = NoiseStream(int(g[f"ts{step}_key"]))
        m = orc.train_step(st, orc.replay_sample(store, cap, B, ns), ns, horizon=H)
        for k in g.files:
            if k.startswith(f"ts{step}_metric/") and k.split("/", 1)[1] in m:
                np.testing.assert_allclose(m[k.split("/", 1)[1]], g[k], rtol=2e-3, atol=1e-5, err_msg=k)
            if k.startswith(f"ts{step}_param/"):
                np.testing.assert_allclose(st.p[k.split("/", 1)[1]].detach().numpy(), g[k], rtol=1e-4, atol=2e-7, err_msg=k)
        i_rm = [n for n, _ in specs].index("epistemic_estimator.running_mean")  # buffer kept on the state
        np.testing.assert_allclose(st.running_mean, g[f"ts{step}_checksum"][i_rm], rtol=1e-4)
        st.p["epistemic_estimator.running_mean"] = torch.tensor(g[f"ts{step}_checksum"][i_rm])
        cs = np.array([float(st.p[n].double().sum()) for n, _ in specs])
        bad = np.abs(cs - g[f"ts{step}_checksum"]) > 2e-6 * g[f"ts{step}_abssum"] + 1e-6
        assert not bad.any(), [specs[i][0] for i in np.nonzero(bad)[0]]


def test_misc_goldens(golden_dir):
    m = np.load(os.path.join(golden_dir, "misc.npz"))
    keys = list(m["sched_keys"])
    for N in (10, 25, 50, 100):
        for kind in ("cosine", "linear"):
            tab = orc.schedule_tables(N, kind)
            np.testing.assert_array_equal(torch.stack([tab[k] for k in keys]).numpy(), m[f"sched_{kind}_{N}"])
    rms = orc.RunningMeanStd()
    for i in range(4):
        rms.update(NoiseStream(90 + i).randn(16) * (i + 1) + i)
        np.testing.assert_allclose([rms.mean, rms.var, rms.count], m["rms_seq"][i], rtol=1e-12)
    mu, ob, sc = m["fp_mu"], m["fp_obs"], m["fp_score"]
    np.testing.assert_allclose(-(mu - ob) / 0.01 ** 2 - mu + sc, m["fp_grad"], rtol=1e-12)
    np.testing.assert_allclose(orc.gaussian_entropy_diag(m["fp_entropy_var"]), m["fp_entropy"], rtol=1e-12)


def test_pixel_infonce_golden(golden_dir):
    """compute_representation_loss (pixel_agent.py:413-437): loss and d(loss)/d(next features) as the reference produced them."""
    from oracle.weights import draw_params, param_specs
    g = np.load(os.path.join(golden_dir, "pixel.npz"))
    p = draw_params(param_specs(17, 6, 17, 64, 4), 1234 + 17, orc.schedule_tables(4))
    tn = torch.from_numpy(g["nce_next"]).requires_grad_(True)
    loss = orc.representation_loss(p, tn, torch.from_numpy(g["nce_act"]), torch.from_numpy(g["nce_lat"]))
    np.testing.assert_allclose(loss.item(), float(g["nce_loss"]), rtol=1e-6)
    np.testing.assert_allclose(torch.autograd.grad(loss, tn)[0].numpy(), g["nce_grad"], rtol=1e-4, atol=1e-8)


def test_pixel_train_step_golden(golden_dir):
    """The oracle's DiffusionPixelAgent.train_step restatement (pixel estimator evaluated inside the EFE) against the reference's own
    post-step state and metrics (pixel_step.npz)."""
    from oracle.encoder_weights import (conv_decoder_specs, draw_conv_decoder_params, draw_encoder_params, draw_pixel_estimator_params, encoder_specs,
                                        pixel_estimator_specs)
    from oracle.weights import draw_feature_decoder, draw_params, param_specs
    g = np.load(os.path.join(golden_dir, "pixel_step.npz"))
    L, A, D, N, H, B, K, S, CAP = (int(v) for v in g["dims"])
    kw, ke, kf, ks, kstep, kdec, kest = (int(v) for v in g["keys"])
    p = draw_params(param_specs(L, A, L, D, N), kw, orc.schedule_tables(N))
    p = {k: v for k, v in p.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}
    p.update(draw_feature_decoder(L, D, kf))
    p.update({"epistemic_estimator.decoder." + k: v for k, v in draw_conv_decoder_params(conv_decoder_specs(L, hidden=D), kdec).items()})
    p.update({"epistemic_estimator." + k: v for k, v in draw_pixel_estimator_params(pixel_estimator_specs(L), kest).items()})
    p.update({"encoder." + k: v for k, v in draw_encoder_params(encoder_specs(L), ke).items()})
    bs = NoiseStream(ks)
    store = {"observations": bs.rand(CAP, 3, 3, 84, 84), "actions": np.tanh(bs.randn(CAP, A)).astype(np.float32), "rewards": bs.randn(CAP),
             "next_observations": bs.rand(CAP, 3, 3, 84, 84), "dones": np.zeros(CAP, bool)}
    st = orc.OraclePixelState(p, N); ns = NoiseStream(kstep)
    batch = orc.replay_sample(store, CAP, B, ns)
    batch["observations"] = batch["observations"].reshape(B, 9, 84, 84); batch["next_observations"] = batch["next_observations"].reshape(B, 9, 84, 84)
    torch.set_num_threads(8)
    m = orc.pixel_train_step(st, batch, ns, horizon=H, num_trajectories=K, pixel_epistemic_samples=S)
    for k in g.files:
        if k.startswith("metric/"):
            np.testing.assert_allclose(m[k.split("/", 1)[1]], float(g[k]), rtol=2e-4, atol=2e-5, err_msg=k)
        if k.startswith("param/"):
            np.testing.assert_allclose(st.p[k.split("/", 1)[1]].detach().numpy(), g[k], rtol=1e-4, atol=2e-7, err_msg=k)
    np.testing.assert_allclose(st.running_mean, float(g["running_mean"]), rtol=1e-4)
    names = [str(n) for n in g["names"]]
    cs = np.array([float(st.p[n].detach().double().sum()) for n in names])
    assert np.all(np.abs(cs - g["checksum"]) <= 2e-6 * g["abssum"] + 2e-6)


def test_train_epistemic_estimator_golden(golden_dir):
    """The oracle's train_epistemic_estimator (clamped MINE bound, EMALoss gradient, Adam with L2) against the reference's own run."""
    from oracle.weights import draw_params, param_specs
    g = np.load(os.path.join(golden_dir, "mine_train.npz"))
    O, A, D, N, B = (int(v) for v in g["dims"])
    kw, kd, k1, k2, k3 = (int(v) for v in g["keys"])
    st = orc.OracleAgentState(draw_params(param_specs(O, A, O, D, N), kw, orc.schedule_tables(N)), N)
    ds = NoiseStream(kd)
    lat, act = torch.from_numpy(ds.randn(B, O)), torch.from_numpy(np.tanh(ds.randn(B, A)).astype(np.float32))
    names = [str(n) for n in g["names"]]
    for step, key in ((1, k1), (2, k2), (3, k3)):
        mi, met, grads = orc.train_epistemic_estimator(st, lat, act, NoiseStream(key))
        np.testing.assert_allclose(mi, float(g[f"s{step}_mi"]), rtol=1e-5, atol=1e-7)
        for k, v in met.items():
            np.testing.assert_allclose(v, float(g[f"s{step}_metric/{k}"]), rtol=1e-5, atol=1e-7, err_msg=k)
        cs = np.array([float(st.p[n].detach().double().sum()) for n in names])
        assert np.all(np.abs(cs - g[f"s{step}_checksum"]) <= 2e-6 * g[f"s{step}_abssum"] + 2e-6)
    assert float(g["s1_mi"]) > 0 and float(g["s3_mi"]) == 0.0     # both branches of the clamp are covered


def test_fokker_planck_full_covariance_golden(golden_dir):
    """Full-covariance belief step: the oracle's restatement against the reference's own sub-functions chained by the harness of
    oracle/gen_golden_fp_full.py (three configurations: a well-conditioned one, one whose eigenvalues hit the clamp and the condition-number
    regularisation, and the defaults with their exp(400) growth per step)."""
    g = np.load(os.path.join(golden_dir, "fp_full.npz"))
    for name in ("sane", "illcond", "default"):
        L, steps = (int(x) for x in g[f"{name}_dims"])
        dt, D, lr, ns, mv = (float(x) for x in g[f"{name}_cfg"])
        cov, _ = orc.fp_ensure_numerical_stability(g[f"{name}_cov_init"], max(mv, 1e-8))
        np.testing.assert_allclose(cov, g[f"{name}_cov_reset"], rtol=1e-9, atol=1e-12 * np.abs(cov).max())
        mean = g[f"{name}_mean_init"]
        for t in range(steps):
            mean, cov, prec, cond = orc.fokker_planck_full_step(mean, cov, g[f"{name}_obs{t}"], g[f"{name}_score{t}"], g[f"{name}_xi{t}"],
                                                               dt=dt, D=D, lr=lr, noise_scale=ns, min_var=mv)
            np.testing.assert_allclose(mean, g[f"{name}_mean{t}"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(cov, g[f"{name}_cov{t}"], rtol=1e-8, atol=1e-10 * np.abs(cov).max())
            np.testing.assert_allclose(prec, g[f"{name}_prec{t}"], rtol=1e-6, atol=1e-9 * np.abs(prec).max())
            np.testing.assert_allclose(orc.gaussian_entropy_full(cov), float(g[f"{name}_entropy{t}"]), rtol=1e-9)
            np.testing.assert_allclose(cond, float(g[f"{name}_cond{t}"]), rtol=1e-6)


def test_round3_surface_goldens(golden_dir):
    """The functions pinned in round 3 (oracle/gen_golden_r3.py ran the reference): lambda-returns without the immediate reward (bit-exact),
    the stand-alone gradient penalty, the deterministic reverse trajectory, pixel-mode update_belief_via_diffusion."""
    from oracle.weights import draw_feature_decoder
    g = np.load(os.path.join(golden_dir, "r3_surface.npz"))
    O, A, D, N, B = (int(x) for x in g["dims"])
    p = draw_params(param_specs(O, A, O, D, N), int(g["weights_key"]), orc.schedule_tables(N))
    for tag in "abc":
        r, nv, d, ns_ = g[f"lam_{tag}_r"], g[f"lam_{tag}_nv"], g[f"lam_{tag}_d"], int(g[f"lam_{tag}_n_steps"])
        np.testing.assert_array_equal(orc.lambda_returns(r, nv, d, float(g["gamma"]), 0.95, ns_, exclude_immediate_rewards=True), g[f"lam_{tag}_excl"])
        np.testing.assert_array_equal(orc.lambda_returns(r, nv, d, float(g["gamma"]), 0.95, ns_), g[f"lam_{tag}_incl"])
        assert not np.array_equal(g[f"lam_{tag}_excl"], g[f"lam_{tag}_incl"])
    for tag, train, tt in (("train_cont", True, T(g["t_cont"])), ("eval_cont", False, T(g["t_cont"])), ("eval_disc", False, torch.full((B,), 3.0))):
        v = orc.gradient_penalty(p, T(g["z"]), tt, T(g["obs"]), NoiseStream(int(g["gp_key"])), train)
        np.testing.assert_allclose(float(v), float(g[f"gp_{tag}"]), rtol=1e-5)
    _, tr = orc.generate_latent_trajectory(p, orc.schedule_tables(N), T(g["obs"]), NoiseStream(int(g["det_key"])), False, keep=True, deterministic=True)
    np.testing.assert_allclose(np.stack([x.numpy() for x in tr]), g["det_traj"], rtol=1e-5, atol=1e-6)
    L, A2, Dp, Np, Bp = (int(x) for x in g["pix_dims"])
    pp = draw_params(param_specs(L, A2, L, Dp, Np), int(g["pix_weights_key"]), orc.schedule_tables(Np))
    pp.update(draw_feature_decoder(L, Dp, int(g["pix_fdec_key"])))
    for tag, train in (("train", True), ("eval", False)):
        out = orc.update_belief_via_diffusion(pp, orc.schedule_tables(Np), T(g["pix_feats"]), NoiseStream(int(g[f"pix_{tag}_key"])), train, pixel=True, keep=True)
        np.testing.assert_allclose(out["latent"].numpy(), g[f"pix_{tag}_latent"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(float(out["reconstruction_error"]), float(g[f"pix_{tag}_rec"]), rtol=1e-5)
        np.testing.assert_allclose(np.stack([x.numpy() for x in out["trajectory"]]), g[f"pix_{tag}_traj"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("L", [17, 128])
def test_a5_forward_process_oracle_vs_reference_golden(golden_dir, L):
    """SURVEY row A5 (core/diffusion.py:56-104, 146-174, 239-255): the oracle's restatement against the reference's own outputs
    (tests/golden/a5_surface.npz, oracle/gen_golden_a5.py; bit-identical when written - 1e-6 here because the host's vector exp / cos may differ by an ulp
    between CPUs)."""
    import numpy as np
    import torch
    from oracle import aid_oracle as orc
    from oracle.gen_golden_a5 import B, N, inputs
    from oracle.noise import NoiseStream
    g = np.load(os.path.join(golden_dir, "a5_surface.npz"))
    key = int(g[f"L{L}/key"])
    d = inputs(L, key)
    T = torch.from_numpy
    p = {"latent_diffusion.latent_prior_mean": T(d["prior_mean"]), "latent_diffusion.latent_prior_log_std": T(d["prior_log_std"]),
         "latent_diffusion.log_snr_min": torch.tensor(float(d["log_snr_min"])), "latent_diffusion.log_snr_max": torch.tensor(float(d["log_snr_max"]))}
    tab = orc.schedule_tables(N)
    z, z_t, eps, t, steps = T(d["z"]), T(d["z_t"]), T(d["eps"]), T(d["t"]), T(d["steps"])
    close = lambda a, k: np.testing.assert_allclose(a.numpy(), g[f"L{L}/{k}"], rtol=2e-6, atol=5e-7, err_msg=k)
    close(orc.log_snr(p, t), "log_snr")
    zn, _, info = orc.continuous_q_sample(p, z, t, eps)
    close(zn, "z_noisy"); close(info["alpha"], "alpha"); close(info["sigma"], "sigma")
    close(orc.loss_weight(p, t), "loss_weight")
    close(orc.sample_latent_prior(p, T(NoiseStream(key + 50).randn(B, L))), "prior")
    close(orc.q_sample(tab, z, steps, eps), "q_sample")
    close(orc.posterior_mean(tab, z, z_t, steps), "posterior_mean")
    assert "out_features" in str(g["uncond_reference_raises"])       # the reference's observation=None branch raises (recorded fact)
    assert orc.score_forward  # the oracle's zero-embedding branch is exercised on the GPU box (tests/test_hip_a5.py)


def test_jacobian_features_oracle_vs_reference_golden(golden_dir):
    """core/active_inference.py:940-993 (state branch): the oracle's `jacobian_features` against the reference's own compute_jacobian_features
    (tests/golden/jacobian_features.npz, oracle/gen_golden_jac.py; bit-identical when written)."""
    import os
    import numpy as np
    import torch
    from oracle import aid_oracle as orc
    from oracle.noise import NoiseStream
    from oracle.weights import draw_params, param_specs
    g = np.load(os.path.join(golden_dir, "jacobian_features.npz"))
    O, A, D, N, R = (int(g[k]) for k in ("O", "A", "D", "N", "R"))
    p = draw_params(param_specs(O, A, O, D, N), int(g["weights_key"]), orc.schedule_tables(N))
    z = torch.from_numpy(NoiseStream(int(g["z_key"])).randn(R, O))
    for mode in ("eval", "train"):
        out = orc.jacobian_features(p, z, NoiseStream(int(g[f"noise_key_{mode}"])), mode == "train")
        np.testing.assert_allclose(out.numpy(), g[f"jac_{mode}"], rtol=1e-5, atol=1e-6)
