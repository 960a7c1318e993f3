"""GPU parity of the training-side entry points (ELBO fwd+bwd incl. gradient penalty, optimiser)
against the oracle's autograd on the same seeded inputs and noise."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import aid_oracle as orc
from oracle.noise import NoiseStream
from oracle.weights import draw_params, param_specs

T = torch.from_numpy


@pytest.fixture(scope="module")
def nat():
    from aid_amd import _native
    _native.lib()
    return _native


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda().contiguous()


def _elbo_noise(key, B, L, D, first, tiw):
    """Materialise the ELBO's draws in the reference order (oracle.diffusion_elbo docstring)."""
    ns = NoiseStream(key)
    n = {"dec_mask0": ns.dropout_mask((B, 2 * D), 0.2), "dec_mask1": ns.dropout_mask((B, 2 * D), 0.2), "dec_mask2": ns.dropout_mask((B, D), 0.2)}
    n["t"] = ns.rand(B) if first else orc.importance_sample_time(tiw, B, ns).numpy()
    n["eps"] = ns.randn(B, L); n["score_mask"] = ns.dropout_mask((B, D), 0.1); n["gp_mask"] = ns.dropout_mask((B, D), 0.1)
    n["prior_eps"] = ns.randn(B, L)
    return n


def _run_elbo_case(nat, O, A, D, N, B, wkey, key, rtol_grad):
    L = O
    specs = param_specs(O, A, L, D, N)
    p = draw_params(specs, wkey, orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    net = nat.score_net_table(pd, "latent_score_network.", L, L, D)
    gnet = nat.score_net_table(gd, "latent_score_network.", L, L, D)
    dp, dg = nat.diffusion_table(pd), nat.diffusion_table(gd)
    dec, rew = nat.decoder_table(pd), nat.reward_table(pd)
    ins = NoiseStream(key + 1000)
    obs, rew_in, z = ins.randn(B, O), ins.randn(B), ins.randn(B, L)
    tiw_dev = torch.zeros(100, device="cuda")
    tiw = None
    names = orc.group_names(p, "score")
    for call in (1, 2):
        for k in gd.values():
            k.zero_()
        nz = _elbo_noise(key + call, B, L, D, call == 1, tiw)
        metrics, t_out, per = nat.elbo_step(net, gnet, dp, dg, dec, rew, dev(obs), dev(rew_in), dev(z), tiw_dev, call == 1,
                                            noise={k: dev(v) for k, v in nz.items()}, train=True)
        for n_ in p:
            p[n_].requires_grad_(orc.is_param(n_))
        loss, info, t_ref, per_ref = orc.diffusion_elbo(p, T(obs), T(rew_in), T(z), NoiseStream(key + call), tiw, True)
        gref = dict(zip(names, torch.autograd.grad(loss, [p[n_] for n_ in names], allow_unused=True)))
        m = metrics.cpu().numpy()
        got = dict(zip(nat.ELBO_METRICS, m))
        for kk in ("reconstruction_loss", "kl_loss", "score_matching_loss", "reward_loss", "grad_penalty", "mean_time", "loss_weight_mean", "elbo"):
            np.testing.assert_allclose(got[kk], float(info[kk].detach()), rtol=1e-4, atol=1e-6, err_msg=kk)
        np.testing.assert_allclose(got["loss"], float(loss.detach()), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(per.cpu().numpy(), per_ref.numpy(), rtol=1e-4, atol=1e-7)
        worst = ("", 0.0)
        for n_ in names:
            g_ref = gref[n_]
            g_hip = gd[n_].cpu()
            if g_ref is None:
                assert float(g_hip.abs().max()) == 0.0, n_
                continue
            scale = float(g_ref.abs().max()) + 1e-12
            err = float((g_hip - g_ref).abs().max()) / scale
            if err > worst[1]:
                worst = (n_, err)
            assert err < rtol_grad, (n_, err, scale)
        tiw = orc.update_time_importance(torch.ones(100) if tiw is None else tiw, t_ref, per_ref)
        np.testing.assert_allclose(tiw_dev.cpu().numpy(), tiw.numpy(), rtol=1e-5, atol=1e-7)
        for n_ in p:
            p[n_].requires_grad_(False)
    return worst


@pytest.mark.parametrize("O,A", [(17, 6), (11, 3)])
def test_elbo_step_small(nat, O, A):
    """Golden-fixture dimensions (D=64, B=8): every score/diffusion gradient incl. the GP double backward."""
    _run_elbo_case(nat, O, A, 64, 4, 8, 1234 + O, 20, 1e-3)


def test_elbo_step_full_size(nat):
    """BASELINE config 2 shape (D=512, B=256)."""
    _run_elbo_case(nat, 17, 6, 512, 10, 256, 77, 500, 1e-3)


def test_elbo_golden_metrics(nat, golden_dir):
    """Against the reference's own numbers (tests/golden): loss terms + a spot set of gradients."""
    g = np.load(os.path.join(golden_dir, "state_hc.npz"))
    O, A, L, D, N, B = (int(g[k]) for k in ("O", "A", "L", "D", "N", "B"))
    p = draw_params(param_specs(O, A, L, D, N), int(g["weights_key"]), orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    net, gnet = nat.score_net_table(pd, "latent_score_network.", L, L, D), nat.score_net_table(gd, "latent_score_network.", L, L, D)
    tiw_dev = torch.zeros(100, device="cuda")
    nz = _elbo_noise(int(g["elbo1_key"]), B, L, D, True, None)
    metrics, _, _ = nat.elbo_step(net, gnet, nat.diffusion_table(pd), nat.diffusion_table(gd), nat.decoder_table(pd), nat.reward_table(pd),
                                  dev(g["obs"]), dev(g["rew"]), dev(g["z"]), tiw_dev, True, noise={k: dev(v) for k, v in nz.items()})
    got = dict(zip(nat.ELBO_METRICS, metrics.cpu().numpy()))
    for kk in ("reconstruction_loss", "kl_loss", "score_matching_loss", "reward_loss", "grad_penalty", "mean_time", "loss_weight_mean"):
        np.testing.assert_allclose(got[kk], g[f"elbo1_{kk}"], rtol=1e-4, atol=1e-6, err_msg=kk)
    np.testing.assert_allclose(got["loss"], g["elbo1_loss"], rtol=1e-4)
    np.testing.assert_allclose(tiw_dev.cpu().numpy(), g["elbo1_tiw"], rtol=1e-5, atol=1e-7)
    for k in g.files:
        if k.startswith("elbo1_grad/"):
            ref = g[k]; got_g = gd[k.split("/", 1)[1]].cpu().numpy()
            np.testing.assert_allclose(got_g, ref, rtol=2e-3, atol=2e-6 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_adamw_clip_ema(nat):
    n = 100003
    rng = np.random.default_rng(0)
    p0, g0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32) * 0.01
    p, g = dev(p0), dev(g0)
    m, v, ema = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), p.clone()
    scratch = torch.zeros(8192, device="cuda")
    pr, mr, vr, er = T(p0.copy()), torch.zeros(n), torch.zeros(n), T(p0.copy())
    for step in (1, 2, 3):
        nat.adamw_step(p, g, m, v, ema, 5e-5, 1e-5, step, 0.5, scratch)
        gg = T(g0.copy())
        total, coef = orc.clip_grad_norm([gg], 0.5)
        orc.adamw_step(pr, gg, mr, vr, step, 5e-5, 1e-5)
        er = 0.9999 * er + (1 - 0.9999) * pr
        np.testing.assert_allclose(float(scratch[0]), float(total), rtol=1e-5)
        np.testing.assert_allclose(p.cpu().numpy(), pr.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ema.cpu().numpy(), er.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(v.cpu().numpy(), vr.numpy(), rtol=1e-5, atol=1e-12)


# ------------------------------------------------------------------------------------------------- EFE / MINE
def _efe_noise(key, B, L, A, H, K, S):
    """Reference draw order (k outer, t inner) -> the batched [t][k] layouts of AidEfeNoise."""
    ns = NoiseStream(key)
    M, R = K * B, K * S * B
    n = {"policy_eps": np.zeros((H, M, A), np.float32), "reparam_eps": np.zeros((H, M, L), np.float32),
         "mine_z_eps": np.zeros((H, R, L), np.float32), "mine_delta": np.zeros((4, H, R, L), np.float32),
         "proj_mask": np.zeros((H, R, 512), np.float32), "mine_mask": np.zeros((4, H, R, 512), np.float32),
         "perms": np.zeros((H, K, S, B), np.int32)}
    for k in range(K):
        rs = slice(k * S * B, (k + 1) * S * B)
        for t in range(H):
            n["policy_eps"][t, k * B:(k + 1) * B] = ns.randn(B, A)
            n["reparam_eps"][t, k * B:(k + 1) * B] = ns.randn(B, L)
            for s_ in range(S):
                n["mine_z_eps"][t, k * S * B + s_ * B:k * S * B + (s_ + 1) * B] = ns.randn(B, L)
            for j in range(4):
                n["mine_delta"][j, t, rs] = ns.randn(S * B, L)
            n["proj_mask"][t, rs] = ns.dropout_mask((S * B, 512), 0.1)
            n["mine_mask"][0, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
            n["mine_mask"][1, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
            for s_ in range(S):
                n["perms"][t, k, s_] = ns.randperm(B)
            n["mine_mask"][2, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
            n["mine_mask"][3, t, rs] = ns.dropout_mask((S * B, 512), 0.1)
    return n


def _efe_case(nat, O, A, D, B, H, K, S, wkey, key, check_grads=True):
    L, N = O, 4
    p = draw_params(param_specs(O, A, L, D, N), wkey, orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    z = NoiseStream(key + 7).randn(B, L)
    nz = _efe_noise(key, B, L, A, H, K, S)
    nzd = {k: (torch.from_numpy(v).cuda().contiguous() if k == "perms" else dev(v)) for k, v in nz.items()}
    rm = torch.zeros(1, device="cuda")
    efe, info = nat.efe_step(nat.policy_table(pd), nat.policy_table(gd), nat.dynamics_table(pd, A), nat.reward_table(pd), nat.value_table(pd),
                             nat.decoder_table(pd), nat.epistemic_table(pd), dev(z), rm, H, K, S, noise=nzd, train=True)
    for n_ in p:
        p[n_].requires_grad_(orc.is_param(n_))
    efe_ref, info_ref, rm_ref = orc.expected_free_energy(p, T(z), NoiseStream(key), 0.0, H, K, S, True)
    np.testing.assert_allclose(efe.cpu().numpy(), efe_ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    got = dict(zip(nat.EFE_INFO, info.cpu().numpy()))
    np.testing.assert_allclose(got["policy_loss"], float(efe_ref.mean().detach()), rtol=2e-4, atol=1e-5)
    for kk in ("epistemic_mean", "pragmatic_mean", "consistency_mean", "epistemic/mi_estimate", "epistemic/joint_term", "epistemic/marginal_term"):
        np.testing.assert_allclose(got[kk], float(torch.as_tensor(info_ref[kk]).detach()), rtol=5e-4, atol=2e-5, err_msg=kk)
    np.testing.assert_allclose(float(rm), rm_ref, rtol=1e-4)
    if check_grads:
        names = orc.group_names(p, "policy")
        gref = torch.autograd.grad(efe_ref.mean(), [p[n_] for n_ in names])
        for n_, gr in zip(names, gref):
            scale = float(gr.abs().max()) + 1e-12
            err = float((gd[n_].cpu() - gr).abs().max()) / scale
            assert err < 2e-3, (n_, err, scale)
        for n_ in p:   # nothing but the policy receives gradients
            if not n_.startswith("policy_network."):
                assert float(gd[n_].abs().max()) == 0.0, n_


@pytest.mark.parametrize("O,A", [(17, 6), (11, 3)])
def test_efe_step_small(nat, O, A):
    _efe_case(nat, O, A, 64, 8, 2, 2, 2, 1234 + O, 30)


def test_efe_step_reference_kh(nat):
    """K=10, S=10, H=5 as hard-coded in train_step, small width."""
    _efe_case(nat, 17, 6, 64, 8, 5, 10, 10, 91, 300)


def test_efe_step_full_size(nat):
    """D=512, B=256, K=2, S=2, H=2: full-width kernels against the oracle."""
    _efe_case(nat, 17, 6, 512, 256, 2, 2, 2, 17, 700)


def test_efe_golden(nat, golden_dir):
    g = np.load(os.path.join(golden_dir, "state_hc.npz"))
    O, A, L, D, N, B, H = (int(g[k]) for k in ("O", "A", "L", "D", "N", "B", "H"))
    p = draw_params(param_specs(O, A, L, D, N), int(g["weights_key"]), orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    nz = _efe_noise(int(g["efe_key"]), B, L, A, H, 2, 2)
    nzd = {k: (torch.from_numpy(v).cuda().contiguous() if k == "perms" else dev(v)) for k, v in nz.items()}
    rm = torch.zeros(1, device="cuda")
    efe, info = nat.efe_step(nat.policy_table(pd), nat.policy_table(gd), nat.dynamics_table(pd, A), nat.reward_table(pd), nat.value_table(pd),
                             nat.decoder_table(pd), nat.epistemic_table(pd), dev(g["z"]), rm, H, 2, 2, noise=nzd, train=True)
    np.testing.assert_allclose(efe.cpu().numpy(), g["efe"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(rm), g["efe_running_mean"], rtol=1e-4)
    got = dict(zip(nat.EFE_INFO, info.cpu().numpy()))
    np.testing.assert_allclose(got["epistemic/mi_estimate"], g["efe_mi"], rtol=5e-4, atol=2e-5)
    np.testing.assert_allclose(got["pragmatic_mean"], g["efe_pragmatic_mean"], rtol=5e-4, atol=2e-5)
    for k in g.files:
        if k.startswith("efe_grad/"):
            ref = g[k]
            np.testing.assert_allclose(gd[k.split("/", 1)[1]].cpu().numpy(), ref, rtol=2e-3, atol=2e-6 * max(1.0, np.abs(ref).max()), err_msg=k)


# ------------------------------------------------------------------------------------------------- value / dynamics / lambda
def test_lambda_returns_golden(nat, golden_dir):
    for tag in ("hc", "hop"):
        g = np.load(os.path.join(golden_dir, f"state_{tag}.npz"))
        out = nat.lambda_returns(dev(g["rew"]), dev(g["lam_next_values"]), dev(g["lam_dones"].astype(np.float32)))
        np.testing.assert_allclose(out.cpu().numpy(), g["lam_returns"], rtol=1e-6, atol=1e-7)
        out = nat.lambda_returns(dev(g["lam64_r"]), dev(g["lam64_v"]), dev(g["lam64_d"].astype(np.float32)))
        np.testing.assert_allclose(out.cpu().numpy(), g["lam64_returns"], rtol=1e-6, atol=1e-7)
    # edge cases: B=1, B=2, all done, 1024 rows vs the oracle
    for B, pd_ in ((1, 0.0), (2, 0.5), (7, 1.0), (1024, 0.01)):
        ns = NoiseStream(B)
        r, v, d = ns.randn(B), ns.randn(B), (ns.rand(B) < pd_)
        out = nat.lambda_returns(dev(r), dev(v), dev(d.astype(np.float32)))
        np.testing.assert_allclose(out.cpu().numpy(), orc.lambda_returns(r, v, d), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("D,B", [(64, 8), (512, 256)])
def test_value_and_dynamics_steps(nat, D, B):
    O, A, L, N = 17, 6, 17, 4
    p = draw_params(param_specs(O, A, L, D, N), 5 + D, orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    ns = NoiseStream(D)
    z, nzl, act, r = ns.randn(B, L), ns.randn(B, L), np.tanh(ns.randn(B, A)).astype(np.float32), ns.randn(B)
    d = (ns.rand(B) < 0.1)
    loss, targets = nat.value_step(nat.value_table(pd), nat.value_table(gd), dev(z), dev(nzl), dev(r), dev(d.astype(np.float32)))
    for n_ in p:
        p[n_].requires_grad_(orc.is_param(n_))
    values = orc.value_forward(p, T(z), torch.zeros(B)).squeeze(-1)
    with torch.no_grad():
        nv = orc.value_forward(p, T(nzl), torch.ones(B)).squeeze(-1)
        tg = T(orc.lambda_returns(r, nv.numpy(), d))
    vloss = torch.nn.functional.huber_loss(values, tg)
    np.testing.assert_allclose(targets.cpu().numpy(), tg.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(loss), float(vloss), rtol=1e-4)
    names = orc.group_names(p, "value")
    for n_, gr in zip(names, torch.autograd.grad(vloss, [p[n_] for n_ in names])):
        scale = float(gr.abs().max()) + 1e-12
        assert float((gd[n_].cpu() - gr).abs().max()) / scale < 2e-3, n_
    dloss = nat.dynamics_step(nat.dynamics_table(pd, A), nat.dynamics_table(gd, A), dev(z), dev(act), dev(nzl))
    pred, _ = orc.predict_next_latent(p, T(z), T(act))
    ref = torch.nn.functional.mse_loss(pred, T(nzl))
    np.testing.assert_allclose(float(dloss), float(ref), rtol=1e-4)
    names = orc.group_names(p, "dynamics")
    for n_, gr in zip(names, torch.autograd.grad(ref, [p[n_] for n_ in names])):
        scale = float(gr.abs().max()) + 1e-12
        assert float((gd[n_].cpu() - gr).abs().max()) / scale < 2e-3, n_


def test_small_net_forwards_golden(nat, golden_dir):
    g = np.load(os.path.join(golden_dir, "state_hc.npz"))
    O, A, L, D, N, B = (int(g[k]) for k in ("O", "A", "L", "D", "N", "B"))
    p = draw_params(param_specs(O, A, L, D, N), int(g["weights_key"]), orc.schedule_tables(N))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    z = dev(g["z"])
    a, lp, ent = nat.policy_forward(nat.policy_table(pd), z, dev(NoiseStream(int(g["policy_key"])).randn(B, A)))
    np.testing.assert_allclose(a.cpu().numpy(), g["policy_action"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lp.cpu().numpy(), g["policy_logp"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ent.cpu().numpy(), g["policy_entropy"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nat.value_forward(nat.value_table(pd), z, dev(np.full(B, 2.0))).cpu().numpy(), g["value_t2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nat.predict_next_latent(nat.dynamics_table(pd, A), z, dev(g["act"])).cpu().numpy(), g["dyn_next_mean"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nat.reward_forward(nat.reward_table(pd), z).cpu().numpy(), g["reward_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nat.decode_observation(nat.decoder_table(pd), z).cpu().numpy(), g["decoder_eval"], rtol=1e-4, atol=1e-5)


def test_replay_gather_bit_exact(nat):
    cap, O, A, B = 4096, 17, 6, 256
    ns = NoiseStream(4)
    store = {"observations": ns.randn(cap, O), "actions": ns.randn(cap, A), "rewards": ns.randn(cap), "next_observations": ns.randn(cap, O),
             "dones": (ns.rand(cap) < 0.1)}
    b = orc.replay_sample(store, cap, B, NoiseStream(9))
    idx = torch.from_numpy(b["indices"]).cuda()
    o, n, a, r, d = nat.replay_gather(dev(store["observations"]), dev(store["next_observations"]), dev(store["actions"]), dev(store["rewards"]),
                                      dev(store["dones"].astype(np.float32)), idx)
    assert np.array_equal(o.cpu().numpy(), b["observations"]) and np.array_equal(n.cpu().numpy(), b["next_observations"])
    assert np.array_equal(a.cpu().numpy(), b["actions"]) and np.array_equal(r.cpu().numpy(), b["rewards"])
    assert np.array_equal(d.cpu().numpy() != 0, b["dones"])


def test_fokker_planck_fp64(nat, golden_dir):
    n, L = 33, 17
    rng = np.random.default_rng(3)
    mean, var = rng.standard_normal((n, L)), np.abs(rng.standard_normal((n, L))) + 0.1
    obs, sc, xi = rng.standard_normal((n, L)), rng.standard_normal((n, L)), rng.standard_normal((n, L))
    for kw in ({}, {"dt": 1e-6, "noise_scale": 0.5, "max_var": 100.0}):
        m_ref, v_ref = orc.fokker_planck_step(mean, var, obs, sc, xi, **kw)
        md, vd = torch.from_numpy(mean.copy()).cuda(), torch.from_numpy(var.copy()).cuda()
        ent = nat.fokker_planck_step(md, vd, torch.from_numpy(obs).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(xi).cuda(), **kw)
        np.testing.assert_allclose(md.cpu().numpy(), m_ref, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(vd.cpu().numpy(), v_ref, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ent.cpu().numpy(), orc.gaussian_entropy_diag(v_ref), rtol=1e-12)
    # pinned sub-function: the gradient the reference's autodiff returns (tests/golden/misc.npz)
    m = np.load(os.path.join(golden_dir, "misc.npz"))
    mu, ob, s_ = m["fp_mu"][None], m["fp_obs"][None], m["fp_score"][None]
    md, vd = torch.from_numpy(mu.copy()).cuda(), torch.ones(1, 6, dtype=torch.float64, device="cuda")
    nat.fokker_planck_step(md, vd, torch.from_numpy(ob).cuda(), torch.from_numpy(s_).cuda(), torch.zeros(1, 6, dtype=torch.float64, device="cuda"))
    g = m["fp_grad"]
    expect = mu[0] + (-0.1 * g) * (0.01 / (1 + 0.1 * np.linalg.norm(g)))
    np.testing.assert_allclose(md.cpu().numpy()[0], expect, rtol=1e-12)


@pytest.mark.parametrize("n", [1, 2, 17, 33, 128])
def test_sym_eig_jacobi_fp64(nat, n):
    """On-device symmetric eigen-decomposition (parallel cyclic Jacobi, fp64) against numpy.linalg.eigh: eigenvalues to 1e-12 of the
    spectral radius, A V = V diag(w) and V^T V = I to 1e-12; a wide spectrum (1e-8 .. 1e3) so small eigenvalues are resolved relatively."""
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    wtrue = np.exp(rng.uniform(np.log(1e-8), np.log(1e3), n))
    A = (Q * wtrue) @ Q.T; A = 0.5 * (A + A.T)
    w, V = nat.sym_eig(torch.from_numpy(A).cuda())
    w, V = w.cpu().numpy(), V.cpu().numpy()
    np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(A), rtol=0, atol=1e-12 * np.abs(wtrue).max())
    assert np.max(np.abs(A @ V - V * w)) < 1e-12 * np.abs(wtrue).max()
    assert np.max(np.abs(V.T @ V - np.eye(n))) < 1e-12


def test_fokker_planck_full_covariance_reference_golden(nat, golden_dir):
    """BeliefDynamics(use_full_covariance=True) on the HIP kernels against the reference's own sub-functions (tests/golden/fp_full.npz):
    reset(initial_cov) stabilisation, then mean / covariance / precision / entropy / condition number of every step.  fp64; the tolerances
    are those the oracle holds against the same vectors (the precision of an ill-conditioned matrix is the loosest: 1e-6)."""
    import aid_amd
    g = np.load(os.path.join(golden_dir, "fp_full.npz"))
    for name in ("sane", "illcond", "default"):
        L, steps = (int(x) for x in g[f"{name}_dims"])
        dt, D, lr, ns, mv = (float(x) for x in g[f"{name}_cfg"])
        cfg = aid_amd.BeliefDynamicsConfig(use_full_covariance=True, dt=dt, diffusion_coefficient=D, learning_rate=lr, noise_scale=ns, min_variance=mv)
        bd = aid_amd.BeliefDynamics(L, cfg).cuda()
        bd.reset(torch.from_numpy(g[f"{name}_mean_init"]), torch.from_numpy(g[f"{name}_cov_init"]))
        ref = g[f"{name}_cov_reset"]
        np.testing.assert_allclose(bd.covariance.cpu().numpy(), ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
        for t in range(steps):
            m32, c32 = bd.update(torch.from_numpy(g[f"{name}_obs{t}"]), torch.from_numpy(g[f"{name}_score{t}"]), xi=torch.from_numpy(g[f"{name}_xi{t}"]))
            cov, prec = g[f"{name}_cov{t}"], g[f"{name}_prec{t}"]
            np.testing.assert_allclose(bd.mean.cpu().numpy()[0], g[f"{name}_mean{t}"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(bd.covariance.cpu().numpy(), cov, rtol=1e-8, atol=1e-10 * np.abs(cov).max())
            np.testing.assert_allclose(bd.precision.cpu().numpy(), prec, rtol=1e-6, atol=1e-9 * np.abs(prec).max())
            np.testing.assert_allclose(float(bd.entropy()), float(g[f"{name}_entropy{t}"]), rtol=1e-9)
            np.testing.assert_allclose(float(bd._last_cond[0]), float(g[f"{name}_cond{t}"]), rtol=1e-6)
            assert m32.dtype == torch.float32 and c32.shape == (L, L)
        # get_diagnostics (belief_dynamics.py:391-409) against the same quantities taken from the reference's covariance matrix
        d = bd.get_diagnostics()
        w = np.linalg.eigvalsh(cov)
        np.testing.assert_allclose([d["min_eigenvalue"], d["max_eigenvalue"], d["condition_number"]], [w.min(), w.max(), w.max() / w.min()], rtol=1e-6)
        np.testing.assert_allclose(np.log(d["determinant"]) if d["determinant"] > 0 and np.isfinite(d["determinant"]) else np.log(w).sum(), np.log(w).sum(), rtol=1e-8)
        np.testing.assert_allclose(d["mean_norm"], np.linalg.norm(g[f"{name}_mean{steps - 1}"]), rtol=1e-12)
        np.testing.assert_allclose(d["entropy"], float(g[f"{name}_entropy{steps - 1}"]), rtol=1e-9)
    dd = aid_amd.BeliefDynamics(6, aid_amd.BeliefDynamicsConfig()).cuda().get_diagnostics()            # diagonal belief after reset: unit variance
    assert dd["min_variance"] == dd["max_variance"] == dd["mean_variance"] == 1.0 and dd["mean_norm"] == 0.0


def test_fokker_planck_full_covariance_batched_vs_oracle(nat):
    """Three beliefs in one launch (L = 24), two steps, against the oracle's per-belief full-matrix restatement (scipy expm + numpy eigh)."""
    import aid_amd
    n, L = 3, 24
    rng = np.random.default_rng(11)
    cfg = aid_amd.BeliefDynamicsConfig(use_full_covariance=True, noise_scale=1.5, dt=0.02)
    bd = aid_amd.BeliefDynamics(L, cfg, n_beliefs=n).cuda()
    m0 = rng.standard_normal((n, L)); c0 = np.stack([(lambda Q: (Q * np.exp(rng.uniform(-9, 3, L))) @ Q.T)(np.linalg.qr(rng.standard_normal((L, L)))[0]) for _ in range(n)])
    bd.reset(torch.from_numpy(m0), torch.from_numpy(c0))
    mean = m0.copy(); cov = [orc.fp_ensure_numerical_stability(c0[b], 1e-6)[0] for b in range(n)]
    for t in range(2):
        obs, sc, xi = rng.standard_normal((n, L)), rng.standard_normal((n, L)), rng.standard_normal((n, L))
        bd.update(torch.from_numpy(obs), torch.from_numpy(sc), xi=torch.from_numpy(xi))
        for b in range(n):
            mean[b], cov[b], prec, cond = orc.fokker_planck_full_step(mean[b], cov[b], obs[b], sc[b], xi[b], dt=0.02, noise_scale=1.5)
            np.testing.assert_allclose(bd.mean.cpu().numpy()[b], mean[b], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(bd.covariance.cpu().numpy()[b], cov[b], rtol=1e-8, atol=1e-10 * np.abs(cov[b]).max())
            np.testing.assert_allclose(bd.precision.cpu().numpy()[b], prec, rtol=1e-6, atol=1e-9 * np.abs(prec).max())
            np.testing.assert_allclose(float(bd.entropy()[b]), orc.gaussian_entropy_full(cov[b]), rtol=1e-9)


def test_train_epistemic_estimator_reference_golden(golden_dir):
    """DiffusionActiveInference.train_epistemic_estimator x3 against a run of the REFERENCE (tests/golden/mine_train.npz; the harness
    attached the optimiser base_agent.py:134-139 defines): steps 1-2 have a positive bound (live gradient through both MINE passes, the
    projector, the 4 feature-extractor passes, the latent processor and perturbation_scale), step 3 a negative one (clamp gates the
    gradient; only Adam's L2 term moves the weights)."""
    from parity_noise import ParityNoise
    from test_hip_agent import _agent
    g = np.load(os.path.join(golden_dir, "mine_train.npz"))
    O, A, D, N, B = (int(v) for v in g["dims"])
    kw, kd, k1, k2, k3 = (int(v) for v in g["keys"])
    agent = _agent(O, A, D, N, 2, B)
    ai = agent.active_inference
    p = draw_params(param_specs(O, A, O, D, N), kw, orc.schedule_tables(N))
    sd0 = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd0 if k.startswith("epistemic_estimator.decoder.")}})
    ds = NoiseStream(kd)
    lat, act = dev(ds.randn(B, O)), dev(np.tanh(ds.randn(B, A)).astype(np.float32))
    names = [str(n) for n in g["names"]]
    pn = ParityNoise(0); ai.noise = pn
    for step, key in ((1, k1), (2, k2), (3, k3)):
        pn.rekey(key)
        mi, met = ai.train_epistemic_estimator(lat, act, None)
        np.testing.assert_allclose(mi, float(g[f"s{step}_mi"]), rtol=2e-3, atol=2e-5)
        for k in ("epistemic/mi_estimate", "epistemic/joint_term", "epistemic/marginal_term", "epistemic/running_mean"):
            np.testing.assert_allclose(met[k], float(g[f"s{step}_metric/{k}"]), rtol=2e-3, atol=2e-5, err_msg=f"step {step} {k}")
        if step == 1:   # gradients (the golden holds them after clip_grad_norm_)
            gv = ai.arenas["epistemic"].grad_views()
            total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in gv.values())))
            coef = min(1.0, agent.config.gradient_clip / (total + 1e-6))
            gmax = max(float(np.abs(g[f"s1_grad/{n}"]).max()) for n in names)
            for n in names:
                ref = g[f"s1_grad/{n}"]; got = gv[n].reshape(-1)
                got = (got if got.numel() <= 4096 else got[::97]).cpu().numpy() * coef
                # bias gradients of the MINE network are sums that cancel (mine_network.6.bias is -1 + mean(exp(t_marg)) / running_mean
                # = 0 up to rounding on the first call), and a pre-activation within rounding of 0 flips its ReLU gate for one row (one element
                # of mine_network.0.bias moves by 5e-3 of the tensor's scale while the rest agree to 1e-5): 1e-2 of the tensor's own scale
                # plus a norm-wise floor of 1e-5 of the largest gradient
                tol = 1e-2 * float(np.abs(ref).max()) + 1e-5 * gmax
                assert np.abs(got - ref).max() < tol, (n, np.abs(got - ref).max(), tol)
                np.testing.assert_allclose(float(gv[n].double().abs().sum()) * coef, g[f"s1_gradsum/{n}"][1], rtol=2e-3, atol=1e-4 * gmax)
        sd = ai.state_dict()
        cs = np.array([float(sd[n].double().sum()) for n in names])
        # + 2 * lr per accumulated step: one element whose gradient is rounding noise (mine_network.6.bias on the first call: -1 + mean(exp(t)) /
        # running_mean = 0) takes its lr-sized Adam step in the direction of the noise
        bad = np.abs(cs - g[f"s{step}_checksum"]) > 2e-6 * g[f"s{step}_abssum"] + 2e-6 + 1.1e-5 * step
        assert not bad.any(), (step, [names[i] for i in np.nonzero(bad)[0]])
        for n in names:
            t = sd[n].reshape(-1)
            # Adam's normalised update is +-lr (5e-6) per step wherever |g| >> eps and anything in between where the gradient is
            # rounding noise (|g| ~ 1e-8, e.g. a bias whose row contributions cancel): there the sign of g, hence the direction of the
            # lr-sized step, is not determined in fp32 -> absolute tolerance 2 * lr (lr = 5e-6) per accumulated step
            np.testing.assert_allclose((t if t.numel() <= 4096 else t[::97]).cpu().numpy(), g[f"s{step}_param/{n}"], rtol=1e-4, atol=1.1e-5 * step, err_msg=f"step {step} {n}")
    # nothing outside the estimator moved
    for n in ("observation_decoder.3.weight", "latent_dynamics.network.0.weight"):
        assert torch.equal(ai.state_dict()[n].cpu(), p[n])
