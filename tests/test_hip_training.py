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
