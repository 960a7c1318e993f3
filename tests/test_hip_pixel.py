"""GPU parity of the pixel path (A13): DrQ-v2 encoder forward/backward incl. legacy spectral norm, random-shift augmentation,
against the reference's own outputs/gradients (tests/golden/pixel.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import aid_oracle as orc
from oracle.encoder_weights import draw_encoder_params, encoder_specs
from oracle.noise import NoiseStream


@pytest.fixture(scope="module")
def nat():
    from aid_amd import _native
    _native.lib()
    return _native


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda().contiguous()


def test_encoder_forward_backward_golden(nat, golden_dir):
    g = np.load(os.path.join(golden_dir, "pixel.npz"))
    FD, B = int(g["feature_dim"]), int(g["B"])
    specs = encoder_specs(FD)
    p = draw_encoder_params(specs, int(g["weights_key"]))
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    enc, genc = nat.encoder_table(pd, FD), nat.encoder_table(gd, FD)
    x = dev(g["x"]).reshape(B, 9, 84, 84)
    # eval forward (no power iteration, no masks)
    f, _ = nat.encoder_forward(enc, x, train=False)
    print(f"encoder golden: max|feat_eval - reference| = {float(np.abs(f.cpu().numpy() - g['feat_eval']).max()):.3e}")
    np.testing.assert_allclose(f.cpu().numpy(), g["feat_eval"], rtol=1e-4, atol=1e-5)
    u0 = pd["convs.0.weight_u"].clone()
    assert torch.equal(u0, p["convs.0.weight_u"].cuda())           # eval leaves u/v untouched
    cot = NoiseStream(int(g["cotangent_key"])).randn(2, B, FD)
    # two training passes: pass 0 on the augmented input (shifts from the reference's draw order), pass 1 on the raw input
    ns = NoiseStream(int(g["aug_key"]))
    sh = torch.from_numpy(ns.randint(0, 9, B).astype(np.int32)).cuda(); sw = torch.from_numpy(ns.randint(0, 9, B).astype(np.int32)).cuda()
    for pas in (0, 1):
        for v in gd.values():
            v.zero_()
        ms = NoiseStream(int(g[f"train{pas}_key"]))
        m1, m2, mo = dev(ms.dropout_mask((B, 64), 0.025)), dev(ms.dropout_mask((B, 128), 0.05)), dev(ms.dropout_mask((B, 2 * FD), 0.1))
        f, saved = nat.encoder_forward(enc, x, sh if pas == 0 else None, sw if pas == 0 else None, 4, m1, m2, mo, train=True)
        print(f"encoder golden: max|feat_train{pas} - reference| = {float(np.abs(f.cpu().numpy() - g[f'feat_train{pas}']).max()):.3e}")
        np.testing.assert_allclose(f.cpu().numpy(), g[f"feat_train{pas}"], rtol=1e-4, atol=1e-5, err_msg=f"pass {pas}")
        for i in range(4):
            np.testing.assert_allclose(pd[f"convs.{i}.weight_u"].cpu().numpy(), g[f"u{pas}/{i}"], rtol=1e-4, atol=1e-6)
        nat.encoder_backward(enc, genc, dev(cot[pas]), saved, m1, m2, mo, train=True)
        gerr = 0.0
        for k in g.files:
            if k.startswith(f"grad{pas}/"):
                name = k.split("/", 1)[1]; ref = g[k]
                got = gd[name].cpu().numpy()
                scale = np.abs(ref).max() + 1e-12
                gerr = max(gerr, np.abs(got - ref).max() / scale)
                assert np.abs(got - ref).max() / scale < 1e-3, (pas, name, np.abs(got - ref).max(), scale)
                assert np.linalg.norm((got - ref).ravel()) / (np.linalg.norm(ref.ravel()) + 1e-300) < 2e-4, (pas, name)
            if k.startswith(f"gradsample{pas}/"):
                name = k.split("/", 1)[1]; ref = g[k]
                got = gd[name].reshape(-1)[::9973].cpu().numpy()
                scale = np.abs(ref).max() + 1e-12
                gerr = max(gerr, np.abs(got - ref).max() / scale)
                assert np.abs(got - ref).max() / scale < 1e-3, (pas, name)
                s_ref = g[f"gradsum{pas}/{name}"]
                np.testing.assert_allclose(float(gd[name].double().abs().sum()), s_ref[1], rtol=1e-3)
        print(f"encoder golden pass {pas}: worst gradient max|err| / max|ref| = {gerr:.3e}")


def test_random_shift_matches_oracle(nat, golden_dir):
    """Augmentation alone: shift via the encoder's first stage is covered above; here the checksum of the reference's output."""
    g = np.load(os.path.join(golden_dir, "pixel.npz"))
    B = int(g["B"])
    x4 = torch.from_numpy(g["x"]).view(B, 9, 84, 84)
    xa = orc.random_shift(x4, 4, NoiseStream(int(g["aug_key"])))
    np.testing.assert_allclose(float(xa.double().sum()), float(g["aug_checksum"]), rtol=1e-12)


def test_encoder_batch16_vs_oracle(nat):
    """A larger batch (spans several GEMM tiles / slabs) against the oracle's autograd."""
    FD, B = 32, 16
    p = draw_encoder_params(encoder_specs(FD), 99)
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    enc, genc = nat.encoder_table(pd, FD), nat.encoder_table(gd, FD)
    ns = NoiseStream(5)
    x = ns.rand(B, 9, 84, 84); cot = ns.randn(B, FD)
    ms = NoiseStream(6)
    masks = [ms.dropout_mask((B, 64), 0.025), ms.dropout_mask((B, 128), 0.05), ms.dropout_mask((B, 2 * FD), 0.1)]
    f, saved = nat.encoder_forward(enc, dev(x), None, None, 4, dev(masks[0]), dev(masks[1]), dev(masks[2]), train=True)
    nat.encoder_backward(enc, genc, dev(cot), saved, dev(masks[0]), dev(masks[1]), dev(masks[2]), train=True)
    gn = [k for k in p if not k.endswith(("weight_u", "weight_v")) and "channel_reduce" not in k]
    for k in gn:
        p[k].requires_grad_(True)
    torch.set_num_threads(8)
    fr = orc.encoder_forward(p, torch.from_numpy(x), NoiseStream(6), True, update_uv=False)
    gr = torch.autograd.grad((fr * torch.from_numpy(cot)).sum(), [p[k] for k in gn])
    # the same function in float64: how far the fp32 oracle (= the reference's own arithmetic) and the HIP path each sit from the exact value - the
    # oracle's distance is the floor no fp32 implementation can be held under when compared with it
    p64 = {k: v.detach().double().requires_grad_(k in gn) for k, v in p.items()}
    f64 = orc.encoder_forward(p64, torch.from_numpy(x).double(), NoiseStream(6), True, update_uv=False)
    g64 = torch.autograd.grad((f64 * torch.from_numpy(cot).double()).sum(), [p64[k] for k in gn])
    fh = f.cpu().double()
    e_hip, e_orc, e_ho = float((fh - f64).abs().max()), float((fr.double() - f64).abs().max()), float((fh - fr.double()).abs().max())
    print(f"encoder B=16 features: max|hip - fp64| = {e_hip:.3e}, max|oracle32 - fp64| = {e_orc:.3e}, max|hip - oracle32| = {e_ho:.3e}")
    nw = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
    worst = sorted(((nw(gd[k].cpu().double(), r.double()), nw(gd[k].cpu().double(), r64), nw(r.double(), r64),
                     float((gd[k].cpu() - r).abs().max()) / (float(r.abs().max()) + 1e-12), k) for k, r, r64 in zip(gn, gr, g64)), reverse=True)
    print("encoder B=16 gradients, norm-wise (hip vs oracle32, hip vs fp64, oracle32 vs fp64, max-abs/max-ref vs oracle32):",
          [(f"{a:.2e}", f"{b:.2e}", f"{c:.2e}", f"{d:.2e}", k) for a, b, c, d, k in worst[:6]])
    # Features are tanh(LN(.)) in [-1, 1].  Against the exact (float64) value the HIP path and the float32 reference both sit at ~4e-6 (printed above): the
    # distance BETWEEN two float32 evaluations cannot be held under the sum of their distances from the truth.  So: (1) the HIP path is at least as exact
    # as the reference's own arithmetic (within 1.5x + 1e-6), (2) HIP vs the reference at rtol 1e-4 (SURVEY 8c's forward tolerance) with the absolute
    # floor 1e-5 that (1) implies instead of 1e-6.
    assert e_hip <= 1.5 * e_orc + 1e-6, (e_hip, e_orc)
    np.testing.assert_allclose(f.cpu().numpy(), fr.detach().numpy(), rtol=1e-4, atol=1e-5)
    for a, b, c, d, k in worst:     # gradients: norm-wise 1e-4 vs the reference or no further from the truth than the reference itself; max-abs 1e-3
        assert a < 1e-4 or b <= 1.5 * c + 1e-6, (k, a, b, c)
        assert d < 1e-3, (k, d)


# ------------------------------------------------------------------------------------------------- pixel-mode training pieces
from oracle.weights import draw_params, param_specs
from parity_noise import ParityNoise
from tolerances import assert_metric


def test_infonce_golden_and_oracle(nat, golden_dir):
    """compute_representation_loss against the reference's own loss / d(loss)/d(next features) (tests/golden/pixel.npz)."""
    g = np.load(os.path.join(golden_dir, "pixel.npz"))
    L, A, D = 17, 6, 64
    p = draw_params(param_specs(L, A, L, D, 4), 1234 + L, orc.schedule_tables(4))      # the weights gen_golden.py gave the reference
    dyn = nat.dynamics_table({k: v.cuda().contiguous() for k, v in p.items()}, A)
    lat, act, nxt = dev(g["nce_lat"]), dev(g["nce_act"]), dev(g["nce_next"])
    tn = torch.from_numpy(g["nce_next"]).requires_grad_(True)
    ref = orc.representation_loss(p, tn, torch.from_numpy(g["nce_act"]), torch.from_numpy(g["nce_lat"]))
    (gref,) = torch.autograd.grad(ref, tn)
    d_next = torch.zeros_like(nxt)
    loss = nat.infonce_step(dyn, lat, act, nxt, 0.5, d_next)
    np.testing.assert_allclose(float(ref), float(g["nce_loss"]), rtol=1e-5)                 # oracle == reference
    np.testing.assert_allclose(float(loss), float(g["nce_loss"]), rtol=2e-4)
    np.testing.assert_allclose(d_next.cpu().numpy(), 0.5 * g["nce_grad"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(d_next.cpu().numpy(), 0.5 * gref.numpy(), rtol=2e-3, atol=1e-6)


def _pixel_agent(L, A, D, N, H, B, cap=32):
    import aid_amd
    from test_hip_agent import Box
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=H, latent_dim=L, hidden_dim=D, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)

    class Env:
        observation_space, action_space = Box((3, 3, 84, 84)), Box((A,))
    return aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=cap), aid_amd.PixelObservationConfig())


def _load_pixel_params(agent, L, A, D, N, key):
    """Oracle-drawn weights (non-zero adaLN / heads) for everything the oracle knows; the feature decoder keeps its own init."""
    ai = agent.active_inference
    p = draw_params(param_specs(L, A, L, D, N), key, orc.schedule_tables(N))
    p = {k: v for k, v in p.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}     # state-mode shapes
    sd = ai.state_dict()
    full = {k: (p[k] if k in p else v.cpu().clone()) for k, v in sd.items()}
    ai.load_state_dict(full, strict=True)
    full = {k: v for k, v in full.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}   # not modelled by these oracle calls
    pe = draw_encoder_params(encoder_specs(L), key + 1)
    agent.encoder.load_state_dict(pe, strict=True)
    merged = {k: v.clone() for k, v in full.items()}
    merged.update({"encoder." + k: v.clone() for k, v in pe.items()})
    return merged


def test_pixel_elbo_feature_gradient_vs_oracle(nat):
    """ELBO with the feature-decoder reconstruction: loss, feature-decoder gradients and d(loss)/d(encoder features) - the
    cotangent the encoder backward consumes (recon target + score-net conditioning incl. the gradient penalty's double backward)."""
    L, A, D, N, B = 32, 6, 64, 4, 8
    agent = _pixel_agent(L, A, D, N, 2, B)
    ai = agent.active_inference
    p = _load_pixel_params(agent, L, A, D, N, 41)
    ns = NoiseStream(12)
    feats, lat, rew = np.tanh(ns.randn(B, L)).astype(np.float32), ns.randn(B, L), ns.randn(B)
    pn = ParityNoise(77, pixel=True); ai.noise = pn
    for a in ai.arenas.values():
        a.zero_grad()
    loss, info = ai.compute_diffusion_elbo(dev(feats), dev(rew), dev(lat))
    for n in p:
        p[n].requires_grad_(orc.is_param(n) and p[n].dtype.is_floating_point and not n.startswith("encoder."))
    ft = torch.from_numpy(feats).requires_grad_(True)
    lref, iref, _, _ = orc.diffusion_elbo(p, ft, torch.from_numpy(rew), torch.from_numpy(lat), NoiseStream(77), None, True, pixel=True)
    names = [n for n in p if n.startswith(("feature_decoder.", "latent_score_network.obs_encoder.0"))]
    gref = torch.autograd.grad(lref, [ft] + [p[n] for n in names])
    np.testing.assert_allclose(float(loss), float(lref), rtol=2e-3)
    np.testing.assert_allclose(float(info["reconstruction_loss"]), float(iref["reconstruction_loss"]), rtol=1e-4)
    sc = float(gref[0].abs().max())
    assert float((ai._last_d_obs.cpu() - gref[0]).abs().max()) / sc < 3e-3
    gv = {**ai.arenas["feature_decoder"].grad_views(), **ai.arenas["score"].grad_views()}
    for n, r in zip(names, gref[1:]):
        assert float((gv[n].cpu() - r).abs().max()) / (float(r.abs().max()) + 1e-12) < 3e-3, n


def test_pixel_agent_train_step_vs_oracle():
    """One whole DiffusionPixelAgent.train_step (encoder x2, merged belief loop, ELBO + InfoNCE, two encoder backward passes,
    shared-norm clip over score + diffusion + encoder, policy / value / dynamics phases) against the oracle's restatement."""
    L, A, D, N, H, B = 32, 6, 64, 3, 2, 8
    agent = _pixel_agent(L, A, D, N, H, B)
    ai = agent.active_inference
    p = _load_pixel_params(agent, L, A, D, N, 17)
    bs = NoiseStream(4); cap = 16
    store = {"observations": bs.rand(cap, 9 * 84 * 84), "actions": np.tanh(bs.randn(cap, A)).astype(np.float32), "rewards": bs.randn(cap),
             "next_observations": bs.rand(cap, 9 * 84 * 84), "dones": bs.rand(cap) < 0.1}
    agent.replay_buffer.add_batch(store["observations"], store["actions"], store["rewards"], store["next_observations"], store["dones"])
    pn = ParityNoise(321, pixel=True); ai.noise = pn; agent.replay_buffer.index_source = pn
    m = agent.train_step()
    st = orc.OraclePixelState(p, N)
    ns = NoiseStream(321)
    batch = orc.replay_sample(store, cap, B, ns)
    batch["observations"] = batch["observations"].reshape(B, 9, 84, 84); batch["next_observations"] = batch["next_observations"].reshape(B, 9, 84, 84)
    torch.set_num_threads(8)
    m_ref, g_ref, aux = orc.pixel_train_step(st, batch, ns, horizon=H, return_grads=True)
    for k_ in ("encoded_obs", "latents", "next_latents"):
        print(f"pixel step B=8 achieved: max|{k_} - oracle| = {float((agent._last[k_].cpu() - aux[k_]).abs().max()):.3e}")
    np.testing.assert_allclose(agent._last["encoded_obs"].cpu().numpy(), aux["encoded_obs"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(agent._last["latents"].cpu().numpy(), aux["latents"].numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(agent._last["next_latents"].cpu().numpy(), aux["next_latents"].numpy(), rtol=1e-3, atol=1e-5)
    for k, v in m_ref.items():
        if isinstance(v, float) and k in m:
            assert_metric(k, float(m[k]), v)
    assert "contrastive_loss" in m
    sd = {**ai.state_dict(), **{"encoder." + k: v for k, v in agent.encoder.state_dict().items()}}
    for n in st.p:
        ref = st.p[n].detach().numpy()
        np.testing.assert_allclose(sd[n].cpu().numpy(), ref, rtol=1e-3, atol=3e-6, err_msg=n)
    # the update really moved the encoder and the feature decoder; channel_reduce stays put
    assert not np.allclose(sd["encoder.convs.0.weight_orig"].cpu().numpy(), p["encoder.convs.0.weight_orig"].detach().numpy(), atol=1e-7)
    assert not np.allclose(sd["feature_decoder.0.weight"].cpu().numpy(), p["feature_decoder.0.weight"].detach().numpy(), atol=1e-7)


def test_pixel_agent_philox_run():
    """Production mode (device Philox, augmentation on): metric keys, finite values, act() from raw uint8 frames."""
    L, A = 32, 4
    agent = _pixel_agent(L, A, 64, 3, 2, 8)
    rng = np.random.default_rng(0)
    assert agent.train_step() == {}
    for _ in range(16):
        agent.replay_buffer.add(rng.random((3, 3, 84, 84), dtype=np.float32), rng.uniform(-1, 1, A), float(rng.standard_normal()),
                                rng.random((3, 3, 84, 84), dtype=np.float32), False)
    w0 = agent.encoder.convs[1].weight_orig.detach().clone()
    for _ in range(2):
        m = agent.train_step()
    want = {"reconstruction_loss", "kl_loss", "score_matching_loss", "elbo", "reward_loss", "grad_penalty", "contrastive_loss", "policy_loss",
            "pragmatic_mean", "consistency_mean", "value_loss", "dynamics_loss"}
    assert want <= set(m) and all(np.isfinite(v) for v in m.values()), m
    assert not torch.equal(w0, agent.encoder.convs[1].weight_orig)
    action, info = agent.act((rng.random((3, 84, 84, 3)) * 255).astype(np.uint8))
    assert action.shape == (A,) and np.all(np.isfinite(action)) and np.all(np.abs(action) <= 1)
    import aid_amd.agents as ag
    assert ag.PixelBasedAgent is ag.DiffusionPixelAgent


def test_pixel_agent_train_step_reference_golden(golden_dir):
    """One whole DiffusionPixelAgent.train_step against the REFERENCE's own post-step state (tests/golden/pixel_step.npz, written by
    oracle/gen_golden_pixel_step.py from a run of the reference) WITH the pixel estimator evaluated inside the EFE: every parameter of
    every optimiser group (checksums + selected tensors), every metric incl. policy_loss and epistemic/*, and the MINE running mean."""
    import functools
    from oracle.encoder_weights import conv_decoder_specs, draw_conv_decoder_params, draw_pixel_estimator_params, pixel_estimator_specs
    from oracle.weights import draw_feature_decoder
    g = np.load(os.path.join(golden_dir, "pixel_step.npz"))
    L, A, D, N, H, B, K, S, CAP = (int(v) for v in g["dims"])
    kw, ke, kf, ks, kstep, kdec, kest = (int(v) for v in g["keys"])
    import aid_amd
    from test_hip_agent import Box
    cfg = aid_amd.ActiveInferenceConfig(device="cuda", batch_size=B, efe_horizon=H, latent_dim=L, hidden_dim=D, pixel_observation=True)
    cfg.diffusion = aid_amd.DiffusionConfig(num_diffusion_steps=N)

    class Env:
        observation_space, action_space = Box((3, 3, 84, 84)), Box((A,))
    agent = aid_amd.DiffusionPixelAgent(Env(), cfg, aid_amd.TrainingConfig(buffer_size=CAP), aid_amd.PixelObservationConfig())
    ai = agent.active_inference
    p = draw_params(param_specs(L, A, L, D, N), kw, orc.schedule_tables(N))
    p = {k: v for k, v in p.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}
    p.update(draw_feature_decoder(L, D, kf))
    pdec = draw_conv_decoder_params(conv_decoder_specs(L, hidden=D), kdec)
    p.update({"observation_decoder." + k: v for k, v in pdec.items()}); p.update({"epistemic_estimator.decoder." + k: v for k, v in pdec.items()})
    p.update({"epistemic_estimator." + k: v for k, v in draw_pixel_estimator_params(pixel_estimator_specs(L), kest).items()})
    sd = ai.state_dict()
    assert set(p) == set(sd), sorted(set(p) ^ set(sd))[:6]          # the reference's pixel-mode state_dict layout, key for key
    ai.load_state_dict(p, strict=True)
    agent.encoder.load_state_dict(draw_encoder_params(encoder_specs(L), ke), strict=True)
    bs = NoiseStream(ks)
    obs = bs.rand(CAP, 3, 3, 84, 84); act = np.tanh(bs.randn(CAP, A)).astype(np.float32); rew = bs.randn(CAP); nobs = bs.rand(CAP, 3, 3, 84, 84)
    for i in range(CAP):
        agent.replay_buffer.add(obs[i], act[i], float(rew[i]), nobs[i], False)
    pn = ParityNoise(kstep, pixel=True); pn.live_epistemic_samples = S
    ai.noise = pn; agent.replay_buffer.index_source = pn
    ai.pixel_epistemic = True
    ai.compute_expected_free_energy_diffusion = functools.partial(ai.compute_expected_free_energy_diffusion, num_trajectories=K, num_ambiguity_samples=S)
    m = agent.train_step()
    for k in g.files:
        if k.startswith("metric/"):
            name = k.split("/", 1)[1]
            assert name in m, name
            np.testing.assert_allclose(m[name], float(g[k]), rtol=5e-3, atol=5e-4 if name.startswith("epistemic") else 5e-5, err_msg=name)
    got = {**ai.state_dict(), **{"encoder." + k: v for k, v in agent.encoder.state_dict().items()}}
    np.testing.assert_allclose(float(got["epistemic_estimator.running_mean"]), float(g["running_mean"]), rtol=5e-3)
    for k in g.files:
        if k.startswith("param/"):
            n = k.split("/", 1)[1]
            np.testing.assert_allclose(got[n].cpu().numpy(), g[k], rtol=1e-3, atol=3e-6, err_msg=n)
    np.testing.assert_allclose(got["encoder.output_layers.0.weight"].reshape(-1)[::9973].cpu().numpy(), g["param_sample/encoder.output_layers.0.weight"],
                               rtol=1e-3, atol=3e-6)
    names = [str(n) for n in g["names"]]
    cs = np.array([float(got[n].double().sum()) for n in names])
    bad = np.abs(cs - g["checksum"]) > 2e-5 * g["abssum"] + 2e-5
    assert not bad.any(), [names[i] for i in np.nonzero(bad)[0]]


def test_conv_decoder_forward_reference_golden(nat, golden_dir):
    """ConvDecoder.forward (eval mode: spectral-normed 3x3 / 1x1 convs, PixelShuffle, InstanceNorm, Mish, sigmoid) against the
    reference's own images (tests/golden/conv_decoder.npz) and the oracle."""
    from oracle.encoder_weights import conv_decoder_specs, draw_conv_decoder_params
    g = np.load(os.path.join(golden_dir, "conv_decoder.npz"))
    L, R = (int(v) for v in g["dims"]); kw, kz = (int(v) for v in g["keys"])
    p = draw_conv_decoder_params(conv_decoder_specs(L), kw)
    pd = {"observation_decoder." + k: v.cuda().contiguous() for k, v in p.items()}
    z = NoiseStream(kz).randn(R, L)
    y = nat.conv_decoder_forward(nat.conv_decoder_table(pd), dev(z), chunk=2).cpu()       # chunk 2: exercises the row chunking (2 + 1)
    assert tuple(y.shape) == (R, 3, 168, 168)
    np.testing.assert_allclose(y.reshape(-1)[::61].numpy(), g["sample"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(y[1, :, 80, :].numpy(), g["row"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose([float(y[i].double().sum()) for i in range(R)], g["checksum"], rtol=2e-4)


def test_pixel_decode_observation_paths():
    """decode_observation in pixel mode: ConvDecoder image (eval) == oracle; feature reconstruction == oracle's feature_decoder."""
    L, A, D, N = 16, 4, 64, 3
    agent = _pixel_agent(L, A, D, N, 1, 2, cap=4)
    ai = agent.active_inference
    sd = {k: v.cpu() for k, v in ai.state_dict().items()}
    z = NoiseStream(3).randn(3, L)
    ai.eval()
    img = ai.decode_observation(dev(z), decode_to_pixels=True)
    assert tuple(img.shape) == (3, 3, 168, 168)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = orc.conv_decoder_forward(sd, torch.from_numpy(z))
        ref_f = orc.feature_decoder(sd, torch.from_numpy(z), None, False)
    np.testing.assert_allclose(img.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(ai.decode_observation(dev(z), decode_to_pixels=False).cpu().numpy(), ref_f.numpy(), rtol=1e-3, atol=2e-5)
    ai.train()
    with pytest.raises(NotImplementedError):
        ai.decode_observation(dev(z), decode_to_pixels=True)


def test_pixel_epistemic_value_reference_golden(nat, golden_dir):
    """Pixel branch of the epistemic estimator (ConvDecoder finite differences -> 5x5 stride-2 convs -> 16-query multi-head attention
    aggregator -> projector / latent processor / MINE statistics) against two calls of the REFERENCE's estimator."""
    from oracle.encoder_weights import conv_decoder_specs, draw_conv_decoder_params, draw_pixel_estimator_params, pixel_estimator_specs
    g = np.load(os.path.join(golden_dir, "pixel_estimator.npz"))
    L, B, S = (int(v) for v in g["dims"]); kd, ke, kz, k1, k2 = (int(v) for v in g["keys"])
    pd = {"observation_decoder." + k: v.cuda().contiguous() for k, v in draw_conv_decoder_params(conv_decoder_specs(L), kd).items()}
    pe = {"epistemic_estimator." + k: v.cuda().contiguous() for k, v in draw_pixel_estimator_params(pixel_estimator_specs(L), ke).items()}
    dec, epi = nat.conv_decoder_table(pd), nat.pixel_epistemic_table(pe)
    nm = dev(NoiseStream(kz).randn(B, L))
    rm = 0.0
    for call, key in ((1, k1), (2, k2)):
        pn = ParityNoise(key, pixel=True)
        mine = nat.pixel_epistemic_value(dec, epi, nm, 1, S, noise=pn.pixel_estimator(B, L, S), train=True).cpu().numpy()[0]
        joint, tlog, texp, mi = (float(v) for v in mine)
        rm = texp if rm == 0.0 else 0.01 * texp + 0.99 * rm
        np.testing.assert_allclose(mi, float(g[f"c{call}/epistemic/mi_estimate"]), rtol=5e-3, atol=2e-4)
        np.testing.assert_allclose(joint, float(g[f"c{call}/epistemic/joint_term"]), rtol=5e-3, atol=2e-4)
        np.testing.assert_allclose(tlog, float(g[f"c{call}/epistemic/marginal_term"]), rtol=5e-3, atol=2e-4)
        np.testing.assert_allclose(rm, float(g[f"c{call}/epistemic/running_mean"]), rtol=2e-3)


def test_pixel_score_step_waits_for_the_next_obs_belief_chain():
    """The default pixel schedule runs the belief over next_obs (chain b) on a chain stream underneath the ELBO, InfoNCE and the encoder backward passes,
    reading the score-network weights in place; score_optimizer.step() on the side stream rewrites them.  The reference computes next_latents before that
    step (pixel_agent.py:307 vs :332), so the step must be ordered behind chain b.  Make chain b LONG (400 denoise steps at 160 rows: ~70 ms, longer than
    everything the side stream does before the step) and compare with the schedule that has no concurrent chain (early_belief_chains off: the paired loop
    finishes before the ELBO): same next_latents, same post-step tensors, bit for bit (the rows of a loop never interact, both forms run the same kernels)."""
    L, A, D, N, H, B, cap = 32, 6, 64, 400, 1, 160, 192
    outs = []
    for early in (True, False):
        torch.manual_seed(0)          # the feature decoder keeps torch's default initialisation: same seed, same weights in both agents
        agent = _pixel_agent(L, A, D, N, H, B, cap=cap)
        ai = agent.active_inference
        _load_pixel_params(agent, L, A, D, N, 23)
        bs = NoiseStream(9)
        agent.replay_buffer.add_batch(bs.rand(cap, 9 * 84 * 84), np.tanh(bs.randn(cap, A)).astype(np.float32), bs.randn(cap), bs.rand(cap, 9 * 84 * 84), bs.rand(cap) < 0.1)
        pn = ParityNoise(555, pixel=True); ai.noise = pn; agent.replay_buffer.index_source = pn
        agent.early_belief_chains = early
        agent.train_step(); torch.cuda.synchronize()
        sd = {**ai.state_dict(), **{"encoder." + k: v for k, v in agent.encoder.state_dict().items()}}
        outs.append(({k: v.detach().cpu().clone() for k, v in sd.items()}, {k: v.detach().cpu().clone() for k, v in agent._last.items()}))
        del agent, ai; torch.cuda.empty_cache()
    (sd_a, last_a), (sd_b, last_b) = outs
    for k in ("latents", "next_latents"):
        assert torch.equal(last_a[k], last_b[k]), (k, float((last_a[k] - last_b[k]).abs().max()))
    diff = [k for k in sd_a if not torch.equal(sd_a[k], sd_b[k])]
    assert not diff, diff[:8]


def test_pixel_train_step_with_the_estimator_at_default_K_S_vs_oracle():
    """The pixel train_step the reference actually executes: its ConvDecoder-based estimator evaluated inside every EFE call
    (core/active_inference.py:362-366 -> 398-415 -> 865-883, 977-983) at the DEFAULT K = 10 trajectories and S = 10 ambiguity samples.  The oracle's CPU cost is
    ~27 ms per decoded 3 x 168 x 168 image and a step decodes 5 K S H B of them, so the batch and the horizon are what is reduced here (B = 2, H = 2: 2 000
    images, ~60 s of oracle time on 16 cores); `ai.pixel_epistemic = True` is the switch a user of the reference's default behaviour sets.  Checked: every metric
    incl. policy_loss, epistemic_mean and epistemic/*, the MINE running mean, and every post-step tensor."""
    import functools
    from oracle.encoder_weights import conv_decoder_specs, draw_conv_decoder_params, draw_pixel_estimator_params, pixel_estimator_specs
    from oracle.weights import draw_feature_decoder
    L, A, D, N, H, B, K, S, CAP = 16, 4, 64, 3, 2, 2, 10, 10, 4
    agent = _pixel_agent(L, A, D, N, H, B, cap=CAP)
    ai = agent.active_inference
    p = draw_params(param_specs(L, A, L, D, N), 611, orc.schedule_tables(N))
    p = {k: v for k, v in p.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}
    p.update(draw_feature_decoder(L, D, 612))
    pdec = draw_conv_decoder_params(conv_decoder_specs(L, hidden=D), 613)
    p.update({"observation_decoder." + k: v for k, v in pdec.items()}); p.update({"epistemic_estimator.decoder." + k: v for k, v in pdec.items()})
    p.update({"epistemic_estimator." + k: v for k, v in draw_pixel_estimator_params(pixel_estimator_specs(L), 614).items()})
    assert set(p) == set(ai.state_dict())
    ai.load_state_dict(p, strict=True)
    pe = draw_encoder_params(encoder_specs(L), 615)
    agent.encoder.load_state_dict(pe, strict=True)
    full = {k: v.clone() for k, v in p.items()}; full.update({"encoder." + k: v.clone() for k, v in pe.items()})
    bs = NoiseStream(616)
    store = {"observations": bs.rand(CAP, 3, 3, 84, 84), "actions": np.tanh(bs.randn(CAP, A)).astype(np.float32), "rewards": bs.randn(CAP),
             "next_observations": bs.rand(CAP, 3, 3, 84, 84), "dones": np.zeros(CAP, bool)}
    for i in range(CAP):
        agent.replay_buffer.add(store["observations"][i], store["actions"][i], float(store["rewards"][i]), store["next_observations"][i], False)
    pn = ParityNoise(617, pixel=True); pn.live_epistemic_samples = S
    ai.noise = pn; agent.replay_buffer.index_source = pn
    ai.pixel_epistemic = True
    ai.compute_expected_free_energy_diffusion = functools.partial(ai.compute_expected_free_energy_diffusion, num_trajectories=K, num_ambiguity_samples=S)
    m = agent.train_step()
    torch.cuda.synchronize()
    got = {k: v.detach().cpu() for k, v in {**ai.state_dict(), **{"encoder." + k: v for k, v in agent.encoder.state_dict().items()}}.items()}
    st = orc.OraclePixelState(full, N)
    ns = NoiseStream(617)
    batch = orc.replay_sample(dict(store), CAP, B, ns)
    batch["observations"] = batch["observations"].reshape(B, 9, 84, 84); batch["next_observations"] = batch["next_observations"].reshape(B, 9, 84, 84)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    m_ref = orc.pixel_train_step(st, batch, ns, horizon=H, num_trajectories=K, pixel_epistemic_samples=S)
    for k, v in m_ref.items():
        if isinstance(v, float):
            assert k in m, k
            np.testing.assert_allclose(m[k], v, rtol=2e-3, atol=5e-4 if k.startswith("epistemic") else 5e-5, err_msg=k)
    for k in ("epistemic_mean", "epistemic/mi_estimate", "epistemic/running_mean", "policy_loss"):
        assert k in m
    np.testing.assert_allclose(float(got["epistemic_estimator.running_mean"]), st.running_mean, rtol=5e-3)
    # every post-step tensor, with the allowance for elements whose first Adam step amplifies gradient rounding (|g| near eps: the update is lr * sign-like;
    # at B = 2 the 14 M-element head gradient has a few dozen such elements) that the headline tests use
    from test_hip_headline import _check_post_step_tensors
    names = [n for n in st.p if st.p[n].dtype.is_floating_point and not n.endswith(("weight_u", "weight_v", "epistemic_estimator.running_mean"))]
    moved = _check_post_step_tensors(got, st.p, full, names, 1, 128, tag="pixel step with the estimator, K = S = 10")
    assert {"latent_score_network", "policy_network", "value_network", "latent_dynamics", "feature_decoder", "encoder"} <= moved, moved


def test_encoder_backward_in_two_phases_equals_the_whole(nat):
    """aid_encoder_backward_phase: head phase (1) + attention / conv-stack phase (2) with a shared workspace accumulate exactly the gradients of the one-call
    backward (phase 0) - what the data-parallel pixel agent relies on to start the all-reduce of the Linear(451 584 -> 2L) gradient between the two."""
    FD, B = 32, 8
    p = draw_encoder_params(encoder_specs(FD), 51)
    pd = {k: v.cuda().contiguous() for k, v in p.items()}
    enc = nat.encoder_table(pd, FD)
    ns = NoiseStream(52)
    x, cot = dev(ns.rand(B, 9, 84, 84)), dev(ns.randn(B, FD))
    ms = NoiseStream(53)
    m1, m2, mo = dev(ms.dropout_mask((B, 64), 0.025)), dev(ms.dropout_mask((B, 128), 0.05)), dev(ms.dropout_mask((B, 2 * FD), 0.1))
    out = []
    for phases in ((0,), (1, 2)):
        for k in ("convs.0.weight_u", "convs.0.weight_v"):      # same power-iteration start for both runs
            pd[k].copy_(p[k].cuda())
        gd = {k: torch.zeros_like(v) for k, v in pd.items()}
        genc = nat.encoder_table(gd, FD)
        f, saved = nat.encoder_forward(enc, x, None, None, 4, m1, m2, mo, train=False)
        ws = nat.Workspace("cuda")
        for ph in phases:
            nat.encoder_backward(enc, genc, cot, saved, m1, m2, mo, train=False, ws=ws, phase=ph)
        torch.cuda.synchronize()
        out.append({k: v.cpu() for k, v in gd.items()})
    head = [k for k in out[0] if k.startswith(("ln.", "output_layers."))]
    assert head and all(float(out[0][k].abs().max()) > 0 for k in head)
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k
    # after phase 1 alone only the head's gradients are non-zero
    gd = {k: torch.zeros_like(v) for k, v in pd.items()}
    nat.encoder_backward(enc, nat.encoder_table(gd, FD), cot, saved, m1, m2, mo, train=False, ws=nat.Workspace("cuda"), phase=1)
    assert all(float(gd[k].abs().max()) == 0 for k in gd if k.startswith(("convs.", "norms.", "attention.")))
