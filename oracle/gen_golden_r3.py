#!/usr/bin/env python3
"""Golden vectors for the surface added in round 3, produced by running the REFERENCE (imported read-only from /root/reference) under the
shared NoiseStream; container-only, same stubs / RNG tape as gen_golden.py (which it imports).  Every vector is also compared against
`oracle.aid_oracle` while it is generated (a mismatch aborts) - that is what pins the restatement of these functions.

  * `compute_lambda_returns(exclude_immediate_rewards=True)`            active_inference.py:638-707
  * `_compute_gradient_penalty` as a stand-alone value                   active_inference.py:709-729   (train / eval, continuous / discrete t)
  * `generate_latent_trajectory(deterministic=True)`                     diffusion.py:176-206, 232-233
  * pixel-mode `update_belief_via_diffusion` (feature-decoder MSE, whole trajectory)   active_inference.py:256-312

Usage: python oracle/gen_golden_r3.py  -> tests/golden/r3_surface.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aid_oracle as orc  # noqa: E402
from oracle import gen_golden as gg  # noqa: E402
from oracle.noise import NoiseStream  # noqa: E402
from oracle.weights import draw_feature_decoder, draw_params, param_specs  # noqa: E402


def main():
    torch.set_num_threads(4)
    Box = gg.install_stubs()
    sys.path.insert(0, gg.REF)
    tape = gg.Tape(); tape.install()
    import active_inference_diffusion as aid
    from active_inference_diffusion.core.active_inference import DiffusionActiveInference

    T = torch.from_numpy
    g = {}
    # ------------------------------------------------------------------------------------------ state mode
    O, A, D, N, B, KEY = 17, 6, 64, 5, 8, 3101
    cfg = aid.ActiveInferenceConfig(device="cpu", batch_size=B, efe_horizon=2, latent_dim=O, hidden_dim=D)
    cfg.diffusion = aid.DiffusionConfig(num_diffusion_steps=N)
    tape.use(999)
    ai = DiffusionActiveInference(observation_dim=O, action_dim=A, latent_dim=O, config=cfg)
    sd = ai.state_dict()
    specs = param_specs(O, A, O, D, N)
    p = draw_params(specs, KEY, orc.schedule_tables(N))
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd if k.startswith("epistemic_estimator.decoder.")}},
                       strict=True)
    tab = orc.schedule_tables(N)
    inp = NoiseStream(3102)
    obs, z, t_cont = inp.randn(B, O), inp.randn(B, O), inp.rand(B)
    g.update(dims=np.array([O, A, D, N, B]), weights_key=KEY, obs=obs, z=z, t_cont=t_cont)

    # ---- lambda-returns without the immediate reward: two batch lengths (one shorter than the window), dones at window edges
    for tag, n, n_steps in (("a", 12, 5), ("b", 4, 5), ("c", 9, 3)):
        r, nv = inp.randn(n), inp.randn(n)
        d = inp.rand(n) < 0.25
        d[-1] = tag == "a"
        ref = ai.compute_lambda_returns(T(r), torch.zeros(n), T(nv), T(d), lambda_=0.95, n_steps=n_steps, exclude_immediate_rewards=True)
        mine = orc.lambda_returns(r, nv, d, cfg.discount_factor, 0.95, n_steps, exclude_immediate_rewards=True)
        assert np.array_equal(mine, ref.numpy()), (tag, mine, ref)          # bit-exact
        ref_incl = ai.compute_lambda_returns(T(r), torch.zeros(n), T(nv), T(d), lambda_=0.95, n_steps=n_steps)
        assert np.array_equal(orc.lambda_returns(r, nv, d, cfg.discount_factor, 0.95, n_steps), ref_incl.numpy())
        g.update({f"lam_{tag}_r": r, f"lam_{tag}_nv": nv, f"lam_{tag}_d": d, f"lam_{tag}_n_steps": n_steps, f"lam_{tag}_excl": ref.numpy(),
                  f"lam_{tag}_incl": ref_incl.numpy()})
    g["gamma"] = np.float64(cfg.discount_factor)

    # ---- gradient penalty as a value
    for tag, train, tt in (("train_cont", True, T(t_cont)), ("eval_cont", False, T(t_cont)), ("eval_disc", False, torch.full((B,), 3.0))):
        ai.train(train)
        tape.use(3103)
        ref = ai._compute_gradient_penalty(T(z), tt, T(obs)).detach()
        mine = orc.gradient_penalty(p, T(z), tt, T(obs), NoiseStream(3103), train).detach()
        gg.close(mine, ref, 1e-5, 1e-7, f"gradient penalty {tag}")
        g[f"gp_{tag}"] = np.float64(ref)
    g["gp_key"] = 3103

    # ---- deterministic reverse trajectory (posterior means only), eval mode
    ai.eval(); tape.use(3104)
    with torch.no_grad():
        traj = ai.latent_diffusion.generate_latent_trajectory(ai.latent_score_network, B, T(obs), deterministic=True)
    zf, tr = orc.generate_latent_trajectory(p, tab, T(obs), NoiseStream(3104), False, keep=True, deterministic=True)
    for a_, b_ in zip(tr, traj):
        gg.close(a_, b_, 1e-5, 1e-6, "deterministic trajectory")
    g.update(det_key=3104, det_traj=np.stack([x.numpy() for x in traj]))
    ai.train()

    # ------------------------------------------------------------------------------------------ pixel mode (features in, no encoder / ConvDecoder needed)
    L, Dp, Np, Bp, KEYP = 16, 64, 4, 6, 3201
    cfgp = aid.ActiveInferenceConfig(device="cpu", batch_size=Bp, efe_horizon=1, latent_dim=L, hidden_dim=Dp, pixel_observation=True)
    cfgp.diffusion = aid.DiffusionConfig(num_diffusion_steps=Np)
    tape.use(999)
    aip = DiffusionActiveInference(observation_dim=L, action_dim=4, latent_dim=L, config=cfgp, pixel_shape=(9, 84, 84))   # the ConvDecoder it builds is not used here
    pp = draw_params(param_specs(L, 4, L, Dp, Np), KEYP, orc.schedule_tables(Np))
    pp = {k: v for k, v in pp.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}
    pp.update(draw_feature_decoder(L, Dp, KEYP + 1))
    sdp = aip.state_dict()
    missing = sorted(set(sdp) - set(pp))
    assert all(k.startswith(("observation_decoder.", "epistemic_estimator.")) for k in missing), missing[:5]
    aip.load_state_dict({**{k: sdp[k] for k in missing}, **pp}, strict=True)
    tabp = orc.schedule_tables(Np)
    feats = np.tanh(NoiseStream(3202).randn(Bp, L)).astype(np.float32)       # encoder features live in (-1, 1): the encoder ends in tanh
    g.update(pix_dims=np.array([L, 4, Dp, Np, Bp]), pix_weights_key=KEYP, pix_fdec_key=KEYP + 1, pix_feats=feats)
    for tag, train, key in (("train", True, 3203), ("eval", False, 3204)):
        aip.train(train); tape.use(key)
        with torch.no_grad():
            bi = aip.update_belief_via_diffusion(T(feats))
        mine = orc.update_belief_via_diffusion(pp, tabp, T(feats), NoiseStream(key), train, pixel=True, keep=True)
        gg.close(mine["latent"], bi["latent"], 1e-4, 1e-6, f"pixel belief latent {tag}")
        gg.close(mine["reconstruction_error"], bi["reconstruction_error"], 1e-5, 1e-7, f"pixel belief reconstruction {tag}")
        gg.close(mine["latent_std"], bi["latent_std"], 1e-4, 1e-6, f"pixel belief std {tag}")
        assert bi["trajectory_length"] == Np + 1 == len(aip.latent_trajectory)
        for a_, b_ in zip(mine["trajectory"], aip.latent_trajectory):
            gg.close(a_, b_, 1e-4, 1e-6, f"pixel belief trajectory {tag}")
        g.update({f"pix_{tag}_key": key, f"pix_{tag}_latent": bi["latent"].numpy(), f"pix_{tag}_rec": np.float64(bi["reconstruction_error"]),
                  f"pix_{tag}_mean": bi["latent_mean"].numpy(), f"pix_{tag}_std": bi["latent_std"].numpy(),
                  f"pix_{tag}_traj": np.stack([x.numpy() for x in aip.latent_trajectory])})
    out = os.path.join(ROOT, "tests", "golden", "r3_surface.npz")
    np.savez_compressed(out, **g)
    print("oracle == reference on every vector; wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
