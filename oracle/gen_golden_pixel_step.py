#!/usr/bin/env python3
"""Golden vectors for ONE whole DiffusionPixelAgent.train_step, produced by running the REFERENCE (imported read-only from
/root/reference) under the shared NoiseStream.  Container-only, like gen_golden.py (same stubs / RNG tape, which it imports).

Harness notes (the reference files are never edited): `total_steps = 1` skips the broken estimator-training branch (D3); the EFE call
is wrapped to `num_trajectories=2, num_ambiguity_samples=2` so that the reference's pixel estimator (ConvDecoder on S*B images, 5
decodes per (k,t)) finishes on CPU.  That estimator is gradient-free: parameter updates do not depend on it.  The oracle skips its
arithmetic and only consumes its draws (oracle.burn_pixel_epistemic_draws); this script asserts that every parameter of every
optimiser group and every non-epistemic metric of the oracle equals the reference, then writes tests/golden/pixel_step.npz.
"""
import functools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aid_oracle as orc  # noqa: E402
from oracle import gen_golden as gg  # noqa: E402
from oracle.encoder_weights import (conv_decoder_specs, draw_conv_decoder_params, draw_encoder_params, draw_pixel_estimator_params, encoder_specs,  # noqa: E402
                                    pixel_estimator_specs)
from oracle.noise import NoiseStream  # noqa: E402
from oracle.weights import draw_feature_decoder, draw_params, param_specs  # noqa: E402


def main():
    torch.set_num_threads(8)
    Box = gg.install_stubs()
    sys.path.insert(0, gg.REF)
    tape = gg.Tape(); tape.install()
    import active_inference_diffusion as aid
    from active_inference_diffusion.agents.pixel_agent import DiffusionPixelAgent

    L, A, D, N, H, B, K, S, CAP = 16, 4, 64, 3, 1, 3, 2, 2, 4
    KEYS = dict(weights=2001, encoder=2002, fdec=2003, store=2004, step=2005, decoder=2006, estimator=2007)

    class Env:
        observation_space = Box(0.0, 1.0, (3, 3, 84, 84)); action_space = Box(-1.0, 1.0, (A,))
    cfg = aid.ActiveInferenceConfig(device="cpu", batch_size=B, efe_horizon=H, latent_dim=L, hidden_dim=D, pixel_observation=True)
    cfg.diffusion = aid.DiffusionConfig(num_diffusion_steps=N)
    tape.use(999)
    agent = DiffusionPixelAgent(Env(), cfg, aid.TrainingConfig(buffer_size=CAP), aid.PixelObservationConfig())
    ai = agent.active_inference
    # ---- weights: everything is drawn from the shared stream (incl. the ConvDecoder and the pixel estimator, so that the GPU test
    #      can rebuild them from the keys)
    p = draw_params(param_specs(L, A, L, D, N), KEYS["weights"], orc.schedule_tables(N))
    p = {k: v for k, v in p.items() if not k.startswith(("observation_decoder.", "epistemic_estimator."))}
    p.update(draw_feature_decoder(L, D, KEYS["fdec"]))
    pdec = draw_conv_decoder_params(conv_decoder_specs(L, hidden=D), KEYS["decoder"])
    pest = draw_pixel_estimator_params(pixel_estimator_specs(L), KEYS["estimator"])
    p.update({"observation_decoder." + k: v for k, v in pdec.items()})
    p.update({"epistemic_estimator.decoder." + k: v for k, v in pdec.items()})
    p.update({"epistemic_estimator." + k: v for k, v in pest.items()})
    sd = ai.state_dict()
    assert set(p) == set(sd), sorted(set(p) ^ set(sd))[:8]
    for k, v in p.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    ai.load_state_dict(p, strict=True)
    pe = draw_encoder_params(encoder_specs(L), KEYS["encoder"])
    agent.encoder.load_state_dict(pe, strict=True)
    agent.score_ema = aid.core.EMAModel(ai.latent_score_network, decay=0.9999, device="cpu")
    full = {k: v.clone() for k, v in p.items()}
    full.update({"encoder." + k: v.clone() for k, v in pe.items()})
    # ---- replay store
    bs = NoiseStream(KEYS["store"])
    store = {"observations": bs.rand(CAP, 3, 3, 84, 84), "actions": np.tanh(bs.randn(CAP, A)).astype(np.float32), "rewards": bs.randn(CAP),
             "next_observations": bs.rand(CAP, 3, 3, 84, 84), "dones": np.zeros(CAP, bool)}
    for i in range(CAP):
        agent.replay_buffer.add(store["observations"][i], store["actions"][i], float(store["rewards"][i]), store["next_observations"][i], False)
    agent.total_steps = 1
    ai.compute_expected_free_energy_diffusion = functools.partial(ai.compute_expected_free_energy_diffusion, num_trajectories=K, num_ambiguity_samples=S)
    ai.train(); agent.encoder.train()
    tape.use(KEYS["step"])
    m_ref = agent.train_step()
    print("reference metrics:", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in m_ref.items()})
    # ---- oracle
    st = orc.OraclePixelState(full, N)
    ns = NoiseStream(KEYS["step"])
    batch = orc.replay_sample({k: v for k, v in store.items()}, CAP, B, ns)
    batch["observations"] = batch["observations"].reshape(B, 9, 84, 84); batch["next_observations"] = batch["next_observations"].reshape(B, 9, 84, 84)
    m = orc.pixel_train_step(st, batch, ns, horizon=H, num_trajectories=K, pixel_epistemic_samples=S)
    skip = ()
    m_ref = {k: (float(v) if torch.is_tensor(v) else v) for k, v in m_ref.items()}
    for k, v in m_ref.items():
        if k in skip or not isinstance(v, float):
            continue
        assert k in m, k
        gg.close(np.float32(m[k]), np.float32(v), 2e-4, 1e-6, f"pixel train_step metric {k}")
    ref_sd = {**ai.state_dict(), **{"encoder." + k: v for k, v in agent.encoder.state_dict().items()}}
    worst = 0.0
    for n in st.p:
        if n.endswith("epistemic_estimator.running_mean"):     # tracked in st.running_mean (checked below)
            continue
        worst = max(worst, gg.close(st.p[n].detach(), ref_sd[n], 1e-4, 2e-7, f"pixel train_step param {n}"))
    print(f"oracle == reference on {len(st.p)} tensors, worst abs err {worst:.2e}")
    moved = sum(1 for n in st.p if not torch.equal(st.p[n].detach(), full[n]))
    print("tensors changed by the step:", moved)
    g = {"dims": np.array([L, A, D, N, H, B, K, S, CAP]), "keys": np.array([KEYS[k] for k in ("weights", "encoder", "fdec", "store", "step", "decoder", "estimator")])}
    for k, v in m_ref.items():
        if isinstance(v, float) and k not in skip:
            g["metric/" + k] = np.float64(v)
    names = sorted(n for n in st.p if not n.startswith(("observation_decoder.", "epistemic_estimator.")))    # the trainable / stateful part
    assert abs(float(ref_sd["epistemic_estimator.running_mean"]) - st.running_mean) < 1e-5 * abs(st.running_mean)
    g["running_mean"] = np.float64(ref_sd["epistemic_estimator.running_mean"])
    g["names"] = np.array(names)
    g["checksum"] = np.array([float(ref_sd[n].double().sum()) for n in names]); g["abssum"] = np.array([float(ref_sd[n].double().abs().sum()) for n in names])
    for n in ("encoder.convs.0.weight_orig", "encoder.convs.3.weight_u", "encoder.norms.2.weight", "encoder.attention.temperature", "encoder.attention.spatial_conv.weight",
              "encoder.output_layers.4.weight", "feature_decoder.0.weight", "feature_decoder.4.bias", "latent_score_network.obs_encoder.0.weight",
              "latent_score_network.output_proj.2.weight", "latent_diffusion.latent_prior_mean", "policy_network.mean_head.2.weight",
              "value_network.network.9.weight", "latent_dynamics.network.0.weight"):
        g["param/" + n] = ref_sd[n].numpy().copy()
    g["param_sample/encoder.output_layers.0.weight"] = ref_sd["encoder.output_layers.0.weight"].reshape(-1)[::9973].numpy().copy()
    out = os.path.join(ROOT, "tests", "golden", "pixel_step.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
