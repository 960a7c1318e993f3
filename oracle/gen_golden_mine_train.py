#!/usr/bin/env python3
"""Golden vectors for DiffusionActiveInference.train_epistemic_estimator, from a run of the REFERENCE (container-only; same stubs /
RNG tape as gen_golden.py).  The concrete agents never create `epistemic_optimizer`; the harness attaches the one the base
class defines (active_inference_diffusion/agents/base_agent.py:134-139: Adam, lr*0.1, weight_decay 1e-5) - the reference files are
not edited.  Asserts oracle == reference for the loss, every estimator gradient and every parameter after two optimiser steps, then
writes tests/golden/mine_train.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aid_oracle as orc  # noqa: E402
from oracle import gen_golden as gg  # noqa: E402
from oracle.noise import NoiseStream  # noqa: E402
from oracle.weights import draw_params, param_specs  # noqa: E402


def main():
    torch.set_num_threads(4)
    Box = gg.install_stubs()
    sys.path.insert(0, gg.REF)
    tape = gg.Tape(); tape.install()
    import torch.nn as nn
    import active_inference_diffusion as aid

    O, A, D, N, B = 17, 6, 64, 4, 8
    KEYS = dict(weights=3021, data=3002, step1=3005, step2=3004, step3=3003)   # steps 1-2: mi > 0 (gradient live), step 3: mi < 0 (clamp gates it)

    class Env:
        observation_space = Box(-np.inf, np.inf, (O,)); action_space = Box(-1.0, 1.0, (A,))
    cfg = aid.ActiveInferenceConfig(device="cpu", batch_size=B, efe_horizon=2, latent_dim=O, hidden_dim=D)
    cfg.diffusion = aid.DiffusionConfig(num_diffusion_steps=N)
    tape.use(999)
    agent = aid.DiffusionStateAgent(Env(), cfg, aid.TrainingConfig(buffer_size=16))
    ai = agent.active_inference

    class CallableDecoder(nn.ModuleList):   # SURVEY Appendix C.4 (defect D2): the state decoder is a ModuleList the estimator calls
        def forward(self, h):
            h1 = self[0](h); h2 = self[1](h1); h2 = h2 + h1
            return self[3](self[2](h2))
    ai.observation_decoder.__class__ = CallableDecoder
    specs = param_specs(O, A, O, D, N)
    p = draw_params(specs, KEYS["weights"], orc.schedule_tables(N))
    sd = ai.state_dict()
    ai.load_state_dict({**p, **{k: p[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd if k.startswith("epistemic_estimator.decoder.")}}, strict=True)
    ai.epistemic_optimizer = torch.optim.Adam(ai.epistemic_estimator.parameters(), lr=cfg.learning_rate * 0.1, weight_decay=1e-5)
    ds = NoiseStream(KEYS["data"])
    lat = ds.randn(B, O); act = np.tanh(ds.randn(B, A)).astype(np.float32); nlat = ds.randn(B, O)
    st = orc.OracleAgentState(p, N)
    g = {"dims": np.array([O, A, D, N, B]), "keys": np.array([KEYS[k] for k in ("weights", "data", "step1", "step2", "step3")])}
    ai.train()
    for step, key in ((1, KEYS["step1"]), (2, KEYS["step2"]), (3, KEYS["step3"])):
        tape.use(key)
        mi_ref, met_ref = ai.train_epistemic_estimator(torch.from_numpy(lat), torch.from_numpy(act), torch.from_numpy(nlat))
        mi, met, grads = orc.train_epistemic_estimator(st, torch.from_numpy(lat), torch.from_numpy(act), NoiseStream(key), lr=cfg.learning_rate, clip=cfg.gradient_clip)
        gg.close(np.float32(mi), np.float32(mi_ref), 1e-5, 1e-6, f"mine train step {step} mi")
        for k in met_ref:
            gg.close(np.float32(met[k]), np.float32(met_ref[k]), 1e-5, 1e-6, f"mine train step {step} {k}")
        ref_sd = ai.state_dict()
        worst = 0.0
        for n in grads:
            worst = max(worst, gg.close(st.p[n].detach(), ref_sd[n], 1e-5, 1e-8, f"mine train step {step} param {n}"))
        print(f"step {step}: mi {mi_ref:.6f}, oracle == reference on {len(grads)} estimator tensors (worst abs err {worst:.2e})")
        g[f"s{step}_mi"] = np.float64(mi_ref)
        for k, v in met_ref.items():
            g[f"s{step}_metric/{k}"] = np.float64(v)
        names = sorted(grads)
        g["names"] = np.array(names)
        g[f"s{step}_checksum"] = np.array([float(ref_sd[n].double().sum()) for n in names]); g[f"s{step}_abssum"] = np.array([float(ref_sd[n].double().abs().sum()) for n in names])
        for n in names:   # small tensors whole, large ones as a strided sample
            t = ref_sd[n].reshape(-1)
            g[f"s{step}_param/{n}"] = (t if t.numel() <= 4096 else t[::97]).numpy().copy()
        if step == 1:   # the reference's clipped gradients are not observable after the step; store the oracle's (== reference by the above)
            for n, gr in grads.items():
                t = gr.reshape(-1)
                g[f"s1_grad/{n}"] = (t if t.numel() <= 4096 else t[::97]).numpy().copy()
                g[f"s1_gradsum/{n}"] = np.array([float(t.double().sum()), float(t.double().abs().sum())])
    # nothing else moved
    for n in ("observation_decoder.3.weight", "latent_dynamics.network.0.weight", "policy_network.mean_head.2.weight"):
        assert torch.equal(ai.state_dict()[n], p[n]), n
    out = os.path.join(ROOT, "tests", "golden", "mine_train.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
