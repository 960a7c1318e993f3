#!/usr/bin/env python3
"""Golden vectors for `FunctionSpaceEpistemicEstimator.compute_jacobian_features` (core/active_inference.py:940-993, state branch) as a call of its own, produced
by running the REFERENCE's estimator (imported read-only from /root/reference; container-only, same stubs / RNG tape / decoder shim as gen_golden.py) and compared
with `oracle.aid_oracle.jacobian_features` while they are written (a mismatch aborts).  Eval and train mode (the projector's dropout), HalfCheetah dims.

Usage: python oracle/gen_golden_jac.py  -> tests/golden/jacobian_features.npz
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
from oracle.weights import draw_params, param_specs  # noqa: E402

O, A, D, N, R, KEY = 17, 6, 64, 4, 48, 7301


def main():
    torch.set_num_threads(1)
    Box = gg.install_stubs()
    sys.path.insert(0, gg.REF)
    tape = gg.Tape(); tape.install()
    import torch.nn as nn
    import active_inference_diffusion as aid

    class FakeEnv:
        def __init__(self):
            self.observation_space = Box(-np.inf, np.inf, (O,)); self.action_space = Box(-1, 1, (A,))

    cfg = aid.ActiveInferenceConfig(device="cpu", batch_size=8, efe_horizon=1, latent_dim=O, hidden_dim=D)
    cfg.diffusion = aid.DiffusionConfig(num_diffusion_steps=N)
    tape.use(999)
    ai = aid.DiffusionStateAgent(FakeEnv(), cfg, aid.TrainingConfig(buffer_size=64)).active_inference

    class CallableDecoder(nn.ModuleList):      # SURVEY Appendix C.4 (defect D2): the ModuleList decoder made callable the way its layers chain
        def forward(self, h):
            h1 = self[0](h); h2 = self[1](h1); h2 = h2 + h1
            return self[3](self[2](h2))
    ai.observation_decoder.__class__ = CallableDecoder
    sd = ai.state_dict()
    params = draw_params(param_specs(O, A, O, D, N), KEY, orc.schedule_tables(N))
    ai.load_state_dict({**params, **{k: params[k.replace("epistemic_estimator.decoder.", "observation_decoder.")] for k in sd if k.startswith("epistemic_estimator.decoder.")}},
                       strict=True)
    est = ai.epistemic_estimator
    z = torch.from_numpy(NoiseStream(KEY + 1).randn(R, O))
    g = {"O": O, "A": A, "D": D, "N": N, "R": R, "weights_key": KEY, "z_key": KEY + 1}
    for mode, key in (("eval", KEY + 2), ("train", KEY + 3)):
        est.train(mode == "train")
        tape.use(key)
        with torch.no_grad():
            ref = est.compute_jacobian_features(z)
        mine = orc.jacobian_features(params, z, NoiseStream(key), mode == "train")
        assert ref.shape == (R, 256)
        assert torch.allclose(mine, ref, rtol=1e-5, atol=1e-6), float((mine - ref).abs().max())
        g[f"jac_{mode}"] = ref.numpy(); g[f"noise_key_{mode}"] = key
        print(mode, "max |oracle - reference| =", float((mine - ref).abs().max()), " |ref| max", float(ref.abs().max()))
    out = os.path.join(ROOT, "tests", "golden", "jacobian_features.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
