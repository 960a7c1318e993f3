#!/usr/bin/env python3
"""The parameter lists the REFERENCE hands to its four torch.optim.AdamW optimisers (state and pixel agents), as dotted names in order
(container-only: imports the reference read-only, same stubs as gen_golden.py).  torch indexes optimiser state by position in that list,
so a checkpoint written by `utils/training.py:72-122` can only be exchanged if the build lays its optimiser state_dict out in exactly
this order.  Writes tests/golden/opt_layout.json (names and shapes only)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as gg  # noqa: E402


def layout(agent, opt):
    names = {}
    for prefix, mod in (("active_inference.", agent.active_inference), ("encoder.", getattr(agent, "encoder", None))):
        if mod is not None:
            for n, p in mod.named_parameters():
                names.setdefault(id(p), prefix + n)
    assert len(opt.param_groups) == 1
    return [[names[id(p)], list(p.shape)] for p in opt.param_groups[0]["params"]]


def main():
    Box = gg.install_stubs()
    sys.path.insert(0, gg.REF)
    import active_inference_diffusion as aid
    from active_inference_diffusion.agents.pixel_agent import DiffusionPixelAgent
    out = {}

    class SEnv:
        observation_space = Box(-np.inf, np.inf, (17,)); action_space = Box(-1.0, 1.0, (6,))
    cfg = aid.ActiveInferenceConfig(device="cpu", batch_size=8, efe_horizon=2, latent_dim=17, hidden_dim=64)
    cfg.diffusion = aid.DiffusionConfig(num_diffusion_steps=4)
    agent = aid.DiffusionStateAgent(SEnv(), cfg, aid.TrainingConfig(buffer_size=16))
    out["state"] = {"dims": {"O": 17, "A": 6, "L": 17, "D": 64, "N": 4},
                    **{k: layout(agent, getattr(agent, k)) for k in ("score_optimizer", "policy_optimizer", "value_optimizer", "dynamics_optimizer")}}
    hyper = {k: {h: agent.__dict__[k].param_groups[0][h] for h in ("lr", "weight_decay", "eps")} for k in ("score_optimizer", "value_optimizer")}
    out["state"]["hyper"] = hyper

    class PEnv:
        observation_space = Box(0.0, 1.0, (3, 3, 84, 84)); action_space = Box(-1.0, 1.0, (4,))
    cfg = aid.ActiveInferenceConfig(device="cpu", batch_size=3, efe_horizon=1, latent_dim=16, hidden_dim=64, pixel_observation=True)
    cfg.diffusion = aid.DiffusionConfig(num_diffusion_steps=3)
    agent = DiffusionPixelAgent(PEnv(), cfg, aid.TrainingConfig(buffer_size=4), aid.PixelObservationConfig())
    out["pixel"] = {"dims": {"A": 4, "L": 16, "D": 64, "N": 3},
                    **{k: layout(agent, getattr(agent, k)) for k in ("score_optimizer", "policy_optimizer", "value_optimizer", "dynamics_optimizer")}}
    path = os.path.join(ROOT, "tests", "golden", "opt_layout.json")
    json.dump(out, open(path, "w"), indent=0)
    for mode in out:
        print(mode, {k: len(v) for k, v in out[mode].items() if k.endswith("optimizer")})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
