"""State-mode `DiffusionActiveInference.state_dict()` layout and a
deterministic weight generator driven by `oracle.noise.NoiseStream`.

TEST INFRASTRUCTURE ONLY.  `param_specs` restates the reference constructors
(aid/core/active_inference.py:59-171, aid/models/*.py) as an ordered (name, shape) list; the
golden generator asserts it equals the reference's own `state_dict()` (names, order, shapes).
`draw_params` fills every tensor from the stream with O(1)-activation scales, *including* the
layers the reference zero-initialises (adaLN modulation, last score layer), because zero weights
would hide most of the arithmetic from a parity check.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .noise import NoiseStream


def param_specs(O: int, A: int, L: int, D: int, N: int, n_blocks: int = 6, te: int = 128
                ) -> List[Tuple[str, Tuple[int, ...]]]:
    s: List[Tuple[str, Tuple[int, ...]]] = []
    add = lambda n, *shape: s.append((n, tuple(shape)))
    add("reward_mean"); add("reward_var"); add("preference_temperature")
    d = "latent_diffusion."
    add(d + "latent_prior_mean", L); add(d + "latent_prior_log_std", L)
    add(d + "log_snr_min"); add(d + "log_snr_max")
    for n in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped"):
        add(d + n, N)
    add(d + "loss_weight_cache", 1000)
    k = "latent_score_network."
    add(k + "time_scale"); add(k + "output_multiplier", 1); add(k + "grad_norm_ema")
    add(k + "time_embed.0.freq_scale", 1)
    add(k + "time_embed.1.weight", 2 * D, te); add(k + "time_embed.1.bias", 2 * D)
    add(k + "time_embed.3.weight", D, 2 * D); add(k + "time_embed.3.bias", D)
    # NB: the reference sizes the obs encoder with latent_dim (defect D1) -> O == L required.
    for i, (fo, fi) in zip((0, 4, 7), ((D, L), (D, D), (D, D))):
        add(k + f"obs_encoder.{i}.weight", fo, fi); add(k + f"obs_encoder.{i}.bias", fo)
        add(k + f"obs_encoder.{i + 1}.weight", fo); add(k + f"obs_encoder.{i + 1}.bias", fo)
    for i, (fo, fi) in zip((0, 2, 4), ((te, 1), (te, te), (D, te))):
        add(k + f"continuous_time_embed.{i}.weight", fo, fi); add(k + f"continuous_time_embed.{i}.bias", fo)
    add(k + "latent_proj.weight", D, L); add(k + "latent_proj.bias", D)
    for b in range(n_blocks):
        t = k + f"transformer_blocks.{b}."
        add(t + "norm1.adaLN_modulation.1.weight", 2 * D, D); add(t + "norm1.adaLN_modulation.1.bias", 2 * D)
        add(t + "norm2.adaLN_modulation.1.weight", 2 * D, D); add(t + "norm2.adaLN_modulation.1.bias", 2 * D)
        add(t + "attention.in_proj_weight", 3 * D, D); add(t + "attention.in_proj_bias", 3 * D)
        add(t + "attention.out_proj.weight", D, D); add(t + "attention.out_proj.bias", D)
        add(t + "mlp.0.weight", 4 * D, D); add(t + "mlp.0.bias", 4 * D)
        add(t + "mlp.2.weight", D, 4 * D); add(t + "mlp.2.bias", D)
    add(k + "norm_final.adaLN_modulation.1.weight", 2 * D, D); add(k + "norm_final.adaLN_modulation.1.bias", 2 * D)
    add(k + "output_proj.0.weight", D // 2, D); add(k + "output_proj.0.bias", D // 2)
    add(k + "output_proj.2.weight", L, D // 2)
    q = "policy_network."
    add(q + "latent_encoder.0.weight", D, L); add(q + "latent_encoder.0.bias", D)
    add(q + "latent_encoder.1.weight", D); add(q + "latent_encoder.1.bias", D)
    add(q + "latent_encoder.3.weight", D, D); add(q + "latent_encoder.3.bias", D)
    for i in (0, 3, 6):
        add(q + f"trunk.{i}.weight", D, D); add(q + f"trunk.{i}.bias", D)
        add(q + f"trunk.{i + 1}.weight", D); add(q + f"trunk.{i + 1}.bias", D)
    for h in ("mean_head", "log_std_head"):
        add(q + f"{h}.0.weight", D // 2, D); add(q + f"{h}.0.bias", D // 2)
        add(q + f"{h}.2.weight", A, D // 2); add(q + f"{h}.2.bias", A)
    v = "value_network."
    add(v + "time_embed.0.freq_scale", 1)
    add(v + "time_embed.1.weight", te, te); add(v + "time_embed.1.bias", te)
    for i, fi in zip((0, 3, 6), (L + te, D, D)):
        add(v + f"network.{i}.weight", D, fi); add(v + f"network.{i}.bias", D)
        add(v + f"network.{i + 1}.weight", D); add(v + f"network.{i + 1}.bias", D)
    add(v + "network.9.weight", 1, D); add(v + "network.9.bias", 1)
    y = "latent_dynamics."
    for i, fi in zip((0, 3, 6), (L + A, D, D)):
        add(y + f"network.{i}.weight", D, fi); add(y + f"network.{i}.bias", D)
        add(y + f"network.{i + 1}.weight", D); add(y + f"network.{i + 1}.bias", D)
    add(y + "network.9.weight", L, D); add(y + "network.9.bias", L)
    o = "observation_decoder."
    for i, (fo, fi) in enumerate(((2 * D, L), (2 * D, 2 * D), (D, 2 * D))):
        add(o + f"{i}.0.weight", fo, fi); add(o + f"{i}.0.bias", fo)
        add(o + f"{i}.1.weight", fo); add(o + f"{i}.1.bias", fo)
    add(o + "3.weight", O, D); add(o + "3.bias", O)
    e = "epistemic_estimator."
    add(e + "perturbation_scale"); add(e + "running_mean")
    for i, (fo, fi) in zip((0, 2, 4), ((128, O), (256, 128), (128, 256))):
        add(e + f"feature_extractor.{i}.weight", fo, fi); add(e + f"feature_extractor.{i}.bias", fo)
    add(e + "jacobian_projector.0.weight", 512, 512); add(e + "jacobian_projector.0.bias", 512)
    add(e + "jacobian_projector.1.weight", 512); add(e + "jacobian_projector.1.bias", 512)
    add(e + "jacobian_projector.4.weight", 256, 512); add(e + "jacobian_projector.4.bias", 256)
    add(e + "latent_processor.0.weight", 128, L); add(e + "latent_processor.0.bias", 128)
    add(e + "latent_processor.2.weight", 128, 128); add(e + "latent_processor.2.bias", 128)
    add(e + "mine_network.0.weight", 512, 384); add(e + "mine_network.0.bias", 512)
    add(e + "mine_network.3.weight", 512, 512); add(e + "mine_network.3.bias", 512)
    add(e + "mine_network.6.weight", 1, 512); add(e + "mine_network.6.bias", 1)
    r = "reward_predictor."
    add(r + "0.weight", D, L); add(r + "0.bias", D); add(r + "1.weight", D); add(r + "1.bias", D)
    add(r + "3.weight", D // 2, D); add(r + "3.bias", D // 2)
    add(r + "5.weight", 2, D // 2); add(r + "5.bias", 2)
    return s


_FIXED = {
    "reward_mean": 0.0, "reward_var": 1.0, "preference_temperature": 1.0,
    "latent_diffusion.log_snr_min": -10.0, "latent_diffusion.log_snr_max": 10.0,
    "latent_score_network.time_scale": 1.0, "latent_score_network.grad_norm_ema": 1.0,
    "epistemic_estimator.perturbation_scale": 0.1, "epistemic_estimator.running_mean": 0.0,
}


def draw_params(specs, key: int, schedule_tables=None, head_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Draw every tensor of `specs` (in order) from NoiseStream(key).

    2-D weights ~ N(0, 1/fan_in) (x0.3 for adaLN modulation so that 1+scale stays positive-ish),
    biases ~ 0.05 N(0,1), LayerNorm weights 1 + 0.1 N(0,1), freq_scale 1 + 0.01 N, the score
    head / output multiplier non-zero.  Schedule buffers are filled from `schedule_tables`
    (a dict from oracle.schedule_tables) when given, else drawn positive.
    """
    ns = NoiseStream(key)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in specs:
        if name in _FIXED:
            out[name] = torch.tensor(_FIXED[name], dtype=torch.float32).reshape(shape)
            continue
        leaf = name.split(".")[-1]
        base = name.split(".")[-2] if "." in name else ""
        if name.startswith("latent_diffusion.") and leaf in (
                "betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped"):
            assert schedule_tables is not None
            out[name] = schedule_tables[leaf].clone().float()
            continue
        if leaf == "loss_weight_cache":
            out[name] = torch.zeros(shape)
            continue
        x = ns.randn(*shape) if len(shape) else ns.randn(1).reshape(())
        if leaf == "output_multiplier":
            v = np.full(shape, 1e-3 * head_scale, dtype=np.float32)
        elif leaf == "freq_scale":
            v = 1.0 + 0.01 * x
        elif leaf == "latent_prior_mean":
            v = 0.1 * x
        elif leaf == "latent_prior_log_std":
            v = 0.1 * x
        elif len(shape) == 2 or leaf in ("in_proj_weight",):
            fan_in = shape[1]
            sc = 1.0 / np.sqrt(fan_in)
            if "adaLN_modulation" in name:
                sc *= 0.3
            v = x * sc
        elif leaf in ("bias", "in_proj_bias"):
            # LayerNorm biases and Linear biases alike
            v = 0.05 * x
        elif leaf == "weight" and len(shape) == 1:
            v = 1.0 + 0.1 * x  # LayerNorm gain
        else:
            v = 0.1 * x
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)
    return out


def draw_feature_decoder(L: int, D: int, key: int) -> Dict[str, torch.Tensor]:
    """Pixel-mode `feature_decoder` (active_inference.py:142-149): Linear(L,D), LayerNorm(D), ReLU, Dropout, Linear(D,L), Tanh."""
    ns = NoiseStream(key)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return {"feature_decoder.0.weight": t(ns.randn(D, L) / np.sqrt(L)), "feature_decoder.0.bias": t(0.05 * ns.randn(D)),
            "feature_decoder.1.weight": t(1.0 + 0.1 * ns.randn(D)), "feature_decoder.1.bias": t(0.05 * ns.randn(D)),
            "feature_decoder.4.weight": t(ns.randn(L, D) / np.sqrt(D)), "feature_decoder.4.bias": t(0.05 * ns.randn(L))}
