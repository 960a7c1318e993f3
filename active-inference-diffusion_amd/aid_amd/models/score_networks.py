"""LatentScoreNetwork s_theta(z_t, t, o) — module tree with the reference's state_dict layout
(active_inference_diffusion/models/score_networks.py:12-291); forward on HIP kernels.

Differences from the reference implementation, none observable in the outputs:
  * single-token MultiheadAttention == out_proj(V_proj(x)); Q/K projections are never evaluated (their
    parameters exist, receive exactly-zero gradients and still weight-decay, as in the reference);
  * the time embedding the reference computes and discards (:119) is skipped;
  * the whole-batch `0 <= t <= 1` branch test (:121, a device->host sync) can be bypassed with `continuous=`.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import _native as nat
from ._holders import TableCache, native_only, params_dict


def _need_gpu(name: str, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"{name}.forward runs on the HIP library (module and inputs on the GPU); there is no CPU path")


class SinusoidalPositionEmbeddings(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.freq_scale = nn.Parameter(torch.ones(1))

    def forward(self, time: torch.Tensor) -> torch.Tensor:
        """[sin(t f_i) | cos(t f_i)], f_i = freq_scale 10000^(-i / (dim/2 - 1))  (score_networks.py:282-291); one HIP launch."""
        _need_gpu("SinusoidalPositionEmbeddings", self.freq_scale)
        return nat.sinusoid_embed(time.to(self.freq_scale.device).float().reshape(-1).contiguous(), self.freq_scale.data, self.dim)


class AdaptiveLayerNorm(nn.Module):
    """LN(x) * (1 + scale) + shift with [scale, shift] = Linear(SiLU(cond)); zero-initialised (identity)."""

    def __init__(self, hidden_dim: int):
        super().__init__()
        self.norm = nn.LayerNorm(hidden_dim, elementwise_affine=False)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_dim, 2 * hidden_dim))
        nn.init.zeros_(self.adaLN_modulation[1].weight)
        nn.init.zeros_(self.adaLN_modulation[1].bias)

    def forward(self, x: torch.Tensor, conditioning: torch.Tensor) -> torch.Tensor:
        """score_networks.py:257-270 as a stand-alone call (inference; inside the network the 13 modulations are one hoisted GEMM)."""
        lin = self.adaLN_modulation[1]
        _need_gpu("AdaptiveLayerNorm", lin.weight, x, conditioning)
        return nat.adaptive_layer_norm(x.contiguous().float(), conditioning.contiguous().float(), lin.weight.data, lin.bias.data)


class _SingleTokenAttention(nn.Module):
    """Holds nn.MultiheadAttention's parameters (in_proj_weight/in_proj_bias/out_proj) with its initialiser."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.embed_dim, self.num_heads = dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = nn.Linear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    forward = native_only("attention")


class DiTBlock(nn.Module):
    def __init__(self, hidden_dim: int, num_heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.norm1, self.norm2 = AdaptiveLayerNorm(hidden_dim), AdaptiveLayerNorm(hidden_dim)
        self.attention = _SingleTokenAttention(hidden_dim, num_heads)
        inner = int(hidden_dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(hidden_dim, inner), nn.GELU(), nn.Linear(inner, hidden_dim))
        for lin in (self.mlp[0], self.mlp[2]):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, x: torch.Tensor, conditioning: torch.Tensor) -> torch.Tensor:
        """score_networks.py:214-235 as a stand-alone call (inference): x + attn(adaLN1(x)) then + MLP(adaLN2(.)); single-token attention = out_proj(V_proj(.))."""
        _need_gpu("DiTBlock", self.mlp[0].weight, x, conditioning)
        t = nat.AidDiTBlock()
        for key, field in nat.BLOCK_KEYMAP.items():
            mod = self
            for part in key.split("."):
                mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
            setattr(t, field, nat._ptr(mod.data).value)
        return nat.dit_block_forward(t, x.contiguous().float(), conditioning.contiguous().float())


class LatentScoreNetwork(nn.Module):
    def __init__(self, latent_dim: int, observation_dim: int, hidden_dim: int = 256, time_embed_dim: int = 128, num_layers: int = 6,
                 use_attention: bool = True, output_scale: float = 1e-3):
        super().__init__()
        if hidden_dim % 8:
            raise ValueError("hidden_dim must be divisible by the 8 attention heads")
        self.latent_dim, self.observation_dim, self.hidden_dim = latent_dim, observation_dim, hidden_dim
        self.time_embed_dim, self.num_heads, self.mlp_ratio = time_embed_dim, 8, 4.0
        self.use_attention, self.output_scale, self.grad_norm_decay = use_attention, output_scale, 0.999
        D, TE = hidden_dim, time_embed_dim
        self.time_scale = nn.Parameter(torch.tensor(1.0))
        self.output_multiplier = nn.Parameter(torch.ones(1) * output_scale)
        self.register_buffer("grad_norm_ema", torch.tensor(1.0))
        self.time_embed = nn.Sequential(SinusoidalPositionEmbeddings(TE), nn.Linear(TE, 2 * D), nn.SiLU(), nn.Linear(2 * D, D))
        self.obs_encoder = nn.Sequential(nn.Linear(observation_dim, D), nn.LayerNorm(D), nn.SiLU(), nn.Dropout(0.1), nn.Linear(D, D),
                                         nn.LayerNorm(D), nn.SiLU(), nn.Linear(D, D), nn.LayerNorm(D))
        self.continuous_time_embed = nn.Sequential(nn.Linear(1, TE), nn.SiLU(), nn.Linear(TE, TE), nn.SiLU(), nn.Linear(TE, D))
        self.latent_proj = nn.Linear(latent_dim, D)
        self.transformer_blocks = nn.ModuleList([DiTBlock(D, self.num_heads, self.mlp_ratio) for _ in range(num_layers if use_attention else 0)])
        self.norm_final = AdaptiveLayerNorm(D)
        self.output_proj = nn.Sequential(nn.Linear(D, D // 2), nn.SiLU(), nn.Linear(D // 2, latent_dim, bias=False))
        nn.init.zeros_(self.output_proj[-1].weight)
        self._tab = TableCache(lambda: nat.score_net_table(params_dict(self), "", latent_dim, observation_dim, D, len(self.transformer_blocks), TE))
        self._gtab = TableCache(lambda: nat.score_net_table(params_dict(self, True), "", latent_dim, observation_dim, D, len(self.transformer_blocks), TE))
        self._ws = None
        self.rng_seed, self.rng_offset = 0x5C0E, 0

    # adaLN weights first, back to back, so the 13 modulations run as one GEMM (see arena.py)
    def arena_order(self):
        named = dict(self.named_parameters())
        ada = [f"transformer_blocks.{i}.norm{j}.adaLN_modulation.1" for i in range(len(self.transformer_blocks)) for j in (1, 2)]
        ada.append("norm_final.adaLN_modulation.1")
        first = [a + ".weight" for a in ada] + [a + ".bias" for a in ada]
        rest = [n for n in named if n not in set(first)]
        return [(n, named[n]) for n in first + rest]

    def table(self):
        return self._tab.get(self)

    def grad_table(self):
        return self._gtab.get(self)

    def workspace(self):
        if self._ws is None or self._ws.device != self.latent_proj.weight.device:
            self._ws = nat.Workspace(self.latent_proj.weight.device)
        return self._ws

    def forward(self, z_t: torch.Tensor, time: torch.Tensor, observation: Optional[torch.Tensor] = None, continuous: Optional[bool] = None,
                drop_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        # observation=None: the reference's zero observation embedding (score_networks.py:144-149) - the native forward takes a NULL obs pointer
        z_t, time = z_t.contiguous().float(), time.contiguous().float()
        observation = None if observation is None else observation.contiguous().float()
        if continuous is None:  # reference semantics (score_networks.py:121): whole-batch test, costs one host sync
            continuous = bool(time.max() <= 1.0 and time.min() >= 0.0)
        if self.training and drop_mask is None:
            drop_mask = torch.empty(z_t.shape[0], self.hidden_dim, device=z_t.device)
            nat.fill_dropout_mask(drop_mask, 0.1, self.rng_seed, self.rng_offset)
            self.rng_offset += (drop_mask.numel() + 3) // 4
        return nat.score_forward(self.table(), z_t, time, observation, continuous, drop_mask if self.training else None, self.workspace())

    def backward(self, z_t: torch.Tensor, time: torch.Tensor, observation: torch.Tensor, d_score: torch.Tensor, continuous: Optional[bool] = None,
                 drop_mask: Optional[torch.Tensor] = None, d_observation: Optional[torch.Tensor] = None):
        """The network as a differentiable op (there is no autograd graph on this path): forward + reverse pass for the cotangent `d_score` = dLoss/dscore.
        ACCUMULATES the parameter gradients into `.grad` of this module's Parameters (zero them first for a fresh gradient, as with autograd) and, when given,
        dLoss/dobservation into `d_observation`.  -> (score, dLoss/dz_t).  Train mode draws / takes the obs encoder's dropout mask like forward()."""
        z_t, time, observation, d_score = (x.contiguous().float() for x in (z_t, time, observation, d_score))
        if continuous is None:
            continuous = bool(time.max() <= 1.0 and time.min() >= 0.0)
        if self.training and drop_mask is None:
            drop_mask = torch.empty(z_t.shape[0], self.hidden_dim, device=z_t.device)
            nat.fill_dropout_mask(drop_mask, 0.1, self.rng_seed, self.rng_offset)
            self.rng_offset += (drop_mask.numel() + 3) // 4
        return nat.score_backward(self.table(), self.grad_table(), z_t, time, observation, d_score, continuous, drop_mask if self.training else None,
                                  d_observation, self.workspace())
