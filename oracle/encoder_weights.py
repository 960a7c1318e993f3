"""DrQV2Encoder state_dict layout + deterministic weights for the pixel-path goldens.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import numpy as np
import torch

from .noise import NoiseStream


def encoder_specs(feature_dim: int, in_ch: int = 9, hw: int = 42):
    ch = (in_ch, 32, 64, 128, 256)
    s = []
    for i in range(4):
        s += [(f"convs.{i}.weight_orig", (ch[i + 1], ch[i], 3, 3)), (f"convs.{i}.weight_u", (ch[i + 1],)), (f"convs.{i}.weight_v", (ch[i] * 9,))]
    for i in range(4):
        s += [(f"norms.{i}.weight", (ch[i + 1],)), (f"norms.{i}.bias", (ch[i + 1],))]
    s += [("attention.temperature", (1,)), ("attention.channel_reduce.weight", (32, 256, 1, 1)), ("attention.channel_reduce.bias", (32,)),
          ("attention.spatial_conv.weight", (1, 2, 7, 7)), ("attention.spatial_conv.bias", (1,))]
    F = 256 * hw * hw
    s += [("ln.weight", (F,)), ("ln.bias", (F,)), ("output_layers.0.weight", (2 * feature_dim, F)), ("output_layers.0.bias", (2 * feature_dim,)),
          ("output_layers.1.weight", (2 * feature_dim,)), ("output_layers.1.bias", (2 * feature_dim,)),
          ("output_layers.4.weight", (feature_dim, 2 * feature_dim)), ("output_layers.4.bias", (feature_dim,)),
          ("output_layers.5.weight", (feature_dim,)), ("output_layers.5.bias", (feature_dim,))]
    return s


def draw_encoder_params(specs, key: int):
    ns = NoiseStream(key)
    out = {}
    for name, shape in specs:
        x = ns.randn(*shape)
        leaf = name.split(".")[-1]
        if leaf == "weight_orig" or name.endswith("spatial_conv.weight") or name.endswith("channel_reduce.weight"):
            v = x * (1.0 / np.sqrt(np.prod(shape[1:])))
        elif leaf in ("weight_u", "weight_v"):
            v = x / np.linalg.norm(x)
        elif leaf == "temperature":
            v = 1.0 + 0.1 * x
        elif len(shape) == 2:
            v = x * (1.0 / np.sqrt(shape[1]))
        elif leaf == "bias":
            v = 0.05 * x
        else:   # LayerNorm / GroupNorm gains
            v = 1.0 + 0.1 * x
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)
    return out
