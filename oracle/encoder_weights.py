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


def conv_decoder_specs(latent_dim: int, hidden: int = 256, img_channels: int = 3, spatial: int = 21):
    """ConvDecoder.state_dict() layout (visual_encoders.py:243-330), in the reference's order."""
    H = hidden
    s = [("latent_proj.0.weight", (4 * H, latent_dim)), ("latent_proj.0.bias", (4 * H,)), ("latent_proj.1.weight", (4 * H,)), ("latent_proj.1.bias", (4 * H,)),
         ("latent_proj.4.weight", (H * spatial * spatial, 4 * H)), ("latent_proj.4.bias", (H * spatial * spatial,)),
         ("latent_proj.5.weight", (H * spatial * spatial,)), ("latent_proj.5.bias", (H * spatial * spatial,))]

    def sn(name, cout, cin, k):
        return [(name + ".bias", (cout,)), (name + ".weight_orig", (cout, cin, k, k)), (name + ".weight_u", (cout,)), (name + ".weight_v", (cin * k * k,))]
    for i, (cin, cout, up) in enumerate(((H, H, False), (H, H // 2, True), (H // 2, H // 4, True), (H // 4, H // 8, True))):
        b = f"decoder_blocks.{i}."
        s += sn(b + "main_path.0", cout * 4 if up else cout, cin, 3)
        s += sn(b + ("main_path.5" if up else "main_path.4"), cout, cout, 3)
        if up:
            s += sn(b + "residual_path.0", cout * 4, cin, 1)
    for i, (cin, cout) in zip((0, 3, 6), ((H // 8, 32), (32, 32), (32, img_channels))):
        s += [(f"output_proj.{i}.weight", (cout, cin, 3, 3)), (f"output_proj.{i}.bias", (cout,))]
    return s


def draw_conv_decoder_params(specs, key: int):
    """Deterministic O(1)-activation weights for the ConvDecoder (same rules as draw_encoder_params); the last conv is scaled down so
    that the sigmoid output stays away from saturation (a saturated image would hide errors from a parity check)."""
    p = draw_encoder_params(specs, key)
    for name, shape in specs:          # plain (not spectral-normed) convs of output_proj: fan-in scaling instead of the norm-gain rule
        if len(shape) == 4 and name.endswith(".weight"):
            p[name] = (p[name] - 1.0) / 0.1 / float(np.sqrt(np.prod(shape[1:])))
    p["output_proj.6.weight"] = p["output_proj.6.weight"] * 0.5
    return p


def pixel_estimator_specs(latent_dim: int, img_channels: int = 3):
    """Pixel-branch FunctionSpaceEpistemicEstimator.state_dict() minus its `decoder.*` alias (active_inference.py:865-925)."""
    e = []
    e += [("perturbation_scale", ()), ("running_mean", ())]
    for i, (cin, cout) in zip((0, 2, 4), ((img_channels, 32), (32, 64), (64, 128))):
        e += [(f"pixel_processor.{i}.weight", (cout, cin, 5, 5)), (f"pixel_processor.{i}.bias", (cout,))]
    a = "spatial_aggregator."
    e += [(a + "pos_encoding", (1, 441, 128)), (a + "epistemic_queries", (1, 16, 128)), (a + "attention.in_proj_weight", (384, 128)),
          (a + "attention.in_proj_bias", (384,)), (a + "attention.out_proj.weight", (128, 128)), (a + "attention.out_proj.bias", (128,)),
          (a + "output_proj.0.weight", (512, 2048)), (a + "output_proj.0.bias", (512,)), (a + "output_proj.1.weight", (512,)), (a + "output_proj.1.bias", (512,)),
          (a + "output_proj.3.weight", (256, 512)), (a + "output_proj.3.bias", (256,))]
    e += [("jacobian_projector.0.weight", (512, 1024)), ("jacobian_projector.0.bias", (512,)), ("jacobian_projector.1.weight", (512,)),
          ("jacobian_projector.1.bias", (512,)), ("jacobian_projector.4.weight", (256, 512)), ("jacobian_projector.4.bias", (256,)),
          ("latent_processor.0.weight", (128, latent_dim)), ("latent_processor.0.bias", (128,)), ("latent_processor.2.weight", (128, 128)),
          ("latent_processor.2.bias", (128,)), ("mine_network.0.weight", (512, 384)), ("mine_network.0.bias", (512,)),
          ("mine_network.3.weight", (512, 512)), ("mine_network.3.bias", (512,)), ("mine_network.6.weight", (1, 512)), ("mine_network.6.bias", (1,))]
    return e


def draw_pixel_estimator_params(specs, key: int):
    ns = NoiseStream(key)
    out = {}
    for name, shape in specs:
        if name == "perturbation_scale":
            v = np.float32(0.1)
        elif name == "running_mean":
            v = np.float32(0.0)
        else:
            x = ns.randn(*shape)
            leaf = name.split(".")[-1]
            if len(shape) == 4:
                v = x / np.sqrt(np.prod(shape[1:]))
            elif name.endswith(("pos_encoding", "epistemic_queries")):
                v = 0.5 * x
            elif len(shape) == 2 or leaf == "in_proj_weight":
                v = x / np.sqrt(shape[-1])
            elif leaf in ("bias", "in_proj_bias"):
                v = 0.05 * x
            else:
                v = 1.0 + 0.1 * x
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)
    return out
