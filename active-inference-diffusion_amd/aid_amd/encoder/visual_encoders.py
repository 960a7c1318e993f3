"""DrQV2Encoder / SpatialAttention / RandomShiftAugmentation (active_inference_diffusion/encoder/visual_encoders.py:13-225,
471-492): module tree with the reference's state_dict layout (legacy spectral-norm keys
`convs.i.weight_orig / weight_u / weight_v`), forward and backward on HIP kernels (csrc/conv.hip).
ConvDecoder (pixel reconstructions for the epistemic term) is SURVEY §8(f) rank 1 and not built yet."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as nat
from ..models._holders import TableCache, native_only


class _SpectralNormConv(nn.Module):
    """Holder for a bias-free 3x3 conv under legacy torch.nn.utils.spectral_norm: Parameter weight_orig + buffers weight_u/_v."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        conv = nn.Conv2d(cin, cout, 3, bias=False)           # default (kaiming-uniform) init == the reference's weight_orig
        self.weight_orig = nn.Parameter(conv.weight.detach().clone())
        self.register_buffer("weight_u", F.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(torch.randn(cin * 9), dim=0, eps=1e-12))

    forward = native_only("spectral-norm conv")


class SpatialAttention(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.temperature = nn.Parameter(torch.ones(1))
        self.channel_reduce = nn.Conv2d(channels, max(channels // 8, 16), 1)     # dead weight in the reference too (never used)
        self.spatial_conv = nn.Conv2d(2, 1, kernel_size=7, padding=3)
        for m in (self.channel_reduce, self.spatial_conv):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x + x * sigmoid(conv7x7([mean_c x, max_c x]) / temperature) on an NCHW tensor (visual_encoders.py:210-225) as a stand-alone call; inside the encoder
        the same arithmetic runs on the CNHW activation with its last step folded into LayerNorm(F)."""
        if not (x.is_cuda and self.temperature.is_cuda):
            raise RuntimeError("SpatialAttention.forward runs on the HIP library (module and input on the GPU); there is no CPU path")
        return nat.spatial_attention_forward(x.contiguous().float(), self.spatial_conv.weight.data.reshape(-1).contiguous(), self.spatial_conv.bias.data,
                                             self.temperature.data)


class DrQV2Encoder(nn.Module):
    def __init__(self, obs_shape: Tuple[int, int, int], feature_dim: int = 50, frame_stack: int = 1, num_layers: int = 4, num_filters: int = 32,
                 use_spectral_norm: bool = True, use_attention: bool = True):
        super().__init__()
        if num_layers != 4 or num_filters != 32 or not use_spectral_norm or not use_attention:
            raise NotImplementedError("only the configuration the reference instantiates is built (pixel_agent.py:82-88)")
        c, h, w = obs_shape
        self.base_channels, self.frame_stack, self.input_channels, self.use_attention = c, frame_stack, c * frame_stack, True
        self.obs_shape, self.feature_dim = (self.input_channels, h, w), feature_dim
        ch = [self.input_channels, 32, 64, 128, 256]
        self.convs = nn.ModuleList([_SpectralNormConv(ch[i], ch[i + 1]) for i in range(4)])
        self.norms = nn.ModuleList([nn.GroupNorm(min(32, ch[i + 1] // 4), ch[i + 1]) for i in range(4)])
        self.dropouts = nn.ModuleList([nn.Dropout2d(0.1 * (i / 4)) for i in range(4)])
        self.attention = SpatialAttention(256)
        ho = (h + 2 - 3) // 2 + 1
        self.conv_out_dim = 256 * ho * ho
        self.ln = nn.LayerNorm(self.conv_out_dim)
        self.output_layers = nn.Sequential(nn.Linear(self.conv_out_dim, feature_dim * 2), nn.LayerNorm(feature_dim * 2), nn.Mish(), nn.Dropout(0.1),
                                           nn.Linear(feature_dim * 2, feature_dim), nn.LayerNorm(feature_dim), nn.Tanh())
        for m in self.output_layers:
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight); nn.init.zeros_(m.bias)
        self._tab = TableCache(lambda: nat.encoder_table({**{n: p.data for n, p in self.named_parameters()}, **dict(self.named_buffers())},
                                                          feature_dim, self.input_channels, h))
        self._gtab = TableCache(lambda: nat.encoder_table({n: p.grad for n, p in self.named_parameters() if p.grad is not None} |
                                                           {n: p.data for n, p in self.named_parameters() if p.grad is None}, feature_dim, self.input_channels, h))
        self._ws = None
        self.rng_seed, self.rng_offset = 0xE0C0DE, 0

    def trained_parameters(self):
        """Everything the reference's optimiser actually updates: `attention.channel_reduce.*` never receives a gradient, so
        AdamW skips it entirely (no weight decay either) — it stays outside the arena."""
        return [(n, p) for n, p in self.named_parameters() if "channel_reduce" not in n]

    def table(self):
        return self._tab.get(self)

    def grad_table(self):
        return self._gtab.get(self)

    def _stack(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 5:
            b, t, c, h, w = x.shape
            assert t == self.frame_stack, f"Expected {self.frame_stack} frames, got {t}"
            x = x.reshape(b, t * c, h, w)
        elif x.dim() == 4:
            if x.shape[1] == self.base_channels and self.frame_stack > 1:
                x = x.repeat(1, self.frame_stack, 1, 1)
            elif x.shape[1] != self.input_channels:
                raise ValueError(f"Expected {self.input_channels} channels, got {x.shape[1]}")
        elif x.dim() == 3:
            x = x.unsqueeze(0)
        else:
            raise ValueError(f"Unexpected observation shape: {x.shape}")
        if x.dtype == torch.uint8:
            x = x.float() / 255.0
        return x.contiguous().float()

    def forward(self, x: torch.Tensor, noise: Optional[dict] = None, keep_for_backward: bool = False, slot: str = "a"):
        """noise (optional): shift_h / shift_w int32 [B] (fused RandomShiftAugmentation), drop2d_1 [B,64], drop2d_2 [B,128], drop_out [B,2L].
        In train mode missing masks are drawn on device.  With keep_for_backward -> (features, ctx); the activation store of
        each `slot` is a persistent buffer (two passes per train step => two slots), reused across steps."""
        x = self._stack(x)
        B, dev = x.shape[0], x.device
        if self._ws is None:
            self._ws = nat.Workspace(dev)
        noise = dict(noise or {})
        if self.training:
            for key, shape, p in (("drop2d_1", (B, 64), 0.025), ("drop2d_2", (B, 128), 0.05), ("drop_out", (B, 2 * self.feature_dim), 0.1)):
                if noise.get(key) is None:
                    m = torch.empty(shape, device=dev); nat.fill_dropout_mask(m, p, self.rng_seed, self.rng_offset)
                    self.rng_offset += (m.numel() + 3) // 4; noise[key] = m
        bufs = self.__dict__.setdefault("_saved_bufs", {})
        feat, saved = nat.encoder_forward(self.table(), x, noise.get("shift_h"), noise.get("shift_w"), int(noise.get("pad", 4)), noise.get("drop2d_1"),
                                          noise.get("drop2d_2"), noise.get("drop_out"), train=self.training, saved=bufs.get(slot), ws=self._ws)
        bufs[slot] = saved
        if keep_for_backward:
            self._saved = (saved, noise, self.training)
            return feat, self._saved
        return feat

    def backward(self, d_features: torch.Tensor, ctx=None, phase: int = 0, ws=None):
        """Accumulate parameter gradients for the forward whose context is `ctx` (default: the last keep_for_backward call).  phase 1 / 2: the head
        (ln.*, output_layers.*) and the rest as two calls that share `ws` (data parallel: the head gradient's all-reduce starts in between)."""
        saved, noise, train = ctx or self._saved
        nat.encoder_backward(self.table(), self.grad_table(), d_features, saved, noise.get("drop2d_1"), noise.get("drop2d_2"), noise.get("drop_out"),
                             train=train, ws=ws or self._ws, phase=phase)


class _SNConvBias(nn.Module):
    """Holder for a conv with bias under legacy spectral_norm (state_dict order: bias, weight_orig, weight_u, weight_v)."""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        conv = nn.Conv2d(cin, cout, k, padding=k // 2)
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")      # ConvDecoder._initialize_weights sees `weight`
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_orig = nn.Parameter(conv.weight.detach().clone())
        self.register_buffer("weight_u", F.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(torch.randn(cin * k * k), dim=0, eps=1e-12))

    forward = native_only("spectral-norm conv")


class DecoderBlock(nn.Module):
    """visual_encoders.py:355-459 (parameter holder; indices of the reference's nn.Sequential are kept so the keys match)."""

    def __init__(self, in_channels: int, out_channels: int, upsample: bool = False, use_spectral_norm: bool = True):
        super().__init__()
        self.upsample = upsample
        noop = nn.Identity
        main = [_SNConvBias(in_channels, out_channels * 4 if upsample else out_channels, 3)]
        main += ([noop()] if upsample else []) + [noop(), noop(), noop(), _SNConvBias(out_channels, out_channels, 3), noop()]
        self.main_path = nn.Sequential(*main)
        if in_channels != out_channels or upsample:
            self.residual_path = nn.Sequential(_SNConvBias(in_channels, out_channels * 4 if upsample else out_channels, 1), noop(), noop())
        else:
            self.residual_path = nn.Identity()
        self.in_channels, self.out_channels = in_channels, out_channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """visual_encoders.py:455-459 in eval mode as a stand-alone call (aid_decoder_block_forward): x [R, cin, h, h] -> [R, cout, h or 2h, ...].  Inside
        ConvDecoder.forward the same kernels run on the CNHW activations without the layout changes.  Train mode (Dropout2d, spectral-norm power iteration)
        is not built: the reference never trains this decoder and the estimator runs it in eval mode."""
        if self.training:
            raise NotImplementedError("DecoderBlock.forward is built for eval mode (the only mode the reference runs the ConvDecoder in)")
        if not (x.is_cuda and self.main_path[0].bias.is_cuda):
            raise RuntimeError("DecoderBlock.forward runs on the HIP library (module and input on the GPU); there is no CPU path")
        return nat.decoder_block_forward(self, x.contiguous().float())


class ConvDecoder(nn.Module):
    """ConvDecoder (visual_encoders.py:228-352): latent -> [B, 3, 168, 168] image in [0, 1]; state_dict layout of the reference.
    Forward (inference only: the reference never trains it - SURVEY D11 - and the estimator runs it in eval mode) on HIP
    (`aid_conv_decoder_forward`)."""

    def __init__(self, latent_dim: int, output_dim: int = 0, img_channels: int = 3, hidden_dim: int = 256, spatial_size: int = 21,
                 use_spectral_norm: bool = True, device=None):
        super().__init__()
        if hidden_dim < 64 or hidden_dim % 32 or spatial_size != 21 or img_channels != 3 or not use_spectral_norm:
            raise NotImplementedError("built for hidden_dim % 32 == 0 (>= 64), spatial_size 21, 3 image channels, spectral norm "
                                      "(what active_inference.py:133-141 instantiates)")
        self.latent_dim, self.output_dim, self.spatial_size, self.img_channels = latent_dim, output_dim, spatial_size, img_channels
        H = hidden_dim
        self.latent_proj = nn.Sequential(nn.Linear(latent_dim, 4 * H), nn.LayerNorm(4 * H), nn.Mish(), nn.Dropout(0.2),
                                         nn.Linear(4 * H, H * spatial_size * spatial_size), nn.LayerNorm(H * spatial_size * spatial_size), nn.Mish())
        for m in self.latent_proj:
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight); nn.init.zeros_(m.bias)
        self.decoder_blocks = nn.ModuleList([DecoderBlock(H, H, False), DecoderBlock(H, H // 2, True), DecoderBlock(H // 2, H // 4, True),
                                             DecoderBlock(H // 4, H // 8, True)])
        self.output_proj = nn.Sequential(nn.Conv2d(H // 8, 32, 3, padding=1), nn.Identity(), nn.Identity(), nn.Conv2d(32, 32, 3, padding=1), nn.Identity(),
                                         nn.Identity(), nn.Conv2d(32, img_channels, 3, padding=1), nn.Identity())
        for m in self.output_proj:
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu"); nn.init.zeros_(m.bias)
        self._tab = TableCache(lambda: nat.conv_decoder_table({**{n: p.data for n, p in self.named_parameters()}, **dict(self.named_buffers())}, ""))
        self._ws = None

    def forward(self, latent: torch.Tensor, chunk: int = 16) -> torch.Tensor:
        if self.training:
            raise NotImplementedError("ConvDecoder is built for inference (eval mode): call .eval() first; the reference never trains it")
        if self._ws is None:
            self._ws = nat.Workspace(latent.device)
        return nat.conv_decoder_forward(self._tab.get(self), latent.contiguous().float(), chunk=chunk, ws=self._ws)


class RandomShiftAugmentation(nn.Module):
    """Reflect-pad by `pad`, then crop each sample at an independent random offset (visual_encoders.py:471-492)."""

    def __init__(self, pad: int = 4):
        super().__init__()
        self.pad = pad

    def sample_shifts(self, n: int, device):
        return (torch.randint(0, 2 * self.pad + 1, (n,), device=device, dtype=torch.int32),
                torch.randint(0, 2 * self.pad + 1, (n,), device=device, dtype=torch.int32))

    def forward(self, x: torch.Tensor, shifts=None) -> torch.Tensor:
        if not self.training:
            return x
        sh, sw = shifts or self.sample_shifts(x.shape[0], x.device)
        n, c, h, w = x.shape
        out = torch.empty_like(x, dtype=torch.float32)
        import ctypes as C
        nat.check(nat.lib().aid_random_shift(nat._ptr(x.contiguous().float()), C.c_void_p(sh.data_ptr()), C.c_void_p(sw.data_ptr()), self.pad, n, c, h, w,
                                             nat._ptr(out), nat._stream()), "aid_random_shift")
        return out
