"""DiffusionActiveInference — the reference's core module surface
(active_inference_diffusion/core/active_inference.py:19-1063) on hand-written gfx950 kernels.

Same constructor, sub-module names, state_dict keys and method names as the reference; every method body is a
thin call into `aid_amd._native` (the C ABI of libaid_hip.so).  What differs by design (MI355X-first):
  * forward AND backward of each optimiser phase are one fused native call that writes gradients straight
    into the flat parameter arenas (`compute_diffusion_elbo` / `compute_expected_free_energy_diffusion(backward=True)`
    accumulate `.grad`; the returned loss is a detached device scalar — there is no autograd graph to `.backward()`);
  * metrics stay on the device (0-dim tensors) so a training step needs ONE device->host read, not ~40 `.item()`s;
  * randomness comes from a `NoiseSource` (device Philox by default; parity tests inject the reference's draws).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _native as nat
from ..arena import ParamArena
from .. import parallel
from ..models import DiffusionConditionedPolicy, LatentDynamicsModel, LatentScoreNetwork, ValueNetwork
from ..models._holders import TableCache, native_only, params_dict
from ctypes import addressof as ctypes_addressof

from ..noise import NoiseSource, PhiloxNoise
from .diffusion import LatentDiffusionProcess


class SpatialAttentionAggregator(nn.Module):
    """Parameter holder of utils/util.py:7-75: 16 learnable queries attend over the 21x21 feature tokens (nn.MultiheadAttention, 8 heads,
    attention dropout 0.1), then Linear-LN-GELU-Linear.  Evaluated by `aid_pixel_epistemic_value`."""

    def __init__(self, feature_dim: int = 128, num_heads: int = 8, spatial_dim: int = 21):
        super().__init__()
        self.feature_dim, self.num_heads, self.head_dim = feature_dim, num_heads, feature_dim // num_heads
        self.pos_encoding = nn.Parameter(torch.randn(1, spatial_dim * spatial_dim, feature_dim) * 0.02)
        self.attention = nn.MultiheadAttention(embed_dim=feature_dim, num_heads=num_heads, dropout=0.1, batch_first=True)
        self.epistemic_queries = nn.Parameter(torch.randn(1, 16, feature_dim) * 0.02)
        self.output_proj = nn.Sequential(nn.Linear(16 * feature_dim, 512), nn.LayerNorm(512), nn.GELU(), nn.Linear(512, 256))

    forward = native_only("SpatialAttentionAggregator")


class FunctionSpaceEpistemicEstimator(nn.Module):
    """Parameter holder of the MINE estimator (active_inference.py:839-1063, state branch).  Evaluated inside `aid_efe_step`;
    trained by `DiffusionActiveInference.train_epistemic_estimator` (`aid_mine_train_step`)."""

    def __init__(self, decoder: nn.Module, latent_dim: int, observation_shape: int, hidden_dim: int = 256,
                 spatial_aggregator_output_dim: int = 256, is_pixel_observation: bool = False, device="cuda"):
        super().__init__()
        self.decoder, self.latent_dim, self.is_pixel = decoder, latent_dim, bool(is_pixel_observation)
        self.ntk_samples, self.alpha = 4, 0.01
        self.perturbation_scale = nn.Parameter(torch.tensor(0.1))
        self.register_buffer("running_mean", torch.tensor(0.0))
        if self.is_pixel:       # active_inference.py:865-883
            self.pixel_shape = observation_shape
            self.pixel_processor = nn.Sequential(nn.Conv2d(observation_shape[0], 32, 5, stride=2, padding=2), nn.ReLU(), nn.Conv2d(32, 64, 5, stride=2, padding=2),
                                                 nn.ReLU(), nn.Conv2d(64, 128, 5, stride=2, padding=2), nn.ReLU())
            self.spatial_aggregator = SpatialAttentionAggregator(feature_dim=128, num_heads=8, spatial_dim=21)
            jacobian_dim = spatial_aggregator_output_dim * self.ntk_samples
        else:
            self.state_dim = observation_shape
            self.feature_extractor = nn.Sequential(nn.Linear(observation_shape, 128), nn.ReLU(), nn.Linear(128, 256), nn.ReLU(), nn.Linear(256, 128))
            jacobian_dim = 128 * self.ntk_samples
        self.jacobian_projector = nn.Sequential(nn.Linear(jacobian_dim, 512), nn.LayerNorm(512), nn.ReLU(), nn.Dropout(0.1), nn.Linear(512, 256))
        self.latent_processor = nn.Sequential(nn.Linear(latent_dim, 128), nn.ReLU(), nn.Linear(128, 128))
        self.mine_network = nn.Sequential(nn.Linear(spatial_aggregator_output_dim + 128, 512), nn.ReLU(), nn.Dropout(0.1), nn.Linear(512, 512),
                                          nn.ReLU(), nn.Dropout(0.1), nn.Linear(512, 1))

    def forward(self, next_latent_mean: torch.Tensor, next_latent_logvar: torch.Tensor, num_samples: int = 5):
        """active_inference.py:995-1063 as the reference's module call: `estimator(mean, logvar, num_samples)` -> (clamp(mi, 0) expanded to [B], metrics), the
        running mean advanced.  Gradient-free here (the trainable form is `DiffusionActiveInference.train_epistemic_estimator` -> aid_mine_train_step): the
        owning module's `compute_epistemic_value` runs it (aid_epistemic_value / aid_pixel_epistemic_value)."""
        owner = self.__dict__.get("_owner")
        owner = owner() if owner is not None else None
        if owner is None:
            raise RuntimeError("FunctionSpaceEpistemicEstimator.forward needs the DiffusionActiveInference that owns it (its decoder / noise source / workspace)")
        return owner.compute_epistemic_value(next_latent_mean, next_latent_logvar, num_samples)

    def compute_jacobian_features(self, z: torch.Tensor, delta: Optional[torch.Tensor] = None, proj_mask: Optional[torch.Tensor] = None,
                                  attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """active_inference.py:940-993: finite-difference function-space features of the decoder at z [R, L] along `ntk_samples` = 4 random unit directions,
        through the feature extractor (state) or pixel processor + spatial aggregator (pixels) and the jacobian projector -> [R, 256]
        (aid_jacobian_features / aid_pixel_jacobian_features).  `delta` [4, R, L]: the standard-normal draws behind the directions (None: device Philox);
        `proj_mask` / `attn_mask`: explicit dropout masks in train mode (None: drawn)."""
        owner = self.__dict__.get("_owner")
        owner = owner() if owner is not None else None
        if owner is None:
            raise RuntimeError("FunctionSpaceEpistemicEstimator.compute_jacobian_features needs the DiffusionActiveInference that owns it")
        z = z.to(owner.device).contiguous().float()
        if self.is_pixel:
            dec, epi = owner.observation_decoder._tab.get(owner.observation_decoder), owner._pepi_tab.get(self)
        else:
            dec, epi = owner.decoder_table(), owner.epistemic_table()
        return nat.jacobian_features(dec, epi, z, delta=delta, proj_mask=proj_mask, attn_mask=attn_mask, train=self.training, seed=owner.noise.seed,
                                     offset=owner.noise.next_offset(), ws=owner._ws)


class EMAModel:
    """Exponential moving average of the score-network weights (active_inference.py:779-813).  The average itself is
    maintained by the fused optimiser kernel (arena.ema); this object exposes the reference's shadow/apply/restore API."""

    def __init__(self, model: nn.Module, decay: float = 0.9999, device=None, arena: Optional[ParamArena] = None, prefix: str = ""):
        self.model, self.decay, self.device, self.backup = model, decay, device, {}
        self._arena, self._prefix = arena, prefix

    @property
    def shadow(self) -> Dict[str, torch.Tensor]:
        views = self._arena.ema_views()
        return {n: views[self._prefix + n] for n, _ in self.model.named_parameters()}

    def update(self):
        """No-op: `ema = decay*ema + (1-decay)*p` runs inside aid_adamw_step right after the parameter update."""

    def apply_shadow(self):
        sh = self.shadow
        for n, p in self.model.named_parameters():
            self.backup[n] = p.data.clone()
            p.data.copy_(sh[n])

    def restore(self):
        for n, p in self.model.named_parameters():
            p.data.copy_(self.backup[n])


class DiffusionActiveInference(nn.Module):
    def __init__(self, observation_dim: int, action_dim: int, latent_dim: int, config, pixel_shape: Optional[Tuple[int, int, int]] = None,
                 noise: Optional[NoiseSource] = None):
        super().__init__()
        self.observation_dim, self.action_dim, self.latent_dim, self.config = observation_dim, action_dim, latent_dim, config
        self.pixel_shape, self.is_pixel_observation = pixel_shape, bool(config.pixel_observation)
        self.epistemic_dropout_rate = 0.2
        self.device = torch.device(config.device)
        if self.device.type != "cuda":
            raise RuntimeError("aid_amd runs on an AMD GPU (torch device 'cuda' on ROCm); there is no CPU path. "
                               f"Got config.device={config.device!r}.")
        nat.lib()  # fail loudly before building anything if libaid_hip.so is missing
        self.noise = noise or PhiloxNoise(seed=0xA1D)
        self.store_trajectory = False
        self._build_models()
        if hasattr(self, "epistemic_estimator"):      # the estimator module is callable like the reference's: it evaluates through its owner (weak reference: no module cycle)
            import weakref
            self.epistemic_estimator.__dict__["_owner"] = weakref.ref(self)
        self._to_device_and_flatten(self.device)
        self.current_latent, self.latent_trajectory = None, []
        self._ws = nat.Workspace(self.device)
        # Pixel mode: the reference always evaluates its ConvDecoder-based estimator inside the EFE (40 GFLOP per sample row, K*S*B = 25 600
        # rows per horizon step at the default sizes: ~5 PFLOP per train_step).  It is gradient-free, so it is opt-in here.
        self.pixel_epistemic = False
        self.param_version = 0          # bumped whenever parameters change outside an optimiser step (load_state_dict, broadcast)

    # The NoiseSource is shared with the latent_diffusion sub-module (the `noise=None` forms of its continuous_q_sample / q_sample /
    # sample_latent_prior draw from it): assigning `ai.noise = ...` (parity tests, data-parallel seeding) re-points both.
    @property
    def noise(self) -> NoiseSource:
        return self.__dict__["_noise"]

    @noise.setter
    def noise(self, value):
        self.__dict__["_noise"] = value
        ld = self._modules.get("latent_diffusion") if "_modules" in self.__dict__ else None
        if ld is not None:
            ld.noise = value

    # ------------------------------------------------------------------------------------------ construction
    def _build_models(self):
        c, L, D = self.config, self.latent_dim, self.config.hidden_dim
        self.latent_diffusion = LatentDiffusionProcess(c.diffusion, latent_dim=L)
        self.latent_diffusion.noise = self.noise
        self.register_buffer("reward_mean", torch.tensor(0.0))
        self.register_buffer("reward_var", torch.tensor(1.0))
        self.register_buffer("preference_temperature", torch.tensor(float(c.preference_temperature)))
        # The reference sizes the score net's obs encoder with latent_dim (active_inference.py:75-80) and then feeds it raw
        # observations, which only works when latent_dim == observation_dim.  In state mode we size it with the
        # true observation_dim: identical when the reference runs at all, and well-defined (parity unpinned) otherwise.
        score_obs_dim = L if self.is_pixel_observation else self.observation_dim
        self.latent_score_network = LatentScoreNetwork(latent_dim=L, observation_dim=score_obs_dim, hidden_dim=D, use_attention=True)
        self.policy_network = DiffusionConditionedPolicy(latent_dim=L, action_dim=self.action_dim, hidden_dim=D, use_state_dependent_std=True)
        self.value_network = ValueNetwork(state_dim=L, hidden_dim=D, time_embed_dim=128, num_layers=3)
        self.latent_dynamics = LatentDynamicsModel(state_dim=L, action_dim=self.action_dim, hidden_dim=D, num_layers=3)
        if self.is_pixel_observation:
            # Pixel mode (active_inference.py:133-149): observations are the DrQ-v2 features, reconstructed by `feature_decoder`.
            # The ConvDecoder / pixel branch of the epistemic estimator is SURVEY §8(f) rank 1 (not built): the gradient-free
            # epistemic term is skipped, which leaves every parameter update unchanged (it only feeds logged diagnostics).
            self.feature_decoder = nn.Sequential(nn.Linear(L, D), nn.LayerNorm(D), nn.ReLU(), nn.Dropout(self.epistemic_dropout_rate),
                                                 nn.Linear(D, L), nn.Tanh())
            self.reward_predictor = nn.Sequential(nn.Linear(L, D), nn.LayerNorm(D), nn.ReLU(), nn.Linear(D, D // 2), nn.ReLU(), nn.Linear(D // 2, 2))
            if self.pixel_shape is not None:      # ConvDecoder (active_inference.py:133-141): latent -> pixels, inference only
                from ..encoder.visual_encoders import ConvDecoder
                self.observation_decoder = ConvDecoder(latent_dim=L, output_dim=int(np.prod(self.pixel_shape)), img_channels=self.pixel_shape[0], hidden_dim=D)
                self.epistemic_estimator = FunctionSpaceEpistemicEstimator(self.observation_decoder, L, tuple(self.pixel_shape), D,
                                                                           self.config.spatial_aggregator_output_dim, True, self.device)
            return
        p = self.epistemic_dropout_rate
        stage = lambda i, o: nn.Sequential(nn.Linear(i, o), nn.LayerNorm(o), nn.SiLU(), nn.Dropout(p))
        self.observation_decoder = nn.ModuleList([stage(L, 2 * D), stage(2 * D, 2 * D), stage(2 * D, D), nn.Linear(D, self.observation_dim)])
        self.epistemic_estimator = FunctionSpaceEpistemicEstimator(self.observation_decoder, L, self.observation_dim, D,
                                                                   self.config.spatial_aggregator_output_dim, False, self.device)
        self.reward_predictor = nn.Sequential(nn.Linear(L, D), nn.LayerNorm(D), nn.ReLU(), nn.Linear(D, D // 2), nn.ReLU(), nn.Linear(D // 2, 2))

    def _to_device_and_flatten(self, device):
        """Move buffers, then re-home every parameter into its optimiser-group arena (see arena.py)."""
        for mod in self.modules():
            for k, b in list(mod._buffers.items()):
                if b is not None:
                    mod._buffers[k] = b.to(device)
        sn, ld = self.latent_score_network, self.latent_diffusion
        score_named = [("latent_score_network." + n, p) for n, p in sn.arena_order()] + [("latent_diffusion." + n, p) for n, p in ld.named_parameters()]
        self.arenas = {
            "score": ParamArena(score_named, device, with_ema=True),
            "policy": ParamArena([("policy_network." + n, p) for n, p in self.policy_network.named_parameters()], device),
            "value": ParamArena([("value_network." + n, p) for n, p in self.value_network.named_parameters()], device),
            "dynamics": ParamArena([("latent_dynamics." + n, p) for n, p in self.latent_dynamics.named_parameters()], device),
        }
        frozen = [("reward_predictor." + n, p) for n, p in self.reward_predictor.named_parameters()]
        self._rew_tab = TableCache(lambda: nat.reward_table(params_dict(self.reward_predictor), ""))
        if self.is_pixel_observation:
            self.arenas["feature_decoder"] = ParamArena([("feature_decoder." + n, p) for n, p in self.feature_decoder.named_parameters()], device)
            self._fd_tab = TableCache(lambda: self._fd_table(False)); self._fd_gtab = TableCache(lambda: self._fd_table(True))
            self._running_mean = torch.zeros(1, device=device)
        else:
            frozen += [("observation_decoder." + n, p) for n, p in self.observation_decoder.named_parameters()]
            # the estimator's own parameters (everything but its alias of the decoder): trained only by train_epistemic_estimator
            self.arenas["epistemic"] = ParamArena([("epistemic_estimator." + n, p) for n, p in self.epistemic_estimator.named_parameters()
                                                   if not n.startswith("decoder.")], device)
            self._epi_gtab = TableCache(lambda: nat.epistemic_table({n: p.grad for n, p in self.epistemic_estimator.named_parameters()
                                                                      if not n.startswith("decoder.")}, ""))
            self._dec_tab = TableCache(lambda: nat.decoder_table(params_dict(self.observation_decoder), ""))
            self._epi_tab = TableCache(lambda: nat.epistemic_table({n: p.data for n, p in self.epistemic_estimator.named_parameters()
                                                                     if not n.startswith("decoder.")}, ""))
            self._running_mean = self.epistemic_estimator.running_mean.reshape(1).float().contiguous()
        if self.is_pixel_observation and hasattr(self, "observation_decoder"):
            frozen += [("observation_decoder." + n, p) for n, p in self.observation_decoder.named_parameters()]
            frozen += [("epistemic_estimator." + n, p) for n, p in self.epistemic_estimator.named_parameters() if not n.startswith("decoder.")]
            self._pepi_tab = TableCache(lambda: nat.pixel_epistemic_table({n: p.data for n, p in self.epistemic_estimator.named_parameters()
                                                                            if not n.startswith("decoder.")}, ""))
            self._running_mean = self.epistemic_estimator.running_mean.reshape(1).float().contiguous()
        self.arenas["frozen"] = ParamArena(frozen, device, trainable=False)   # never stepped: parameters only
        self.device = torch.device(device)

    def to(self, device, *a, **k):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if device.type != "cuda":
            raise RuntimeError("aid_amd modules live on the GPU; no CPU path exists")
        if device != self.device and not (device.index is None and self.device.type == "cuda"):
            self._to_device_and_flatten(device)
        return self

    def _fd_table(self, grads: bool):
        d = params_dict(self.feature_decoder, grads)
        t = nat.AidFeatureDecoder(); t.l0 = nat._lin(d, "0"); t.ln = nat._ln(d, "1"); t.l4 = nat._lin(d, "4")
        return t

    def state_dict(self, *a, **k):
        if hasattr(self, "epistemic_estimator"):
            self.epistemic_estimator.running_mean = self._running_mean.reshape(()).clone()
        return super().state_dict(*a, **k)

    def load_state_dict(self, sd, strict: bool = True):
        out = super().load_state_dict(sd, strict=strict)
        if hasattr(self, "epistemic_estimator"):
            self._running_mean = self.epistemic_estimator.running_mean.reshape(1).float().contiguous()
        self.arenas["score"].reset_ema()
        self.latent_diffusion._coefs = None
        self.param_version += 1
        return out

    # ------------------------------------------------------------------------------------------ tables
    def decoder_table(self):
        return self._dec_tab.get(self.observation_decoder)

    def reward_table(self):
        return self._rew_tab.get(self.reward_predictor)

    def epistemic_table(self):
        return self._epi_tab.get(self.epistemic_estimator)

    # ------------------------------------------------------------------------------------------ A4
    def decode_observation(self, latent: torch.Tensor, decode_to_pixels: bool = True, masks=None) -> torch.Tensor:
        """active_inference.py:217-242.  Pixel mode: the ConvDecoder image (inference only: the decoder must be in eval mode) or, with
        decode_to_pixels=False, the feature decoder's reconstruction of the encoder features."""
        latent = latent.to(self.device).contiguous().float()
        if self.is_pixel_observation:
            if decode_to_pixels:
                return self.observation_decoder(latent)
            fd = self._fd_tab.get(self.feature_decoder)
            m = None
            if self.feature_decoder.training:
                m = masks
                if m is None:
                    m = torch.empty(latent.shape[0], self.config.hidden_dim, device=self.device)
                    nat.fill_dropout_mask(m, self.epistemic_dropout_rate, self.noise.seed, self.noise.next_offset(m.numel()))
            return nat.feature_decoder_forward(fd, latent, m, self._ws)
        if self.observation_decoder.training and masks is None:
            D, B = self.config.hidden_dim, latent.shape[0]
            masks = [torch.empty(B, w, device=self.device) for w in (2 * D, 2 * D, D)]
            for m in masks:
                nat.fill_dropout_mask(m, self.epistemic_dropout_rate, self.noise.seed, self.noise.next_offset(m.numel()))
        return nat.decode_observation(self.decoder_table(), latent, masks if self.observation_decoder.training else None, self._ws)

    def predict_reward_from_latent(self, latent: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        out = nat.reward_forward(self.reward_table(), latent.to(self.device).contiguous().float(), self._ws)
        return out[:, 0], torch.exp(torch.clamp(out[:, 1], min=-5, max=2))

    def update_belief_via_diffusion(self, observation: torch.Tensor, raw_observation: Optional[torch.Tensor] = None,
                                    noise: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        """active_inference.py:256-312: reverse-diffusion belief + batch statistics + decoded reconstruction error."""
        observation = observation.to(self.device).float()
        if observation.dim() == 1:
            observation = observation.unsqueeze(0)
        observation = observation.contiguous()
        B, L, N = observation.shape[0], self.latent_dim, self.config.diffusion.num_diffusion_steps
        train = self.latent_score_network.training
        if noise is None:
            noise = self.noise.belief(B, L, self.config.hidden_dim, N, train) or {}
        tab, coefs = self.latent_score_network.table(), self.latent_diffusion.coefs()
        if self.is_pixel_observation or self.store_trajectory:
            # the loop itself (optionally keeping all N+1 iterates), then the diagnostics of :286-299 as separate small launches
            out = nat.denoise_loop(tab, coefs, observation, N, noise.get("z_init"), noise.get("step_noise"), noise.get("drop_masks"), train=train,
                                   seed=self.noise.seed, offset=self.noise.next_offset(), keep_traj=self.store_trajectory, ws=self._ws)
            latent, traj = out if self.store_trajectory else (out, None)
            if self.is_pixel_observation:      # :292-295: decode to FEATURE space, compare with the encoder features
                pred = self.decode_observation(latent, decode_to_pixels=False, masks=noise.get("fd_mask"))
            else:
                pred = self.decode_observation(latent, masks=noise.get("dec_masks"))
            self.current_latent = latent
            self.latent_trajectory = list(traj.unbind(0)) if traj is not None else [latent]
            return {"latent": latent, "latent_mean": latent.mean(0), "latent_std": latent.std(0) if B > 1 else torch.zeros_like(latent[0]),
                    "trajectory_length": N + 1, "reconstruction_error": nat.mse(pred, observation), "observation": observation,
                    "raw_observation": raw_observation}
        latent, stats = nat.belief_update(tab, self.decoder_table(), coefs, observation, N,
                                          noise.get("z_init"), noise.get("step_noise"), noise.get("drop_masks"), noise.get("dec_masks"),
                                          train=train, seed=self.noise.seed, offset=self.noise.next_offset(), ws=self._ws)
        self.current_latent, self.latent_trajectory = latent, [latent]      # store_trajectory = False: the iterates are not kept (N+1 x [B,L])
        return {"latent": latent, "latent_mean": stats[:L], "latent_std": stats[L:2 * L], "trajectory_length": N + 1,
                "reconstruction_error": stats[2 * L], "observation": observation, "raw_observation": raw_observation}

    # -- hipGraph replay of small-batch reverse loops (act(), the collector) ----------------------------------------------------
    graph_small_batches = True     # a 50-step loop is ~1 750 launches: below ~64 rows the host (5.4 us per launch) is the bottleneck
    GRAPH_MAX_ROWS = 128

    def _graphed_denoise(self, observation: torch.Tensor) -> torch.Tensor:
        """The loop captured once per (rows, steps, train, parameter storage) and replayed: inputs live in persistent buffers, the
        randomness (initial latent, per-step noise, dropout masks) is drawn by Philox fills BEFORE the replay with fresh offsets, so
        the graph itself holds no RNG state.  Weights are read in place, so optimiser steps need no re-capture."""
        B, L, D = observation.shape[0], self.latent_dim, self.config.hidden_dim
        N, train = self.config.diffusion.num_diffusion_steps, self.latent_score_network.training
        tab = self.latent_score_network.table()
        key = (B, N, train, ctypes_addressof(tab), torch.cuda.current_stream().cuda_stream, nat.gemm_emulation_mode())
        cache = self.__dict__.setdefault("_denoise_graphs", {})
        g = cache.get(key)
        if g is None:
            if len(cache) > 8:
                torch.cuda.synchronize(self.device)      # a replay of an evicted graph may still be in flight on another stream
                cache.clear()
            buf = {"obs": torch.empty(B, self.latent_score_network.observation_dim, device=self.device), "z": torch.empty(B, L, device=self.device),
                   "eps": torch.empty(N, B, L, device=self.device), "mask": torch.empty(N, B, D, device=self.device) if train else None,
                   "ws": nat.Workspace(self.device)}
            buf["obs"].copy_(observation); buf["z"].zero_(); buf["eps"].zero_()
            if train:
                buf["mask"].fill_(1.0)
            run = lambda: nat.denoise_loop(tab, self.latent_diffusion.coefs(), buf["obs"], N, buf["z"], buf["eps"], buf["mask"], train=train, ws=buf["ws"])
            run(); torch.cuda.synchronize()                       # warm-up outside capture: one-time function attributes, workspace growth
            graph = torch.cuda.CUDAGraph()
            from ..streams import side_stream
            with torch.cuda.graph(graph, stream=side_stream(self.device)):     # capture on the process's side stream: torch would otherwise take a fresh one
                buf["out"] = run()                                              # (a process has four hardware queues, aid_amd/streams.py)
            g = cache[key] = (graph, buf)
        graph, buf = g
        buf["obs"].copy_(observation)
        nat.fill_normal(buf["z"], self.noise.seed, self.noise.next_offset(buf["z"].numel()))
        nat.fill_normal(buf["eps"], self.noise.seed, self.noise.next_offset(buf["eps"].numel()))
        if train:
            nat.fill_dropout_mask(buf["mask"], 0.1, self.noise.seed, self.noise.next_offset(buf["mask"].numel()))
        graph.replay()
        return buf["out"].clone()

    def denoise_only(self, observation: torch.Tensor, noise: Optional[dict] = None, ws=None) -> torch.Tensor:
        """The reverse loop without the diagnostic decode/statistics: what train_step needs (it discards the rest).
        `ws`: a private workspace when the loop runs on a side stream next to other native calls."""
        observation = observation.to(self.device).float().contiguous()
        N, train = self.config.diffusion.num_diffusion_steps, self.latent_score_network.training
        if noise is None:
            noise = self.noise.belief(observation.shape[0], self.latent_dim, self.config.hidden_dim, N, train) or {}
            if not noise and ws is None and self.graph_small_batches and observation.shape[0] <= self.GRAPH_MAX_ROWS:
                return self._graphed_denoise(observation)
        return nat.denoise_loop(self.latent_score_network.table(), self.latent_diffusion.coefs(), observation, N, noise.get("z_init"),
                                noise.get("step_noise"), noise.get("drop_masks"), train=train, seed=self.noise.seed,
                                offset=self.noise.next_offset(), ws=ws or self._ws)

    def denoise_many(self, obs_list, noises=None, ws=None) -> torch.Tensor:
        """Several independent belief updates as ONE reverse loop over the concatenated rows (the batch-256 layers are latency-bound:
        512 rows cost 1.4x the time of 256, 768 rows 2x).  `noises`: one hook result per entry (parity tests), or None => the
        NoiseSource is asked once per entry, in order.  Returns [sum(rows), L], rows of obs_list[0] first."""
        N, train = self.config.diffusion.num_diffusion_steps, self.latent_score_network.training
        if noises is None:
            noises = [self.noise.belief(o.shape[0], self.latent_dim, self.config.hidden_dim, N, train) for o in obs_list]
        if self.concurrent_belief_loops and len(obs_list) > 1 and all(128 < o.shape[0] <= self.CHAIN_MAX_ROWS for o in obs_list):
            streams = self._chain_streams_for(len(obs_list))
            if streams is not None:
                return self._denoise_chains(obs_list, noises, ws, N, train, streams)
        noise = {}
        if noises[0] is not None:
            for k, dim in (("z_init", 0), ("step_noise", 1), ("drop_masks", 1)):
                if noises[0].get(k) is not None:
                    noise[k] = torch.cat([n[k] for n in noises], dim=dim).contiguous()
        obs = torch.cat([o.to(self.device).float() for o in obs_list], dim=0).contiguous()
        return nat.denoise_loop(self.latent_score_network.table(), self.latent_diffusion.coefs(), obs, N, noise.get("z_init"),
                                noise.get("step_noise"), noise.get("drop_masks"), train=train, seed=self.noise.seed,
                                offset=self.noise.next_offset(), ws=ws or self._ws)

    # Independent belief updates of up to CHAIN_MAX_ROWS rows each run as separate reverse loops on separate streams instead of one merged loop: a
    # 256-row layer of the loop is a 48 KB-LDS workgroup per CU, so three chains share every CU and each chain's kernel boundaries (drain, launch,
    # ramp: ~4 us of every ~13 us kernel in a dependent chain) are covered by the other two.  3 x 256 rows: 21.1 ms against 25.4 ms merged for the
    # 50-step loop alone, 53.4 against 57.3 ms for the cfg-2 train_step (tools/perf/README.md).  Same results: the rows of a loop never interact.
    # The extra streams are the process-wide pair of aid_amd/streams.py (a process has four hardware queues; a fifth stream would serialise with one of
    # the others, which for two dependent chains is a 2x cliff).
    concurrent_belief_loops = True
    CHAIN_MAX_ROWS = 1024
    MAX_CHAINS = 3

    def _chain_streams_for(self, n: int):
        """[calling stream, extra streams...] for n concurrent loops, or None when they cannot be had."""
        if n > self.MAX_CHAINS:
            return None
        from ..streams import chain_streams
        first = torch.cuda.current_stream()
        extra = [st for st in chain_streams(first.device) if st.cuda_stream != first.cuda_stream]
        if len(extra) < n - 1:
            return None
        if getattr(self, "_chain_ws", None) is None:
            self._chain_ws = [nat.Workspace(self.device) for _ in range(self.MAX_CHAINS - 1)]
            for st in extra:
                nat.stream_hint(st, 1)          # like the agent's side stream: small GEMMs take the low-LDS variants
        return [first] + extra[:n - 1]

    # Each chain is replayed from a hipGraph (captured once per chain slot, row count, step count and mode): a 50-step loop is ~1 700 launches, three of
    # them ~21 ms of host time per train_step - and while the side stream is still busy with the look-ahead ELBO its queue back-pressures the host, so
    # chains two and three used to reach the GPU ten and more milliseconds after the first (tools/perf/timeline.py).  Inputs live in persistent buffers;
    # the randomness is drawn by Philox fills (or copied from the explicit tensors of a parity test) before the replay, so the graph holds no RNG state;
    # weights are read in place, so optimiser steps need no re-capture.  With device Philox randomness the graph path and the direct path draw from
    # different counter offsets (the graph path fills whole buffers before the replay, the direct path draws inside aid_denoise_loop): same
    # distribution, not the same numbers - a run that falls back after a refused capture is not bit-reproducible against one that did not.
    # With explicit noise tensors (parity tests) both paths consume exactly those tensors.
    graph_belief_loops = True

    def _chain_graph(self, slot, rows: int, N: int, train: bool, stream):
        """`slot`: an int for the chains of _denoise_chains (slot i replays on streams[i]), ("start", i) for denoise_start's (replays on chain stream i):
        the two APIs map a slot number to DIFFERENT streams, so they never share a captured graph or its persistent buffers."""
        from ..streams import chain_streams
        tab = self.latent_score_network.table()
        stream = chain_streams(stream.device)[0]          # captured on a chain stream (never the default stream), replayed on whichever stream runs the chain
        key = (slot, rows, N, train, ctypes_addressof(tab), nat.gemm_emulation_mode())      # a captured graph holds the kernels of the arithmetic it was captured under
        cache = self.__dict__.setdefault("_chain_graphs", {})
        g = cache.get(key)
        if g is None:
            if len(cache) > 12:
                # the graphs' buffers were allocated on chain stream 0 and are replayed on the side / main / chain-1 streams without
                # record_stream: nothing may be in flight when the allocator gets them back
                torch.cuda.synchronize(self.device)
                self.__dict__.setdefault("_open_chains", {}).clear()
                cache.clear()
            L, D = self.latent_dim, self.config.hidden_dim
            with torch.cuda.stream(stream):
                buf = {"obs": torch.zeros(rows, self.latent_score_network.observation_dim, device=self.device), "z": torch.zeros(rows, L, device=self.device),
                       "eps": torch.zeros(N, rows, L, device=self.device), "mask": torch.ones(N, rows, D, device=self.device) if train else None,
                       "ws": nat.Workspace(self.device)}
                run = lambda: nat.denoise_loop(tab, self.latent_diffusion.coefs(), buf["obs"], N, buf["z"], buf["eps"], buf["mask"], train=train, ws=buf["ws"])
                run()                                                 # warm-up outside capture: one-time function attributes, workspace growth
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                buf["out"] = run()
            g = cache[key] = (graph, buf)
        return g

    def _launch_chain(self, slot: int, o: torch.Tensor, nz, st, first, N: int, train: bool, w, off: int, graph) -> torch.Tensor:
        """One belief update on stream `st` (already ordered behind `first`): hipGraph replay when `graph` = (graph, buffers), else direct launches."""
        nz = nz or {}
        with torch.cuda.stream(st):
            if st is not first:
                o.record_stream(st)
                for v in nz.values():
                    if torch.is_tensor(v): v.record_stream(st)
            if graph is not None:
                g, buf = graph
                buf["obs"].copy_(o)
                for k, name, kind in (("z_init", "z", 0), ("step_noise", "eps", 0), ("drop_masks", "mask", 1)):
                    if buf[name] is None:
                        continue
                    if nz.get(k) is not None:
                        buf[name].copy_(nz[k])
                    elif kind == 0:
                        nat.fill_normal(buf[name], self.noise.seed, self.noise.next_offset(buf[name].numel()))
                    else:
                        nat.fill_dropout_mask(buf[name], 0.1, self.noise.seed, self.noise.next_offset(buf[name].numel()))
                g.replay()
                out = buf["out"]
            else:
                out = nat.denoise_loop(self.latent_score_network.table(), self.latent_diffusion.coefs(), o, N, nz.get("z_init"), nz.get("step_noise"),
                                       nz.get("drop_masks"), train=train, seed=self.noise.seed, offset=off, ws=w)
        if st is not first:
            out.record_stream(first)
        return out

    def _chain_graphs_for(self, rows_list, N, train, streams):
        if not self.graph_belief_loops:
            return None
        try:
            return [self._chain_graph(i, r, N, train, st) for i, (r, st) in enumerate(zip(rows_list, streams))]
        except RuntimeError as e:          # capture refused (e.g. another capture in flight): same kernels, launched one by one
            import warnings
            warnings.warn(f"aid_amd: hipGraph capture of the belief-update loops failed ({e}); launching them directly from now on")
            self.graph_belief_loops = False
            return None

    def _settle_open_chains(self, streams):
        """A denoise_start handle that was never joined (an exception or early return between start and join) leaves its loop in flight on a chain
        stream: anything that is about to use one of `streams` first orders itself behind it, so no two loops ever interleave on the workspaces
        and stream-ordered buffers of a chain stream."""
        open_ = self.__dict__.setdefault("_open_chains", {})
        for key in [k for k in open_ if any(k == st.cuda_stream for st in streams)]:
            h = open_.pop(key)
            for st in streams:
                st.wait_stream(h["stream"])
            h["joined"] = True

    def _denoise_chains(self, obs_list, noises, ws, N, train, streams) -> torch.Tensor:
        first = streams[0]
        self._settle_open_chains(streams)
        wss = [ws or self._ws] + (getattr(self, "_chain_ws", None) or [])[:len(streams) - 1]
        obs_list = [o.to(self.device).float().contiguous() for o in obs_list]
        offs = [self.noise.next_offset() for _ in obs_list]
        graphs = self._chain_graphs_for([o.shape[0] for o in obs_list], N, train, streams)
        for st in streams[1:]:
            st.wait_stream(first)
        outs = [self._launch_chain(i, o, nz, st, first, N, train, w, off, graphs[i] if graphs is not None else None)
                for i, (o, nz, st, w, off) in enumerate(zip(obs_list, noises, streams, wss, offs))]
        for st in streams[1:]:
            first.wait_stream(st)
        return torch.cat(outs, dim=0)

    def denoise_start(self, observation: torch.Tensor, slot: int, noise: Optional[dict] = None):
        """Start ONE belief update on the process's chain stream `slot` (0 or 1), ordered behind everything enqueued so far on the current stream,
        and return a handle; `denoise_join(handle)` makes the current stream wait for it and returns the latents.  What the pixel agent uses to
        run the belief over obs underneath encode(next_obs), and the belief over next_obs underneath the ELBO / InfoNCE / encoder backward
        (agents/pixel_agent.py).  None when the chains cannot be had (row count outside (128, CHAIN_MAX_ROWS], concurrent loops switched off): the
        caller then takes the merged loop."""
        rows = observation.shape[0]
        if not (self.concurrent_belief_loops and 128 < rows <= self.CHAIN_MAX_ROWS and slot in (0, 1)):
            return None
        first = torch.cuda.current_stream()
        from ..streams import chain_streams
        sts = chain_streams(first.device)
        if any(st.cuda_stream == first.cuda_stream for st in sts):
            return None
        self._chain_streams_for(self.MAX_CHAINS)             # workspaces + stream hints of the chain streams (once)
        st = sts[slot]
        N, train = self.config.diffusion.num_diffusion_steps, self.latent_score_network.training
        observation = observation.to(self.device).float().contiguous()
        if noise is None:
            noise = self.noise.belief(rows, self.latent_dim, self.config.hidden_dim, N, train)
        off = self.noise.next_offset()
        graph = None
        if self.graph_belief_loops:
            try:
                graph = self._chain_graph(("start", slot), rows, N, train, st)
            except RuntimeError as e:
                import warnings
                warnings.warn(f"aid_amd: hipGraph capture of the belief-update loops failed ({e}); launching them directly from now on")
                self.graph_belief_loops = False
        self._settle_open_chains([st])                      # a previous, never-joined loop on this chain stream: this one queues behind it
        st.wait_stream(first)
        out = self._launch_chain(slot, observation, noise, st, first, N, train, self._chain_ws[slot], off, graph)
        h = {"out": out, "stream": st, "joined": False}
        self.__dict__.setdefault("_open_chains", {})[st.cuda_stream] = h
        return h

    def denoise_join(self, handle) -> torch.Tensor:
        if handle.get("joined"):
            raise RuntimeError("denoise_join: this belief-update handle was already joined (its output buffer belongs to the chain's next replay)")
        torch.cuda.current_stream().wait_stream(handle["stream"])
        handle["joined"] = True
        open_ = self.__dict__.setdefault("_open_chains", {})
        if open_.get(handle["stream"].cuda_stream) is handle:
            del open_[handle["stream"].cuda_stream]
        return handle["out"].clone()          # the chain's output buffer is reused by its next replay

    def denoise_pair(self, obs_a: torch.Tensor, obs_b: torch.Tensor, noise_a: Optional[dict] = None, noise_b: Optional[dict] = None) -> torch.Tensor:
        """Two independent belief updates (state_agent.py:135-140) as one 2B-row loop.  Returns [2B, L] (rows of obs_a first)."""
        return self.denoise_many([obs_a, obs_b], None if noise_a is None else [noise_a, noise_b])

    # ------------------------------------------------------------------------------------------ A8 / A10
    def compute_expected_free_energy_diffusion(self, latent: torch.Tensor, horizon: int = 5, num_trajectories: int = 10,
                                               num_ambiguity_samples: int = 10, noise: Optional[dict] = None, backward: bool = False
                                               ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        latent = latent.to(self.device).contiguous().float()
        c = self.config
        if noise is None:
            noise = self.noise.efe(latent.shape[0], self.latent_dim, self.action_dim, horizon, num_trajectories, num_ambiguity_samples)
        pix = self.is_pixel_observation
        pix_epi = pix and self.pixel_epistemic and hasattr(self, "epistemic_estimator")
        B, K, S, H = latent.shape[0], num_trajectories, num_ambiguity_samples, horizon
        next_means = torch.empty(H, K * B, self.latent_dim, device=self.device) if pix_epi else None
        efe, info = nat.efe_step(self.policy_network.table(), self.policy_network.grad_table() if backward else None, self.latent_dynamics.table(),
                                 self.reward_table(), self.value_network.table(), None if pix else self.decoder_table(),
                                 None if pix else self.epistemic_table(), latent,
                                 self._running_mean, horizon, num_trajectories, num_ambiguity_samples, noise=noise,
                                 train=self.training, skip_epistemic=pix, epistemic_weight=c.epistemic_weight, pragmatic_weight=c.pragmatic_weight,
                                 consistency_weight=c.consistency_weight, discount=c.discount_factor, temperature=float(c.preference_temperature),
                                 mine_alpha=0.01, seed=self.noise.seed, offset=self.noise.next_offset(), ws=self._ws, next_means_out=next_means)
        if pix_epi:   # gradient-free pixel estimator on the rollout's dynamics means, added to the EFE afterwards (same value, no gradient)
            dec, epi = self.observation_decoder._tab.get(self.observation_decoder), self._pepi_tab.get(self.epistemic_estimator)
            per_t = (noise or {}).get("pixel_epistemic")
            mine_all = torch.stack([nat.pixel_epistemic_value(dec, epi, next_means[t], K, S, noise=per_t[t] if per_t else None, train=self.training,
                                                              seed=self.noise.seed, offset=self.noise.next_offset(), ws=self._ws) for t in range(H)])
            nat.efe_add_epistemic(mine_all, efe, info, self._running_mean, c.epistemic_weight, c.discount_factor, 0.01)
        out = {k: info[i] for i, k in enumerate(nat.EFE_INFO) if k != "policy_loss"}
        out.update({"num_trajectories": num_trajectories, "horizon": horizon, "epistemic/running_mean": self._running_mean[0],
                    "policy_loss": info[0]})
        return efe, out

    def compute_epistemic_value(self, next_latent_mean: torch.Tensor, next_latent_logvar: torch.Tensor, num_samples: int = 5,
                                noise: Optional[dict] = None) -> Tuple[torch.Tensor, Dict[str, float]]:
        """active_inference.py:398-418: one gradient-free forward of the epistemic estimator (MINE lower bound on I(o; theta | z)) on
        `num_samples` draws of N(mean, exp(logvar)) per row.  -> (clamp(mi, 0) expanded to [B], metrics).  Mutates
        `epistemic_estimator.running_mean` as the reference does under no_grad.  (Inside the EFE rollout the same estimator runs batched
        over the K trajectories in aid_efe_step; this is the stand-alone public method.)"""
        mean = next_latent_mean.to(self.device).contiguous().float()
        logvar = next_latent_logvar.to(self.device).float().expand_as(mean).contiguous()
        B = mean.shape[0]
        if self.is_pixel_observation:
            if not hasattr(self, "epistemic_estimator"):
                raise RuntimeError("pixel-mode compute_epistemic_value needs the ConvDecoder estimator (construct with pixel_shape)")
            if not bool(torch.allclose(logvar, torch.full_like(logvar, float(np.log(0.1))))):
                raise NotImplementedError("the pixel estimator entry point samples with the dynamics model's fixed log-variance ln 0.1")
            dec, epi = self.observation_decoder._tab.get(self.observation_decoder), self._pepi_tab.get(self.epistemic_estimator)
            mine = nat.pixel_epistemic_value(dec, epi, mean, 1, num_samples, noise=noise, train=self.training, seed=self.noise.seed,
                                             offset=self.noise.next_offset(), ws=self._ws)        # [1][4] = joint, marginal, t_exp, mi
            joint, marg, texp, mi = mine[0].cpu().tolist()
            rm = float(self._running_mean[0])
            rm = texp if rm == 0.0 else 0.01 * texp + (1.0 - 0.01) * rm
            self._running_mean.fill_(rm)
        else:
            if noise is None:
                hook = getattr(self.noise, "mine_train", None)        # same draws, same order as one estimator forward
                noise = hook(B, self.latent_dim, num_samples) if hook is not None else None
            out = nat.epistemic_value(self.decoder_table(), self.epistemic_table(), mean, logvar, self._running_mean, num_samples=num_samples,
                                      noise=noise, train=self.epistemic_estimator.training, seed=self.noise.seed, offset=self.noise.next_offset(),
                                      alpha=self.epistemic_estimator.alpha, ws=self._ws)
            mi, joint, marg, _ = out.cpu().tolist()
            rm = float(self._running_mean[0])
        value = torch.full((B,), max(mi, 0.0), device=self.device)
        return value, {"epistemic/mi_estimate": mi, "epistemic/joint_term": joint, "epistemic/marginal_term": marg, "epistemic/running_mean": rm}

    def train_epistemic_estimator(self, latents: torch.Tensor, actions: torch.Tensor, next_latents: torch.Tensor = None, noise: Optional[dict] = None):
        """active_inference.py:420-445: one optimiser step of the MINE estimator on -clamp(MI lower bound, 0).  Needs
        `self.epistemic_optimizer` (the agents attach the one base_agent.py:134-139 defines).  -> (mi, metrics of floats)"""
        if self.is_pixel_observation:
            # the pixel estimator is built for inference (aid_pixel_epistemic_value, what the reference's EFE runs under no_grad); its
            # backward through the ConvDecoder features / attention aggregator is not, and the reference cannot train it either
            raise NotImplementedError("train_epistemic_estimator: the pixel-mode estimator is inference-only here (its training step raises "
                                      "in the reference too, SURVEY D3); use compute_epistemic_value / ai.pixel_epistemic for its value")
        opt = getattr(self, "epistemic_optimizer", None)
        if opt is None:
            raise AttributeError("epistemic_optimizer is not set (the reference's concrete agents never create it, SURVEY D3); "
                                 "aid_amd agents attach it in _setup_optimizers")
        latents = latents.to(self.device).contiguous().float(); actions = actions.to(self.device).contiguous().float()
        S = 5
        if noise is None:
            hook = getattr(self.noise, "mine_train", None)
            noise = hook(latents.shape[0], self.latent_dim, S) if hook is not None else None
        opt.zero_grad()
        out = nat.mine_train_step(self.latent_dynamics.table(), self.decoder_table(), self.epistemic_table(), self._epi_gtab.get(self.epistemic_estimator),
                                  latents, actions, self._running_mean, num_samples=S, noise=noise, train=self.epistemic_estimator.training,
                                  seed=self.noise.seed, offset=self.noise.next_offset(), alpha=self.epistemic_estimator.alpha, ws=self._ws)
        opt.step()
        mi, joint, marg, _ = out.cpu().tolist()
        return max(mi, 0.0), {"epistemic/mi_estimate": mi, "epistemic/joint_term": joint, "epistemic/marginal_term": marg,
                              "epistemic/running_mean": float(self._running_mean[0])}

    def reparameterize(self, mean: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
        eps = torch.empty_like(mean)
        nat.fill_normal(eps, self.noise.seed, self.noise.next_offset(eps.numel()))
        return mean + eps * torch.exp(0.5 * logvar)

    def predict_next_latent(self, latent: torch.Tensor, action: torch.Tensor):
        mean = self.latent_dynamics.predict_next_mean(latent.to(self.device), action.to(self.device))
        return mean, torch.full_like(mean, float(np.log(0.1)))

    # ------------------------------------------------------------------------------------------ A5 / A7
    def compute_diffusion_elbo(self, observations: torch.Tensor, rewards: torch.Tensor, latents: Optional[torch.Tensor] = None,
                               raw_observations: Optional[torch.Tensor] = None, noise: Optional[dict] = None, ws=None
                               ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """Forward + backward of -ELBO in one native call; gradients are ACCUMULATED into the score arena
        (`agent.score_optimizer.zero_grad()` first, as the reference does).  `ws`: workspace to use when the call is enqueued on a
        stream other than the one the module's own workspace serves."""
        ws = ws if ws is not None else self._ws
        observations = observations.to(self.device).contiguous().float()
        rewards = rewards.to(self.device).contiguous().float()
        if latents is None:
            latents = self.update_belief_via_diffusion(observations, raw_observations)["latent"]
        latents = latents.contiguous().float()
        first = not hasattr(self, "time_importance_weights")
        if first:
            self.time_importance_weights = torch.ones(100, device=self.device)
        c, B = self.config, observations.shape[0]
        if noise is None:
            noise = self.noise.elbo(B, self.latent_dim, c.hidden_dim, first, self.time_importance_weights)
        d_obs = None
        preset = None
        if self.is_pixel_observation:   # reconstruction through the feature decoder (gradients to it and to the encoder features)
            d_obs = torch.zeros_like(observations)
            fd_mask = (noise or {}).get("fd_mask")
            if fd_mask is None and self.feature_decoder.training:
                fd_mask = torch.empty(B, c.hidden_dim, device=self.device)
                nat.fill_dropout_mask(fd_mask, self.epistemic_dropout_rate, self.noise.seed, self.noise.next_offset(fd_mask.numel()))
            preset = nat.feature_decoder_step(self._fd_tab.get(self.feature_decoder), self._fd_gtab.get(self.feature_decoder), latents, observations,
                                              fd_mask if self.feature_decoder.training else None, d_obs, ws)
            noise = dict(noise or {}); noise["d_obs_out"] = d_obs
        self._last_d_obs = d_obs
        dp_run = parallel.is_dp()
        t_all = None
        if dp_run:
            # Global-batch semantics for the two batch statistics of the ELBO: t is drawn first so that its all-gather
            # yields (a) the global mean behind the KL annealing weight exp(-5 mean t) and (b) the t column of the rank-major
            # time-importance replay below.
            noise = dict(noise or {})
            if noise.get("t") is None:
                noise["t"] = nat.sample_time(self.time_importance_weights, first, B, self.noise.seed, self.noise.next_offset(4 * B), ws)
            t_all = parallel.allgather_cat(noise["t"].contiguous())
            noise["mean_t"] = t_all.mean().reshape(1)
        metrics, t, per = nat.elbo_step(self.latent_score_network.table(), self.latent_score_network.grad_table(), self.latent_diffusion.table(),
                                        self.latent_diffusion.grad_table(), None if self.is_pixel_observation else self.decoder_table(),
                                        self.reward_table(), observations, rewards, latents,
                                        self.time_importance_weights, first, noise=noise, train=self.latent_score_network.training, recon_preset=preset,
                                        kl_weight=c.kl_weight, diffusion_weight=c.diffusion_weight, reward_weight=c.reward_weight, gp_weight=0.1,
                                        seed=self.noise.seed, offset=self.noise.next_offset(), ws=ws,
                                        defer_time_importance=dp_run)
        if dp_run:   # one global, rank-major order for the order-dependent EMA
            nat.time_importance_update(self.time_importance_weights, t_all, parallel.allgather_cat(per), first)
        info = {k: metrics[i] for i, k in enumerate(nat.ELBO_METRICS) if k != "loss"}
        self._last_elbo = {"t": t, "per_sample_losses": per}
        return metrics[8], info

    def _compute_gradient_penalty(self, noisy_latents: torch.Tensor, t: torch.Tensor, observations: torch.Tensor, drop_mask=None) -> torch.Tensor:
        """active_inference.py:709-729 as a value: mean_b (||d(sum score)/dz_b|| - 1)^2, a device scalar.  The training path does not call
        this: aid_elbo_step computes the same number (its `grad_penalty` metric) AND its double backward in one pass; there is no autograd
        graph here, so the returned scalar cannot be back-propagated.  `t` in [0, 1] takes the score network's continuous-time branch,
        anything else the discrete one (score_networks.py:143-150, decided for the whole batch as the reference does)."""
        f = lambda x: x.to(self.device).float().contiguous()
        z, t, obs = f(noisy_latents), f(t).reshape(-1), f(observations)
        cont = bool(((t >= 0) & (t <= 1)).all())
        if drop_mask is None and self.latent_score_network.training:
            drop_mask = torch.empty(z.shape[0], self.config.hidden_dim, device=self.device)
            nat.fill_dropout_mask(drop_mask, 0.1, self.noise.seed, self.noise.next_offset(drop_mask.numel()))
        pen, _ = nat.gradient_penalty(self.latent_score_network.table(), z, t, obs, cont, drop_mask if self.latent_score_network.training else None,
                                      ws=self._ws)
        return pen

    # ------------------------------------------------------------------------------------------ A11
    def compute_lambda_returns(self, rewards, values, next_values, dones, lambda_: float = 0.95, n_steps: int = 5,
                               exclude_immediate_rewards: bool = False) -> torch.Tensor:
        f = lambda x: x.to(self.device).float().contiguous()
        return nat.lambda_returns(f(rewards), f(next_values), f(dones), self.config.discount_factor, lambda_, n_steps,
                                  exclude_immediate_rewards=exclude_immediate_rewards)

    # ------------------------------------------------------------------------------------------ act
    def act(self, observation: torch.Tensor, deterministic: bool = False, raw_observation: Optional[torch.Tensor] = None):
        """active_inference.py:478-531.  Returns (action on the host, info dict of floats)."""
        belief = self.update_belief_via_diffusion(observation, raw_observation)
        latent = belief["latent"]
        efe, efe_info = self.compute_expected_free_energy_diffusion(latent, horizon=self.config.efe_horizon)
        action, log_prob, dist = self.policy_network(latent, deterministic=deterministic)
        action = action.cpu()
        if action.dim() == 2 and action.shape[0] == 1:
            action = action.squeeze(0)
        info = {**belief, "expected_free_energy": float(efe.mean()), "action_log_prob": float(log_prob.mean()),
                "policy_entropy": float(dist.entropy().sum(dim=-1).mean()),
                **{k: (float(v) if torch.is_tensor(v) else v) for k, v in efe_info.items()}}
        return action, info
